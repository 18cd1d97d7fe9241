/* oracle/pm_oracle.c -- TEST INFRASTRUCTURE ONLY (see pm_oracle.h).
 *
 * A CPU restatement of the reference's primer_match scan path: what each PatternMatch engine
 * returns from find_patterns() over a whole text.  Written from the reference's behaviour, not
 * its code layout; every function cites the reference lines it follows.  Pinned against the real
 * reference (oracle/_ref) by tests/test_oracle_vs_ref.py and against tests/golden/.
 */
#define _POSIX_C_SOURCE 200809L
#include "pm_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */

typedef struct { pmo_hit *v; size_t n, cap; } hitvec;

static int hv_push(hitvec *h, int64_t end, uint32_t pid, int k) {
  if (h->n == h->cap) {
    size_t nc = h->cap ? h->cap * 2 : 1024;
    pmo_hit *nv = (pmo_hit *)realloc(h->v, nc * sizeof(pmo_hit));
    if (!nv) return -1;
    h->v = nv; h->cap = nc;
  }
  pmo_hit *x = &h->v[h->n++];
  memset(x, 0, sizeof(*x));
  x->end = end; x->pid = pid; x->k = (uint8_t)k;
  return 0;
}

void pmo_free(void *p) { free(p); }

void pmo_text_raw(pmo_text *t, const uint8_t *bytes, int64_t n) {
  /* MapFileChars (char_io.h:150-170): ch() and nch() are the identity, size() is 256. */
  t->codes = bytes; t->n = n; t->size = 256;
  for (int i = 0; i < 256; i++) { t->ch[i] = (uint8_t)i; t->nch[i] = i; }
}

void pmo_text_normalized(pmo_text *t, const uint8_t *codes, int64_t n, const uint8_t *table, int table_len) {
  /* Normalized<T> (char_io.t:216-278): chmap_ = .tbl bytes, invchmap_ = inverse or -1. */
  t->codes = codes; t->n = n; t->size = table_len;
  for (int i = 0; i < 256; i++) { t->ch[i] = 0; t->nch[i] = -1; }
  for (int i = 0; i < table_len; i++) { t->ch[i] = table[i]; t->nch[table[i]] = i; }
}

/* cp.pos(p); cp.getch(): the raw character at stream index p.  Reads past the end of the mmap
 * (mapFile.h:49-57, unchecked) see the zero padding of the last page, i.e. code 0. */
static inline uint8_t text_char(const pmo_text *t, int64_t p) {
  uint8_t code = (p >= 0 && p < t->n) ? t->codes[p] : 0;
  return t->ch[code];
}

typedef struct {
  const char *s; int len; uint32_t id; int esb, eeb;
} pat_t;

/* ------------------------------------------------------------------------------------------ */
/* shift_and / shift_and_inexact: masks over the concatenated pattern bit string               */

typedef struct {
  int       W;          /* words */
  int       A;          /* alphabet size */
  uint64_t *u;          /* u[c*W + w] */
  uint64_t *s;          /* first-bit mask */
  uint64_t *last;       /* last-bit mask */
  int      *endbit;     /* per pattern: global bit index of its last char */
} bitmasks;

/* shift_and::computeu (shift_and.cc:53-202) and shift_and_inexact::computeu
 * (shift_and_inexact.cc:90-220), literal patterns only (no -w classes, no regex classes):
 * pattern j occupies bits [sum len<j, sum len<=j) of one long bit string; u[c] has a bit set
 * where the pattern char's code is c; s marks first chars, last marks last chars. */
/* iupac_compatible(w) of the reference (util.cc:121-162), as data */
static const char *iupac_set(unsigned char w) {
  switch (w) {
    case 'A': return "ARMWDHVN"; case 'B': return "GTUCYKSBN"; case 'C': return "CYMSBHVN"; case 'D': return "GATURWKDN";
    case 'G': return "GRKSBDVN"; case 'H': return "ACTUMYWHN"; case 'K': return "GTKBDN"; case 'M': return "ACMHVN";
    case 'N': return "ACGTURYKMSWVDHVN"; case 'R': return "GARDVN"; case 'S': return "GCSBVN"; case 'T': return "TUYKWVDHN";
    case 'U': return "UTYKWVDHN"; case 'V': return "GCARSMVN"; case 'W': return "ATUWDHN"; case 'Y': return "TUCYBHN";
    case 'X': return "MRWSYKVHDBXN";
    case 'a': return "armwdhvn"; case 'b': return "gtucyksbn"; case 'c': return "cymsbhvn"; case 'd': return "gaturwkdn";
    case 'g': return "grksbdvn"; case 'h': return "actumywhn"; case 'k': return "gtkbdn"; case 'm': return "acmhvn";
    case 'n': return "acgturykmswvdhvn"; case 'r': return "gardvn"; case 's': return "gcsbvn"; case 't': return "tuykwvdhn";
    case 'u': return "utykwvdhn"; case 'v': return "gcarsmvn"; case 'w': return "atuwdhn"; case 'y': return "tucybhn";
    case 'x': return "mrwsykvhdbxn";
  }
  return 0;
}

static int g_wc = 0, g_tn = 0;   /* -w / -W for the mask build of the current pmo_find_all call */

static int build_masks(bitmasks *m, const pmo_text *t, const pat_t *p, int np) {
  int64_t bits = 0;
  for (int j = 0; j < np; j++) bits += p[j].len;
  m->W = (int)((bits + 63) / 64);
  if (m->W == 0) m->W = 1;
  m->A = t->size;
  m->u = (uint64_t *)calloc((size_t)m->A * m->W, 8);
  m->s = (uint64_t *)calloc(m->W, 8);
  m->last = (uint64_t *)calloc(m->W, 8);
  m->endbit = (int *)calloc(np ? np : 1, sizeof(int));
  if (!m->u || !m->s || !m->last || !m->endbit) return -1;
  int64_t b = 0;
  for (int j = 0; j < np; j++) {
    for (int i = 0; i < p[j].len; i++, b++) {
      const char *set = g_wc ? iupac_set((uint8_t)p[j].s[i]) : 0;      /* shift_and.cc:108-117 */
      if (set) {
        for (const char *q = set; *q; ++q) {
          int c1 = t->nch[(uint8_t)*q];
          if (c1 >= 0 && c1 < m->A && (*q != 'N' || g_tn)) m->u[(size_t)c1 * m->W + (b >> 6)] |= 1ULL << (b & 63);
        }
      } else {
        int code = t->nch[(uint8_t)p[j].s[i]];
        if (code >= 0 && code < m->A) m->u[(size_t)code * m->W + (b >> 6)] |= 1ULL << (b & 63);
      }
      if (i == 0) m->s[b >> 6] |= 1ULL << (b & 63);
      if (i == p[j].len - 1) { m->last[b >> 6] |= 1ULL << (b & 63); m->endbit[j] = (int)b; }
    }
  }
  return 0;
}

static void free_masks(bitmasks *m) { free(m->u); free(m->s); free(m->last); free(m->endbit); }

/* (X << 1 with carry between words) | s  -- the reference's shift step
 * (shift_and.cc:219-222, shift_and_inexact.cc:266-269). */
static inline uint64_t sh1(const uint64_t *x, const uint64_t *s, int i) {
  return (x[i] << 1) | (i ? (x[i - 1] >> 63) : 0) | s[i];
}

/* shift_and::find_patterns (shift_and.cc:208-255) run to end of text: hit (pos, pattern, 0) for
 * every pattern whose last bit is set after consuming the char at pos-1. */
static int run_shift_and(const pmo_text *t, const pat_t *p, int np, hitvec *out) {
  bitmasks m;
  if (build_masks(&m, t, p, np)) return -1;
  uint64_t *R = (uint64_t *)calloc(m.W, 8), *N = (uint64_t *)calloc(m.W, 8);
  if (!R || !N) return -1;
  /* patterns sorted by end bit already (insertion order) */
  for (int64_t pos = 0; pos < t->n; pos++) {
    int c = t->codes[pos];
    const uint64_t *uc = (c < m.A) ? m.u + (size_t)c * m.W : NULL;
    int any = 0;
    for (int i = 0; i < m.W; i++) {
      N[i] = uc ? (sh1(R, m.s, i) & uc[i]) : 0;
      any |= (N[i] & m.last[i]) != 0;
    }
    uint64_t *tmp = R; R = N; N = tmp;
    if (any)
      for (int j = 0; j < np; j++)
        if (R[m.endbit[j] >> 6] >> (m.endbit[j] & 63) & 1)
          if (hv_push(out, pos + 1, p[j].id, 0)) return -1;
  }
  free(R); free(N); free_masks(&m);
  return 0;
}

/* shift_and_inexact::find_patterns (shift_and_inexact.cc:249-352) run to end of text.
 * State: rows R[0..k]; row l starts with bits 0..l-1 of every pattern set
 * (shift_and_inexact.cc:162-164).  Per char c (code), with sh(X) = (X<<1 | carry) | s:
 *   R0' = sh(R0) & u[c]                                                        (:266-277)
 *   Rl' = sh(Rl) & u[c]; and, only if c is not the eos code (:293):
 *         Rl' |= sh(R(l-1) old)                         substitution            (:295,305)
 *         with indels: |= R(l-1) old (insertion, :278-280,:300-303),
 *                      |= sh(R(l-1) new) | R(l-1) new (deletion, :297-298,:307-308)
 * A hit is a pattern whose last bit is set in row k; its reported level is found by walking
 * down from row k-1 while the bit stays set (:323-328). */
static int run_shift_and_inexact(const pmo_text *t, const pat_t *p, int np, int k, int indels,
                                 uint8_t eos_char, hitvec *out) {
  bitmasks m;
  if (build_masks(&m, t, p, np)) return -1;
  int W = m.W;
  int eos_code = t->nch[eos_char];            /* eos_ = cp.nch(eos_) (:131); -1 never matches */
  uint64_t *R = (uint64_t *)calloc((size_t)(k + 1) * W, 8);
  uint64_t *carry = (uint64_t *)calloc(W, 8);   /* "m1": what row l-1 hands to row l */
  uint64_t *shl = (uint64_t *)calloc(W, 8);     /* "m3": sh(Rl old) */
  uint64_t *old = (uint64_t *)calloc(W, 8);     /* "m0"/"m2": Rl old */
  if (!R || !carry || !shl || !old) return -1;
  {
    int64_t b = 0;
    for (int j = 0; j < np; j++)
      for (int i = 0; i < p[j].len; i++, b++)
        for (int l = i + 1; l <= k; l++) R[(size_t)l * W + (b >> 6)] |= 1ULL << (b & 63);
  }
  for (int64_t pos = 0; pos < t->n; pos++) {
    int c = t->codes[pos];
    const uint64_t *uc = (c < m.A) ? m.u + (size_t)c * W : NULL;
    uint64_t *R0 = R;
    for (int i = W - 1; i >= 0; i--) {          /* descending: sh() reads word i-1 before it changes */
      uint64_t o = R0[i];
      uint64_t x = sh1(R0, m.s, i);
      old[i] = o;
      carry[i] = x | (indels ? o : 0);
      shl[i] = x;
    }
    for (int i = 0; i < W; i++) R0[i] = uc ? (shl[i] & uc[i]) : 0;
    for (int l = 1; l <= k; l++) {
      uint64_t *Rl = R + (size_t)l * W, *Rp = R + (size_t)(l - 1) * W;   /* Rp already updated */
      for (int i = W - 1; i >= 0; i--) { old[i] = Rl[i]; shl[i] = sh1(Rl, m.s, i); }
      for (int i = 0; i < W; i++) Rl[i] = uc ? (shl[i] & uc[i]) : 0;
      if (c != eos_code) {
        for (int i = W - 1; i >= 0; i--) {
          Rl[i] |= carry[i];
          if (indels) Rl[i] |= sh1(Rp, m.s, i) | Rp[i];
          carry[i] = shl[i] | (indels ? old[i] : 0);
        }
      }
    }
    const uint64_t *Rk = R + (size_t)k * W;
    int any = 0;
    for (int i = 0; i < W; i++) any |= (Rk[i] & m.last[i]) != 0;
    if (any)
      for (int j = 0; j < np; j++) {
        int w = m.endbit[j] >> 6, bit = m.endbit[j] & 63;
        if (Rk[w] >> bit & 1) {
          int lvl = k - 1;
          while (lvl >= 0 && (R[(size_t)lvl * W + w] >> bit & 1)) lvl--;
          lvl++;
          if (hv_push(out, pos + 1, p[j].id, lvl)) return -1;
        }
      }
  }
  free(R); free(carry); free(shl); free(old); free_masks(&m);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* editdist_alignment::align  (pattern_alignment.cc:117-705)                                   */

enum { F_EQ = 2, F_WEQ = 4, F_SUB = 8, F_INS = 16, F_DEL = 32, F_VIOL = 64, F_END = 128 };
enum { AC_NONE = 0, AC_EQ = 1, AC_WEQ = 2, AC_SUB = 3, AC_INS = 4, AC_DEL = 5, AC_VIOL = 6 };

/* iupac_compatible(w, c) (util.cc:164-183): c is listed in w's compatibility string */
static int iupac_pair(uint8_t w, uint8_t c) {
  const char *set = w < 128 ? iupac_set(w) : 0;
  return set && c && strchr(set, (char)c) != 0;
}

/* Banded DP filled right to left: row p = last p pattern chars, column t = last t chars of the
 * window [textstart, end2) with textstart = max(0, end - L - k) (:137-148).  Row 0 is free for
 * t <= end2-end (floating right end, :276-279).  Band half-width b = indels ? k : 0.
 * Costs: match 0, substitution/insertion/deletion 1, constraint violation 5k+1 (:131). */
int pmo_editdist_align(const pmo_text *t, const char *pat, int L, int64_t end, int64_t end2,
                       int k, uint8_t eos, int indels, int lconst, int rconst, int yesno,
                       pmo_alignment *out) {
  const int viol = 5 * k + 1;
  int64_t textstart = 0;
  if (end > (int64_t)L + k) textstart = end - L - k;
  const int buflen = (int)(end2 - textstart);
  const int b = indels ? k : 0;
  const int delta = (int)(end2 - end);
  const int cols = buflen + 1;
  unsigned *dp = (unsigned *)malloc(sizeof(unsigned) * (size_t)(L + 1) * cols);
  int *fl = (int *)malloc(sizeof(int) * (size_t)(L + 1) * cols);
  uint8_t *buf = (uint8_t *)malloc((size_t)buflen + 1);
  if (!dp || !fl || !buf) return -1;
  for (int i = 0; i < buflen; i++) buf[i] = text_char(t, textstart + i);
#define D(p_, t_) dp[(size_t)(p_) * cols + (t_)]
#define B(p_, t_) fl[(size_t)(p_) * cols + (t_)]
  int lbexact = 0, rbexact = L + 1;                       /* :230-233 */
  if (lconst > 0) rbexact = L + 1 - lconst;
  if (rconst > 0) lbexact = rconst;
  int rc = 0;
  out->start = 0; out->end = end; out->editdist = INT32_MAX; out->value = viol;

  D(0, 0) = 0; B(0, 0) = F_END;
  int ub = b > L ? L : b;
  for (int p = 1; p <= ub; p++) {                         /* column 0, :253-268 */
    if (!indels || p < lbexact || p >= rbexact || (uint8_t)pat[L - p] == eos) { D(p, 0) = viol; B(p, 0) = F_VIOL; }
    else { D(p, 0) = D(p - 1, 0) + 1; B(p, 0) = F_DEL; }
  }
  ub = delta + b; if (buflen < ub) ub = buflen;
  for (int tt = 1; tt <= ub; tt++) {                      /* row 0, :276-294 */
    if (tt <= delta) { D(0, tt) = 0; B(0, tt) = F_END; }
    else if (!indels || lbexact > 0) { D(0, tt) = viol; B(0, tt) = F_VIOL; }
    else { D(0, tt) = D(0, tt - 1) + 1; B(0, tt) = F_INS; }
  }
  for (int p = 1; p <= L; p++) {                          /* :296-437 */
    int lb = p - b; if (lb < 1) lb = 1;
    ub = p + delta + b; if (buflen < ub) ub = buflen;
    int rowmin = viol;
    const uint8_t pc = (uint8_t)pat[L - p];
    const int p_exact_sub = (p <= lbexact || p >= rbexact);
    const int p_exact_ins = (p < lbexact || p >= rbexact);
    for (int tt = lb; tt <= ub; tt++) {
      const uint8_t tc = buf[buflen - tt];
      unsigned v, v1; int ac;
      if (tc == pc) { v = D(p - 1, tt - 1); ac = F_EQ; }
      else if (g_wc && iupac_pair(pc, tc) && (tc != 'N' || g_tn)) { v = D(p - 1, tt - 1); ac = F_WEQ; }   /* :317-319 */
      else if (tc == eos || pc == eos || p_exact_sub) { v = (unsigned)viol; ac = F_VIOL; }
      else { v = D(p - 1, tt - 1) + 1; ac = F_SUB; }
      if (tc == eos || pc == eos || !indels || tt <= lb || p_exact_ins) {
        v1 = (unsigned)viol; if (v1 < v) { v = v1; ac = F_VIOL; }
      } else {
        v1 = D(p, tt - 1) + 1;
        if (v1 < v) { v = v1; ac = F_INS; } else if (v1 == v) ac |= F_INS;
      }
      if (!indels || pc == eos || tt >= ub || p_exact_sub) {
        v1 = (unsigned)viol; if (v1 < v) { v = v1; ac = F_VIOL; }
      } else {
        v1 = D(p - 1, tt) + 1;
        if (v1 < v) { v = v1; ac = F_DEL; } else if (v1 == v) ac |= F_DEL;
      }
      D(p, tt) = v; B(p, tt) = ac;
      if ((int)v < rowmin) rowmin = (int)v;
    }
    if (rowmin > k) {                                     /* :425-436 */
      out->editdist = INT32_MAX;                          /* alignment_ = [violation] */
      goto done;
    }
  }
  {
    int beststart = L - b;                                /* :443-475 */
    if (beststart > buflen) beststart = buflen;
    if (beststart < 0) beststart = 0;
    int bestval = (int)D(L, beststart);
    ub = L + delta + b; if (buflen < ub) ub = buflen;
    for (int c = beststart + 1; c <= ub; c++) {
      int v = (int)D(L, c);
      if (v < bestval || (v <= bestval && (B(L, c) & (F_EQ | F_WEQ | F_SUB)))) { bestval = v; beststart = c; }
    }
    int p = L, tt = beststart;
    if (tt < p - b || tt > p + b + delta) { out->editdist = INT32_MAX; goto done; }   /* :482-490 */

    /* traceback (:509-590): prefer the diagonal unless it would split an indel run */
    int lastac = AC_NONE, nsub = 0, nins = 0, ndel = 0, nviol = 0;
    while (!(B(p, tt) & F_END)) {
      int ac = B(p, tt);
      int match = ac & (F_EQ | F_WEQ | F_SUB), wcf = ac & F_WEQ, sub = ac & F_SUB, ins = ac & F_INS, del = ac & F_DEL;
      if (match && !((lastac == AC_INS && ins) || (lastac == AC_DEL && del) || (lastac == AC_WEQ && !wcf && (ins || del)))) {
        p--; tt--;
        if ((ac & F_EQ) && !((lastac == AC_WEQ && wcf) || (lastac == AC_SUB && sub))) lastac = AC_EQ;
        else if (wcf) lastac = AC_WEQ;
        else if (sub) lastac = AC_SUB;
      } else if (del) { p--; lastac = AC_DEL; }
      else if (ins) { tt--; lastac = AC_INS; }
      else if (ac & F_VIOL) { p = 0; tt = 0; lastac = AC_VIOL; }
      else { rc = -2; goto done; }
      switch (lastac) { case AC_SUB: nsub++; break; case AC_INS: nins++; break;
                        case AC_DEL: ndel++; break; case AC_VIOL: nviol++; break; default: break; }
    }
    out->start = end2 - beststart;                        /* :603-610 */
    out->end = end2 - tt;
    out->value = bestval;
    out->editdist = nviol ? INT32_MAX : nsub + nins + ndel;
    (void)yesno;
    rc = bestval <= k;
  }
done:
#undef D
#undef B
  free(dp); free(fl); free(buf);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* filter_bitvec  (filter_bitvec.cc:73-196)                                                    */

static int cmp_hit_end(const void *a, const void *b) {
  const pmo_hit *x = (const pmo_hit *)a, *y = (const pmo_hit *)b;
  if (x->end != y->end) return x->end < y->end ? -1 : 1;
  return 0;
}

/* Candidates from shift_and_inexact are sorted by position (:92); each still-live candidate
 * opens a cluster that absorbs later same-pattern candidates while they lie within 2k+1 of the
 * last absorbed one (:103-116); the cluster [first,last] is verified once with
 * editdist_alignment(yesno) and yields at most one hit (pa.end(), pattern, pa.value())
 * (:125-136).  Chunking/deferral (:118-121) does not change the result, so the whole text is
 * treated as one batch. */
static int run_filter_bitvec(const pmo_text *t, const pat_t *p, int np, const pmo_config *cfg, hitvec *out) {
  pat_t *inner = (pat_t *)malloc(sizeof(pat_t) * (np ? np : 1));
  if (!inner) return -1;
  for (int j = 0; j < np; j++) { inner[j] = p[j]; inner[j].id = (uint32_t)(j + 1); }   /* :191-194 */
  hitvec cand = {0, 0, 0};
  if (run_shift_and_inexact(t, inner, np, cfg->k, cfg->indels, cfg->eos, &cand)) return -1;
  qsort(cand.v, cand.n, sizeof(pmo_hit), cmp_hit_end);
  const int win = 2 * cfg->k + 1;
  for (size_t i = 0; i < cand.n; i++) {
    if (cand.v[i].end <= 0) continue;
    const uint32_t pid = cand.v[i].pid;
    const int64_t first = cand.v[i].end;
    int64_t last = first;
    cand.v[i].end = 0;
    for (size_t j = i + 1; j < cand.n && (cand.v[j].end <= last + win); j++) {
      if (cand.v[j].end > 0 && cand.v[j].pid == pid) { last = cand.v[j].end; cand.v[j].end = 0; }
      else if (cand.v[j].end == 0) continue;
    }
    const pat_t *q = &p[pid - 1];
    pmo_alignment al;
    int ok = pmo_editdist_align(t, q->s, q->len, first, last, cfg->k, cfg->eos, cfg->indels,
                                q->esb, q->eeb, 1, &al);
    if (ok < 0) return -1;
    if (ok) if (hv_push(out, al.end, q->id, al.value)) return -1;
  }
  free(cand.v); free(inner);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* keyword_tree (Aho-Corasick)  (keyword_tree.t:190-217, 369-404, 427-486)                     */

typedef struct {
  int  nrel;             /* distinct codes that occur in patterns (relchars_, keyword_tree.t) */
  int  dense[256];       /* code -> 0..nrel-1, or -1 */
  int *child;            /* child[node*nrel + d] or 0 */
  int *fail, *outl;      /* failure link; nearest proper-suffix node that ends a pattern */
  int *head;             /* first pattern ending at node (index into next[]) or -1 */
  int *next;             /* next pattern ending at the same node */
  int  nnodes;
} actrie;

static int ac_build(actrie *a, const pmo_text *t, const pat_t *p, int np) {
  memset(a, 0, sizeof(*a));
  for (int i = 0; i < 256; i++) a->dense[i] = -1;
  int64_t total = 1;
  for (int j = 0; j < np; j++) {
    total += p[j].len;
    for (int i = 0; i < p[j].len; i++) {
      int code = t->nch[(uint8_t)p[j].s[i]];
      if (code >= 0 && code < 256 && a->dense[code] < 0) a->dense[code] = a->nrel++;
    }
  }
  if (a->nrel == 0) a->nrel = 1;
  a->child = (int *)calloc((size_t)total * a->nrel, sizeof(int));
  a->fail = (int *)calloc(total, sizeof(int));
  a->outl = (int *)calloc(total, sizeof(int));
  a->head = (int *)malloc(sizeof(int) * total);
  a->next = (int *)malloc(sizeof(int) * (np ? np : 1));
  int *tail = (int *)malloc(sizeof(int) * total);
  if (!a->child || !a->fail || !a->outl || !a->head || !a->next || !tail) return -1;
  for (int64_t i = 0; i < total; i++) { a->head[i] = -1; tail[i] = -1; }
  a->nnodes = 1;
  for (int j = 0; j < np; j++) {               /* add_keyword_ (keyword_tree.t:77-105) */
    int node = 0, ok = 1;
    for (int i = 0; i < p[j].len; i++) {
      int code = t->nch[(uint8_t)p[j].s[i]];
      if (code < 0) { ok = 0; break; }         /* char absent from the DB alphabet: the -1 code
                                                  (keyword_tree.t:205-207) can never be read */
      int d = a->dense[code];
      int *c = &a->child[(size_t)node * a->nrel + d];
      if (!*c) *c = a->nnodes++;
      node = *c;
    }
    a->next[j] = -1;
    if (!ok || p[j].len == 0) continue;
    if (tail[node] < 0) a->head[node] = j; else a->next[tail[node]] = j;
    tail[node] = j;
  }
  free(tail);
  /* failure_links_ (keyword_tree.t:369-404): BFS; output link = nearest suffix node with a pattern */
  int *queue = (int *)malloc(sizeof(int) * a->nnodes);
  if (!queue) return -1;
  int qh = 0, qt = 0;
  for (int d = 0; d < a->nrel; d++) { int c = a->child[d]; if (c) { a->fail[c] = 0; queue[qt++] = c; } }
  while (qh < qt) {
    int v = queue[qh++];
    for (int d = 0; d < a->nrel; d++) {
      int c = a->child[(size_t)v * a->nrel + d];
      if (!c) continue;
      int w = a->fail[v];
      while (w && !a->child[(size_t)w * a->nrel + d]) w = a->fail[w];
      int u = a->child[(size_t)w * a->nrel + d];
      if (u == c) u = 0;
      a->fail[c] = u;
      a->outl[c] = (a->head[u] >= 0) ? u : a->outl[u];
      queue[qt++] = c;
    }
  }
  free(queue);
  return 0;
}

static void ac_free(actrie *a) { free(a->child); free(a->fail); free(a->outl); free(a->head); free(a->next); }

/* keyword_tree::find_patterns (keyword_tree.t:427-486): on each char follow goto, else fail
 * links; at every node reached emit its own patterns, then those along the output chain. */
static int run_keyword_tree(const pmo_text *t, const pat_t *p, int np, hitvec *out) {
  actrie a;
  if (ac_build(&a, t, p, np)) return -1;
  int node = 0;
  for (int64_t pos = 0; pos < t->n; pos++) {
    int d = a.dense[t->codes[pos]];
    if (d < 0) { node = 0; continue; }
    while (node && !a.child[(size_t)node * a.nrel + d]) node = a.fail[node];
    node = a.child[(size_t)node * a.nrel + d];
    for (int v = node; v; v = a.outl[v])
      for (int j = a.head[v]; j >= 0; j = a.next[j])
        if (hv_push(out, pos + 1, p[j].id, 0)) return -1;
  }
  ac_free(&a);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* primer_alignment::global_align and the l/r extenders (primer_alignment.cc:10-463,568-728)   */

/* Banded global-start DP of `pat` against text[0..textlen) read forwards (dirn>0) or backwards
 * from the end (dirn<0).  yesno form only: returns 1 and (matchlen,value) when some end column
 * within the band has cost <= k.  lbexact/rbexact are passed already resolved (see callers). */
static int global_align(const uint8_t *text, int textlen, const char *pat, int L, int dirn,
                        int lbexact, int rbexact, int k, int indels, uint8_t eos,
                        int *matchlen, int *value) {
  const int viol = 5 * k + 1;
  const int b = indels ? k : 0;
  const int cols = textlen + 1;
  unsigned *dp = (unsigned *)malloc(sizeof(unsigned) * (size_t)(L + 1) * cols);
  int *fl = (int *)malloc(sizeof(int) * (size_t)(L + 1) * cols);
  if (!dp || !fl) return -1;
#define D(p_, t_) dp[(size_t)(p_) * cols + (t_)]
#define B(p_, t_) fl[(size_t)(p_) * cols + (t_)]
  int rc = 0;
  D(0, 0) = 0; B(0, 0) = 0;
  int ub = b; if (L < ub) ub = L;
  for (int p = 1; p <= ub; p++) {                                  /* :64-82 */
    if (!indels || p < lbexact || p >= rbexact) { D(p, 0) = viol; B(p, 0) = F_VIOL; }
    else { D(p, 0) = D(p - 1, 0) + 1; B(p, 0) = F_DEL; }
  }
  ub = b; if (textlen < ub) ub = textlen;
  for (int tt = 1; tt <= ub; tt++) {                               /* :88-112 */
    uint8_t tc = dirn > 0 ? text[tt - 1] : text[textlen - tt];
    /* the reference also tests an uninitialised `patch == eos_` here (:95); taken as false */
    if (!indels || 0 < lbexact || 0 >= rbexact || tc == eos) { D(0, tt) = viol; B(0, tt) = F_VIOL; }
    else { D(0, tt) = D(0, tt - 1) + 1; B(0, tt) = F_INS; }
  }
  for (int p = 1; p <= L; p++) {                                   /* :116-249 */
    int lb = p - b; if (lb < 1) lb = 1;
    ub = p + b; if (textlen < ub) ub = textlen;
    int rowmin = viol;
    const uint8_t pc = (uint8_t)(dirn > 0 ? pat[p - 1] : pat[L - p]);
    for (int tt = lb; tt <= ub; tt++) {
      const uint8_t tc = dirn > 0 ? text[tt - 1] : text[textlen - tt];
      int v, v1, ac, ac1;
      if (tc == pc) { v = (int)D(p - 1, tt - 1); ac = F_EQ; }
      else if (g_wc && iupac_pair(tc, pc) && (g_tn || tc != 'N')) { v = (int)D(p - 1, tt - 1); ac = F_WEQ; }   /* primer_alignment.cc:151-154 */
      else if (tc == eos || pc == eos || p <= lbexact || p >= rbexact) { v = viol; ac = F_VIOL; }
      else { v = (int)D(p - 1, tt - 1) + 1; ac = F_SUB; }
      if (tc == eos || pc == eos || !indels || tt <= lb || p < lbexact || p >= rbexact) { v1 = viol; ac1 = F_VIOL; }
      else { v1 = (int)D(p, tt - 1) + 1; ac1 = F_INS; }
      if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
      if (!indels || tt >= ub || p <= lbexact || p >= rbexact) { v1 = viol; ac1 = F_VIOL; }
      else { v1 = (int)D(p - 1, tt) + 1; ac1 = F_DEL; }
      if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
      D(p, tt) = (unsigned)v; B(p, tt) = ac;
      if (rowmin > v) rowmin = v;
    }
    if (rowmin > k) goto done;                                     /* :243-247 */
  }
  {
    int bestpos = L - b;                                           /* :252-280 */
    if (textlen < bestpos) bestpos = textlen;
    if (bestpos < 0) bestpos = 0;
    int bestval = (int)D(L, bestpos);
    ub = L + b; if (textlen < ub) ub = textlen;
    for (int c = bestpos + 1; c <= ub; c++) {
      int v = (int)D(L, c);
      if (v < bestval || (v <= bestval && (B(L, c) & (F_EQ | F_WEQ | F_SUB)))) { bestval = v; bestpos = c; }
    }
    if (bestpos < L - b || bestpos > L + b) goto done;             /* :285-289 */
    *matchlen = bestpos; *value = bestval;                         /* yesno, :290-299 */
    rc = 1;
  }
done:
#undef D
#undef B
  free(dp); free(fl);
  return rc;
}

/* primer_alignment_lmatch::align, yesno (primer_alignment.cc:568-617): the left part `p1`
 * matched exactly ending at end1; extend `p2` rightwards through len(p2)+k chars.
 * lmatch_-len(p1) is computed unsigned and lands in an int (:608): negative = no constraint. */
static int lmatch_align(const pmo_text *t, int64_t end1, int len1, const char *p2, int len2,
                        int esb, int eeb, int k, int indels, uint8_t eos, int64_t *end, int *value) {
  int buflen = len2 + k;
  uint8_t *buf = (uint8_t *)malloc((size_t)buflen + 1);
  if (!buf) return -1;
  for (int i = 0; i < buflen; i++) buf[i] = text_char(t, end1 + i);
  int lm = (int)(uint32_t)((uint32_t)esb - (uint32_t)len1);
  int lbexact = 0, rbexact = len2 + 1;                       /* dirn>0, :50-53 */
  if ((uint32_t)lm > 0) lbexact = lm;
  if (eeb > 0) rbexact = len2 + 1 - eeb;
  int ml = 0, v = 0;
  int r = global_align(buf, buflen, p2, len2, +1, lbexact, rbexact, k, indels, eos, &ml, &v);
  free(buf);
  if (r == 1) { *end = end1 + ml; *value = v; }
  return r;
}

/* primer_alignment_rmatch::align, yesno (primer_alignment.cc:651-704): the right part `p2`
 * matched exactly ending at end2; extend `p1` leftwards. */
static int rmatch_align(const pmo_text *t, int64_t end2, const char *p1, int len1, int len2,
                        int esb, int eeb, int k, int indels, uint8_t eos, int64_t *end, int *value) {
  int64_t textstart = 0;
  int patlen = len1 + len2 + k;
  if (end2 > (int64_t)patlen) textstart = end2 - patlen;
  int buflen = (int)(end2 - len2 - textstart);
  if (buflen < 0) return 0;
  uint8_t *buf = (uint8_t *)malloc((size_t)buflen + 1);
  if (!buf) return -1;
  for (int i = 0; i < buflen; i++) buf[i] = text_char(t, textstart + i);
  int rm = (int)(uint32_t)((uint32_t)eeb - (uint32_t)len2);
  int lbexact = 0, rbexact = len1 + 1;                       /* dirn<0, :47-49 */
  if (esb > 0) rbexact = len1 + 1 - esb;
  if ((uint32_t)rm > 0) lbexact = rm;
  int ml = 0, v = 0;
  int r = global_align(buf, buflen, p1, len1, -1, lbexact, rbexact, k, indels, eos, &ml, &v);
  free(buf);
  if (r == 1) { *end = end2; *value = v; }
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* exact_halves (exact_halves.cc:120-224) and exact_bases (exact_bases.cc:69-160)              */

static int cmp_seed(const void *a, const void *b) {        /* exact_halves::hit_lessthan (:114-118) */
  const pmo_hit *x = (const pmo_hit *)a, *y = (const pmo_hit *)b;
  if (x->end != y->end) return x->end < y->end ? -1 : 1;
  if (x->pid != y->pid) return x->pid > y->pid ? -1 : 1;
  return 0;
}

static int run_inner_exact(const pmo_text *t, const pat_t *p, int np, int use_shift_and, hitvec *out) {
  return use_shift_and ? run_shift_and(t, p, np, out) : run_keyword_tree(t, p, np, out);
}

/* Each pattern is split at floor(L/2) into inner patterns 2j-1 (left) and 2j (right)
 * (:208-221); exact seed hits are processed in (position asc, inner id desc) order (:142);
 * a left seed is extended rightwards, a right seed leftwards; a verified hit is kept only if
 * its end exceeds the pattern's previous kept end by more than (indels ? 2k : 0) (:163,178). */
static int run_exact_halves(const pmo_text *t, const pat_t *p, int np, const pmo_config *cfg,
                            int inner_sa, hitvec *out) {
  pat_t *halves = (pat_t *)malloc(sizeof(pat_t) * (size_t)(2 * np + 1));
  int64_t *lasthit = (int64_t *)calloc((size_t)np + 1, sizeof(int64_t));
  if (!halves || !lasthit) return -1;
  for (int j = 0; j < np; j++) {
    int l1 = p[j].len / 2;
    halves[2 * j].s = p[j].s; halves[2 * j].len = l1; halves[2 * j].id = (uint32_t)(2 * j + 1);
    halves[2 * j + 1].s = p[j].s + l1; halves[2 * j + 1].len = p[j].len - l1; halves[2 * j + 1].id = (uint32_t)(2 * j + 2);
    halves[2 * j].esb = halves[2 * j].eeb = halves[2 * j + 1].esb = halves[2 * j + 1].eeb = 0;
  }
  hitvec seeds = {0, 0, 0};
  if (run_inner_exact(t, halves, 2 * np, inner_sa, &seeds)) return -1;
  qsort(seeds.v, seeds.n, sizeof(pmo_hit), cmp_seed);
  for (size_t i = 0; i < seeds.n; i++) {
    uint32_t hid = seeds.v[i].pid;                 /* 1-based inner id */
    int j = (int)((hid - 1) / 2);
    const pat_t *q = &p[j];
    int l1 = q->len / 2, l2 = q->len - l1;
    int64_t end = 0; int val = 0, ok;
    if (hid % 2 == 1) ok = lmatch_align(t, seeds.v[i].end, l1, q->s + l1, l2, q->esb, q->eeb, cfg->k, cfg->indels, cfg->eos, &end, &val);
    else              ok = rmatch_align(t, seeds.v[i].end, q->s, l1, l2, q->esb, q->eeb, cfg->k, cfg->indels, cfg->eos, &end, &val);
    if (ok < 0) return -1;
    if (ok && end > lasthit[j + 1] + (cfg->indels ? 2 * cfg->k : 0)) {
      if (hv_push(out, end, q->id, val)) return -1;
      lasthit[j + 1] = end;
    }
  }
  free(seeds.v); free(halves); free(lasthit);
  return 0;
}

/* Seed = the mandated exact prefix (esb >= eeb) or suffix; the remainder is verified by the
 * same extenders; no dedup, hits in inner emission order (exact_bases.cc:92-121,138-154). */
static int run_exact_bases(const pmo_text *t, const pat_t *p, int np, const pmo_config *cfg,
                           int inner_sa, hitvec *out) {
  pat_t *seedp = (pat_t *)malloc(sizeof(pat_t) * (size_t)(np ? np : 1));
  if (!seedp) return -1;
  for (int j = 0; j < np; j++) {
    seedp[j].id = (uint32_t)(j + 1); seedp[j].esb = seedp[j].eeb = 0;
    if (p[j].esb >= p[j].eeb) { seedp[j].s = p[j].s; seedp[j].len = p[j].esb; }
    else { seedp[j].s = p[j].s + (p[j].len - p[j].eeb); seedp[j].len = p[j].eeb; }
  }
  hitvec seeds = {0, 0, 0};
  if (run_inner_exact(t, seedp, np, inner_sa, &seeds)) return -1;
  for (size_t i = 0; i < seeds.n; i++) {
    int j = (int)seeds.v[i].pid - 1;
    const pat_t *q = &p[j];
    int64_t end = 0; int val = 0, ok;
    if (q->esb >= q->eeb) ok = lmatch_align(t, seeds.v[i].end, q->esb, q->s + q->esb, q->len - q->esb, q->esb, q->eeb, cfg->k, cfg->indels, cfg->eos, &end, &val);
    else ok = rmatch_align(t, seeds.v[i].end, q->s, q->len - q->eeb, q->eeb, q->esb, q->eeb, cfg->k, cfg->indels, cfg->eos, &end, &val);
    if (ok < 0) return -1;
    if (ok) if (hv_push(out, end, q->id, val)) return -1;
  }
  free(seeds.v); free(seedp);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* pick_pattern_index, automatic branch (select.cc:31-148, NOPRIMEGEN, seedlen 0)              */

int pmo_pick_engine(const pmo_text *t, int k, int indels, int wildcards, int npat,
                    const int32_t *patlen, const int32_t *esb, const int32_t *eeb) {
  (void)indels;
  long min_exact = INT_MAX, cumbooldiff = 0, cumdiff = 0, min_inexact = INT_MAX, min_len = INT_MAX;
  for (int i = 0; i < npat; i++) {
    int e = esb ? esb[i] : 0, f = eeb ? eeb[i] : 0;
    int c = (e >= f) ? e : f;                                      /* :41-62 */
    if (min_exact > c) min_exact = c;
    cumdiff += c - patlen[i] / 2;
    cumbooldiff += (c - patlen[i] / 2) >= 0 ? 1 : 0;
    if (min_inexact > -(c - patlen[i])) min_inexact = -(c - patlen[i]);
    if (min_len > patlen[i]) min_len = patlen[i];
  }
  if (min_inexact > min_len) min_inexact = min_len;
  if (k >= min_inexact && k > 0) return -1;                        /* :87-90 */
  int sel;
  if (wildcards) sel = 4;
  else if (t->size < 255) sel = (t->nch['A'] == 0 && t->nch['C'] == 1 && t->nch['G'] == 2 && t->nch['T'] == 3) ? 2 : 3;
  else sel = 3;
  if (k > 0) {
    if (k == 1 && ((min_len >= 12 && t->size < 10) || (min_len >= 8 && t->size >= 10)) &&
        (cumbooldiff <= 0 || cumdiff <= 0)) sel = 11 + sel - 1;    /* :121-126 */
    else if (min_exact >= 6) sel = 7 + sel - 1;                     /* :131-133 */
    else sel = 5;                                                   /* :137-139 */
  }
  return sel;
}

/* ------------------------------------------------------------------------------------------ */

int pmo_find_all(const pmo_text *t, const pmo_config *cfg,
                 const char *patbuf, const int64_t *patoff, int npat, const uint32_t *ids,
                 const int32_t *esb, const int32_t *eeb, pmo_hit **out, size_t *nout) {
  if (cfg->wildcards && (cfg->engine >= 1 && cfg->engine <= 3)) return -3;   /* keyword trees know no IUPAC classes (select.cc:101-102 picks shift_and) */
  g_wc = cfg->wildcards; g_tn = cfg->text_n;
  pat_t *p = (pat_t *)malloc(sizeof(pat_t) * (size_t)(npat ? npat : 1));
  int32_t *plen = (int32_t *)malloc(sizeof(int32_t) * (size_t)(npat ? npat : 1));
  if (!p || !plen) return -1;
  for (int j = 0; j < npat; j++) {
    p[j].s = patbuf + patoff[j]; p[j].len = (int)(patoff[j + 1] - patoff[j]);
    p[j].id = ids ? ids[j] : (uint32_t)(j + 1);
    p[j].esb = esb ? esb[j] : 0; p[j].eeb = eeb ? eeb[j] : 0;
    plen[j] = p[j].len;
  }
  int engine = cfg->engine;
  if (engine == PMO_AUTO) engine = pmo_pick_engine(t, cfg->k, cfg->indels, cfg->wildcards, npat, plen, esb, eeb);
  hitvec hv = {0, 0, 0};
  int rc;
  switch (engine) {
    case 1: case 2: case 3: rc = run_keyword_tree(t, p, npat, &hv); break;
    case 4: rc = run_shift_and(t, p, npat, &hv); break;
    case 5: rc = run_filter_bitvec(t, p, npat, cfg, &hv); break;
    case 7: case 8: case 9: rc = run_exact_bases(t, p, npat, cfg, 0, &hv); break;
    case 10: rc = run_exact_bases(t, p, npat, cfg, 1, &hv); break;
    case 11: case 12: case 13: rc = run_exact_halves(t, p, npat, cfg, 0, &hv); break;
    case 14: rc = run_exact_halves(t, p, npat, cfg, 1, &hv); break;
    case PMO_SHIFT_AND_INEXACT: rc = run_shift_and_inexact(t, p, npat, cfg->k, cfg->indels, cfg->eos, &hv); break;
    default: rc = -4;
  }
  free(p); free(plen);
  if (rc) { free(hv.v); return rc; }
  *out = hv.v; *nout = hv.n;
  return 0;
}

int pmo_cli_align(const pmo_text *t, const pmo_config *cfg, const char *pat, int patlen,
                  int esb, int eeb, int64_t end, pmo_alignment *out) {
  g_wc = cfg->wildcards; g_tn = cfg->text_n;
  if (cfg->k == 0 && !cfg->wildcards) {   /* exact_alignment (pattern_alignment.cc:29-43) */
    out->start = end - patlen; out->end = end; out->editdist = 0; out->value = 0;
    return 1;
  }
  if (cfg->k == 0) {                 /* exact_wc_alignment (pattern_alignment.cc:70-93) */
    int subs = 0;
    for (int i = 0; i < patlen; i++) {
      const uint8_t tc = text_char(t, end - patlen + i), pc = (uint8_t)pat[i];
      if (tc != pc && !(iupac_pair(tc, pc) && (g_tn || tc != 'N'))) subs++;
    }
    out->start = end - patlen; out->end = end; out->editdist = subs; out->value = 0;
    return subs <= 0;
  }
  /* editdist_alignment(key,key,k,eos,wc,tn,indels,dm,esb,eeb,false) (primer_match.cc:1143-1149) */
  return pmo_editdist_align(t, pat, patlen, end, end, cfg->k, cfg->eos, cfg->indels, esb, eeb, 0, out);
}

double pmo_time_find_all(const pmo_text *t, const pmo_config *cfg,
                         const char *patbuf, const int64_t *patoff, int npat, size_t *nout) {
  struct timespec a, b;
  pmo_hit *h = NULL; size_t n = 0;
  clock_gettime(CLOCK_MONOTONIC, &a);
  int rc = pmo_find_all(t, cfg, patbuf, patoff, npat, NULL, NULL, NULL, &h, &n);
  clock_gettime(CLOCK_MONOTONIC, &b);
  free(h);
  if (nout) *nout = rc ? 0 : n;
  if (rc) return -1.0;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

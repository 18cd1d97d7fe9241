// pm_align.cpp -- see pm_align.h.  Banded storage: row p keeps only columns t in
// [p-b, p+delta+b] (b = band half-width, delta = floating-end slack), addressed by
// o = t-(p-b), so a verify costs O(L*(delta+2b+1)) memory and time instead of O(L*window).
#include "pm_align.h"

#include <algorithm>
#include <climits>
#include <cstring>

#include "pm_iupac.h"

namespace pm {
namespace {

enum : uint16_t { EQ = 2, WEQ = 4, SUB = 8, INS = 16, DEL = 32, VIOL = 64, END = 128 };
enum Step { S_NONE, S_EQ, S_WEQ, S_SUB, S_INS, S_DEL, S_VIOL };

// iupac_compatible(w, c) (reference util.cc:164-183): c is listed in w's compatibility string
inline bool iupac_pair(uint8_t w, uint8_t c) {
  const char *set = w < 128 ? iupac_compatible_set(w) : nullptr;
  return set && c && strchr(set, (char)c) != nullptr;
}

struct Band {
  uint32_t *dp; uint16_t *fl; int W, b;
  inline int at(int p, int t) const { return p * W + (t - (p - b)); }
};

Band make_band(AlignScratch &scr, int rows, int W, int b) {
  size_t need = (size_t)rows * W;
  if (scr.dp.size() < need) { scr.dp.resize(need); scr.fl.resize(need); }
  return Band{scr.dp.data(), scr.fl.data(), W, b};
}

}  // namespace

// reference pattern_alignment.cc:117-705.  Row p = last p pattern chars, column t = last t
// window chars.  Row 0 is free while t <= delta (:276-279); costs 0/1/1/1, violation 5k+1 (:131).
AlignResult editdist_align(const uint8_t *win, int64_t win_start, int64_t end, int64_t end2,
                           const char *pat, int L, int lconst, int rconst,
                           const AlignParams &prm, AlignScratch &scr, std::string *ops) {
  AlignResult r;
  r.end = end; r.editdist = INT32_MAX;
  const int k = prm.k, viol = 5 * k + 1, b = prm.indels ? k : 0;
  const uint8_t eos = prm.eos;
  const int buflen = (int)(end2 - win_start);
  const int delta = (int)(end2 - end);
  const int W = delta + 2 * b + 2;
  Band B = make_band(scr, L + 1, W, b);
  int lbexact = 0, rbexact = L + 1;                       // :230-233
  if (lconst > 0) rbexact = L + 1 - lconst;
  if (rconst > 0) lbexact = rconst;

  B.dp[B.at(0, 0)] = 0; B.fl[B.at(0, 0)] = END;
  for (int p = 1, ub = std::min(b, L); p <= ub; ++p) {    // column 0 (:253-268)
    int i = B.at(p, 0);
    if (!prm.indels || p < lbexact || p >= rbexact || (uint8_t)pat[L - p] == eos) { B.dp[i] = viol; B.fl[i] = VIOL; }
    else { B.dp[i] = B.dp[B.at(p - 1, 0)] + 1; B.fl[i] = DEL; }
  }
  for (int t = 1, ub = std::min(buflen, delta + b); t <= ub; ++t) {   // row 0 (:276-294)
    int i = B.at(0, t);
    if (t <= delta) { B.dp[i] = 0; B.fl[i] = END; }
    else if (!prm.indels || lbexact > 0) { B.dp[i] = viol; B.fl[i] = VIOL; }
    else { B.dp[i] = B.dp[B.at(0, t - 1)] + 1; B.fl[i] = INS; }
  }
  for (int p = 1; p <= L; ++p) {                          // :296-437
    const int lb = std::max(1, p - b), ub = std::min(buflen, p + delta + b);
    const uint8_t pc = (uint8_t)pat[L - p];
    const bool zone_sub = (p <= lbexact || p >= rbexact);
    const bool zone_ins = (p < lbexact || p >= rbexact);
    int rowmin = viol;
    for (int t = lb; t <= ub; ++t) {
      const uint8_t tc = win[buflen - t];
      uint32_t v, v1; uint16_t ac;
      if (tc == pc) { v = B.dp[B.at(p - 1, t - 1)]; ac = EQ; }
      else if (prm.wc && iupac_pair(pc, tc) && (tc != 'N' || prm.tn)) { v = B.dp[B.at(p - 1, t - 1)]; ac = WEQ; }   // :317-319
      else if (tc == eos || pc == eos || zone_sub) { v = (uint32_t)viol; ac = VIOL; }
      else { v = B.dp[B.at(p - 1, t - 1)] + 1; ac = SUB; }
      if (tc == eos || pc == eos || !prm.indels || t <= lb || zone_ins) {
        if ((uint32_t)viol < v) { v = (uint32_t)viol; ac = VIOL; }
      } else {
        v1 = B.dp[B.at(p, t - 1)] + 1;
        if (v1 < v) { v = v1; ac = INS; } else if (v1 == v) ac |= INS;
      }
      if (!prm.indels || pc == eos || t >= ub || zone_sub) {
        if ((uint32_t)viol < v) { v = (uint32_t)viol; ac = VIOL; }
      } else {
        v1 = B.dp[B.at(p - 1, t)] + 1;
        if (v1 < v) { v = v1; ac = DEL; } else if (v1 == v) ac |= DEL;
      }
      const int i = B.at(p, t);
      B.dp[i] = v; B.fl[i] = ac;
      rowmin = std::min(rowmin, (int)v);
    }
    if (rowmin > k) return r;                             // :425-436
  }
  int best = std::max(0, std::min(L - b, buflen));        // :443-475
  int bestval = (int)B.dp[B.at(L, best)];
  for (int c = best + 1, ub = std::min(buflen, L + delta + b); c <= ub; ++c) {
    const int v = (int)B.dp[B.at(L, c)];
    if (v < bestval || (v <= bestval && (B.fl[B.at(L, c)] & (EQ | WEQ | SUB)))) { bestval = v; best = c; }
  }
  int p = L, t = best;
  if (t < p - b || t > p + b + delta) return r;           // :482-490
  Step last = S_NONE;
  int nsub = 0, nins = 0, ndel = 0, nviol = 0;
  while (!(B.fl[B.at(p, t)] & END)) {                     // traceback (:514-590)
    const uint16_t ac = B.fl[B.at(p, t)];
    const bool match = ac & (EQ | WEQ | SUB), wcf = ac & WEQ, sub = ac & SUB, ins = ac & INS, del = ac & DEL;
    if (match && !((last == S_INS && ins) || (last == S_DEL && del) || (last == S_WEQ && !wcf && (ins || del)))) {
      --p; --t;
      if ((ac & EQ) && !((last == S_WEQ && wcf) || (last == S_SUB && sub))) last = S_EQ;
      else if (wcf) last = S_WEQ;
      else if (sub) { last = S_SUB; }
      if (last == S_SUB) ++nsub;
    } else if (del) { --p; last = S_DEL; ++ndel; }
    else if (ins) { --t; last = S_INS; ++nins; }
    else if (ac & VIOL) { p = 0; t = 0; last = S_VIOL; ++nviol; }
    else return r;
    // pattern_alignment::alignment_string's characters (pattern_alignment.h:122-165), pattern start first
    if (ops) ops->push_back(last == S_EQ ? '|' : last == S_WEQ ? '+' : last == S_SUB ? '*' : last == S_INS ? '^' : last == S_DEL ? 'v' : '!');
  }
  r.start = end2 - best;                                  // :603-610
  r.end = end2 - t;
  r.value = bestval;
  r.editdist = nviol ? INT32_MAX : nsub + nins + ndel;
  r.ok = bestval <= k;
  return r;
}

namespace {

// reference primer_alignment.cc:10-299 (yesno form).  text is read forwards (dirn>0) or from
// its end backwards (dirn<0); band |t-p| <= b; the end column is chosen by the same rule as
// above (:253-280).
bool global_align(const uint8_t *text, int textlen, const char *pat, int L, int dirn,
                  int lbexact, int rbexact, const AlignParams &prm, AlignScratch &scr,
                  int *matchlen, int *value) {
  const int k = prm.k, viol = 5 * k + 1, b = prm.indels ? k : 0;
  const uint8_t eos = prm.eos;
  const int W = 2 * b + 2;
  Band B = make_band(scr, L + 1, W, b);
  B.dp[B.at(0, 0)] = 0; B.fl[B.at(0, 0)] = 0;
  for (int p = 1, ub = std::min(b, L); p <= ub; ++p) {    // :64-82
    int i = B.at(p, 0);
    if (!prm.indels || p < lbexact || p >= rbexact) { B.dp[i] = viol; B.fl[i] = VIOL; }
    else { B.dp[i] = B.dp[B.at(p - 1, 0)] + 1; B.fl[i] = DEL; }
  }
  for (int t = 1, ub = std::min(b, textlen); t <= ub; ++t) {   // :88-112
    const uint8_t tc = dirn > 0 ? text[t - 1] : text[textlen - t];
    int i = B.at(0, t);
    if (!prm.indels || 0 < lbexact || 0 >= rbexact || tc == eos) { B.dp[i] = viol; B.fl[i] = VIOL; }
    else { B.dp[i] = B.dp[B.at(0, t - 1)] + 1; B.fl[i] = INS; }
  }
  for (int p = 1; p <= L; ++p) {                          // :116-249
    const int lb = std::max(1, p - b), ub = std::min(textlen, p + b);
    const uint8_t pc = (uint8_t)(dirn > 0 ? pat[p - 1] : pat[L - p]);
    int rowmin = viol;
    for (int t = lb; t <= ub; ++t) {
      const uint8_t tc = dirn > 0 ? text[t - 1] : text[textlen - t];
      int v, v1; uint16_t ac, ac1;
      if (tc == pc) { v = (int)B.dp[B.at(p - 1, t - 1)]; ac = EQ; }
      else if (prm.wc && iupac_pair(tc, pc) && (prm.tn || tc != 'N')) { v = (int)B.dp[B.at(p - 1, t - 1)]; ac = WEQ; }   // primer_alignment.cc:151-154
      else if (tc == eos || pc == eos || p <= lbexact || p >= rbexact) { v = viol; ac = VIOL; }
      else { v = (int)B.dp[B.at(p - 1, t - 1)] + 1; ac = SUB; }
      if (tc == eos || pc == eos || !prm.indels || t <= lb || p < lbexact || p >= rbexact) { v1 = viol; ac1 = VIOL; }
      else { v1 = (int)B.dp[B.at(p, t - 1)] + 1; ac1 = INS; }
      if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
      if (!prm.indels || t >= ub || p <= lbexact || p >= rbexact) { v1 = viol; ac1 = VIOL; }
      else { v1 = (int)B.dp[B.at(p - 1, t)] + 1; ac1 = DEL; }
      if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
      const int i = B.at(p, t);
      B.dp[i] = (uint32_t)v; B.fl[i] = ac;
      rowmin = std::min(rowmin, v);
    }
    if (rowmin > k) return false;                         // :243-247
  }
  int best = std::max(0, std::min(L - b, textlen));       // :252-280
  int bestval = (int)B.dp[B.at(L, best)];
  for (int c = best + 1, ub = std::min(textlen, L + b); c <= ub; ++c) {
    const int v = (int)B.dp[B.at(L, c)];
    if (v < bestval || (v <= bestval && (B.fl[B.at(L, c)] & (EQ | WEQ | SUB)))) { bestval = v; best = c; }
  }
  if (best < L - b || best > L + b) return false;         // :285-289
  *matchlen = best; *value = bestval;
  return true;
}

}  // namespace

// reference primer_alignment.cc:568-617.  `lmatch_ - pattern1.length()` is evaluated unsigned
// and lands in an int (:608,:51): a negative result means "no exact-prefix constraint left".
bool lmatch_extend(const uint8_t *win, int64_t end1, int len1, const char *p2, int len2,
                   int esb, int eeb, const AlignParams &prm, AlignScratch &scr,
                   int64_t *end, int *value) {
  const uint32_t lm = (uint32_t)esb - (uint32_t)len1;
  int lbexact = 0, rbexact = len2 + 1;
  if (lm > 0) lbexact = (int)lm;
  if (eeb > 0) rbexact = len2 + 1 - eeb;
  int ml = 0, v = 0;
  if (!global_align(win, len2 + prm.k, p2, len2, +1, lbexact, rbexact, prm, scr, &ml, &v)) return false;
  *end = end1 + ml; *value = v;
  return true;
}

// reference primer_alignment.cc:651-704.
bool rmatch_extend(const uint8_t *win, int winlen, int64_t end2, const char *p1, int len1, int len2,
                   int esb, int eeb, const AlignParams &prm, AlignScratch &scr,
                   int64_t *end, int *value) {
  if (winlen < 0) return false;
  const uint32_t rm = (uint32_t)eeb - (uint32_t)len2;
  int lbexact = 0, rbexact = len1 + 1;
  if (esb > 0) rbexact = len1 + 1 - esb;
  if (rm > 0) lbexact = (int)rm;
  int ml = 0, v = 0;
  if (!global_align(win, winlen, p1, len1, -1, lbexact, rbexact, prm, scr, &ml, &v)) return false;
  *end = end2; *value = v;
  return true;
}

}  // namespace pm

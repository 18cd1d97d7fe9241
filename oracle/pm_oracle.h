/* oracle/pm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's primer_match scan path, used as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product path (sequence-alignment-tools_amd/, include/) may include, link or call this.
 *
 * Parity is PINNED: every engine below is fuzzed against the real reference, built from
 * /root/reference into oracle/_ref/ (oracle/Makefile, oracle/ref_harness.cc), by
 * tests/test_oracle_vs_ref.py, and checked against the committed vectors in tests/golden/.
 *
 * All citations are file:line into the reference tree.
 */
#ifndef PM_ORACLE_H
#define PM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as pm_hit in include/pm_gpu.h. */
typedef struct {
  int64_t  end;    /* stream index after the last matched char (cp.pos() at emission) */
  uint32_t pid;    /* pattern id as given to add_pattern (1-based in the CLIs) */
  uint8_t  k;      /* errors reported by the engine */
  uint8_t  pad[3];
} pmo_hit;

/* The CharacterProducer view an engine needs (char_io.h:18-71): the byte stream as getnch()
 * returns it, the code->char map ch() and the char->code map nch() (-1 when absent). */
typedef struct {
  const uint8_t *codes;
  int64_t        n;
  int            size;       /* cp.size(): 256 for raw streams, |table| for Normalized<> */
  uint8_t        ch[256];
  int32_t        nch[256];
} pmo_text;

/* Engine selectors: the reference's -N numbers (select.cc:197-265) plus two extras. */
enum {
  PMO_AUTO = 0,
  PMO_KT_LIST = 1, PMO_KT_DNA = 2, PMO_KT_JTABLE = 3,   /* keyword_tree<...>: same hit set */
  PMO_SHIFT_AND = 4,
  PMO_FILTER_BITVEC = 5,
  PMO_EXACT_BASES_KT = 8, PMO_EXACT_BASES_SA = 10,
  PMO_EXACT_HALVES_KT = 12, PMO_EXACT_HALVES_SA = 14,
  PMO_SHIFT_AND_INEXACT = 100                             /* bare candidate generator */
};

typedef struct {
  int     engine;       /* PMO_* */
  int     k;            /* max errors (-k / -K) */
  int     indels;       /* 1: -k (edits), 0: -K (substitutions only) */
  int     wildcards;    /* -w / -W: not restated yet, must be 0 */
  int     text_n;
  uint8_t eos;          /* raw end-of-sequence char, '\n' in the CLIs */
} pmo_config;

/* Fill a pmo_text for a raw byte stream (MapFileChars: identity maps, size 256). */
void pmo_text_raw(pmo_text *t, const uint8_t *bytes, int64_t n);
/* Fill a pmo_text for a normalized stream; table = contents of the .tbl file (char_io.t:216). */
void pmo_text_normalized(pmo_text *t, const uint8_t *codes, int64_t n, const uint8_t *table, int table_len);

/* pick_pattern_index's automatic choice (select.cc:19-148), NOPRIMEGEN build.
 * Returns the -N number, or -1 for the "edits >= inexact bases" fatal error (select.cc:87-90). */
int pmo_pick_engine(const pmo_text *t, int k, int indels, int wildcards, int npat,
                    const int32_t *patlen, const int32_t *esb, const int32_t *eeb);

/* Run one engine over the whole text.  Patterns are patbuf[patoff[i] .. patoff[i+1]) with ids
 * ids[i] (NULL: 1..npat).  esb/eeb may be NULL (all zero).  Hits are returned in a malloc'ed
 * array (caller frees with pmo_free) in the reference engine's emission order where that is
 * defined; callers compare sorted sets (testscript.sh:353-422).  Returns 0, or <0 on error. */
int pmo_find_all(const pmo_text *t, const pmo_config *cfg,
                 const char *patbuf, const int64_t *patoff, int npat, const uint32_t *ids,
                 const int32_t *esb, const int32_t *eeb,
                 pmo_hit **out, size_t *nout);

/* primer_match's per-hit re-alignment (primer_match.cc:1135-1156): for an engine hit at `end`,
 * returns the alignment's start, end and editdist() (INT32_MAX = constraint violation /
 * "bogus hit").  k==0 follows exact_alignment (pattern_alignment.cc:29-43). */
typedef struct { int64_t start, end; int32_t editdist; int32_t value; } pmo_alignment;
int pmo_cli_align(const pmo_text *t, const pmo_config *cfg, const char *pat, int patlen,
                  int esb, int eeb, int64_t end, pmo_alignment *out);

/* editdist_alignment::align (pattern_alignment.cc:117-705) exposed for direct fuzzing.
 * Returns 1/0 like align(); out->value is meaningful only when the DP completed. */
int pmo_editdist_align(const pmo_text *t, const char *pat, int patlen, int64_t end, int64_t end2,
                       int k, uint8_t eos, int indels, int lconst, int rconst, int yesno,
                       pmo_alignment *out);

void pmo_free(void *p);

/* Single-threaded timing helper for bench.py's cpu_baseline: runs pmo_find_all and returns
 * wall seconds (hits discarded, count returned through nout). */
double pmo_time_find_all(const pmo_text *t, const pmo_config *cfg,
                         const char *patbuf, const int64_t *patoff, int npat, size_t *nout);

#ifdef __cplusplus
}
#endif
#endif

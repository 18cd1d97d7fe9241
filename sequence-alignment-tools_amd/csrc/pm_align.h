// pm_align.h -- verify-stage dynamic programs of the product path (host side, sparse work).
//
// These are the product's own implementations of the two banded DPs the reference runs per
// candidate: editdist_alignment::align (reference pattern_alignment.cc:117-705) and
// primer_alignment::global_align with its l/r extenders (reference primer_alignment.cc:10-463,
// 568-728).  They decide the reported end position and value, so every tie-break of the
// reference is honoured; parity is checked against oracle/ and tests/golden by the tests.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pm {

struct AlignResult {
  bool    ok = false;      // align() return value
  int64_t start = 0;       // stream index of first aligned text char
  int64_t end = 0;         // stream index after the last aligned text char
  int     value = 0;       // DP value at the chosen cell
  int     editdist = 0;    // subs+ins+dels along the traceback; INT32_MAX on constraint violation
};

struct AlignParams {
  int     k = 0;
  bool    indels = true;
  uint8_t eos = '\n';
  bool    wc = false, tn = false;   // -w / -W: IUPAC-compatible characters cost nothing; text N only with tn
};

// Reusable scratch so the verify loop does not allocate per candidate.
struct AlignScratch {
  std::vector<uint32_t> dp;
  std::vector<uint16_t> fl;
};

// Windowed right-to-left DP with floating right end.  `win` holds the raw characters of stream
// range [win_start, end2); the caller guarantees win_start == max(0, end - L - k).
AlignResult editdist_align(const uint8_t *win, int64_t win_start, int64_t end, int64_t end2,
                           const char *pat, int L, int lconst, int rconst,
                           const AlignParams &prm, AlignScratch &scr, std::string *ops = nullptr);

// Seed extension, yes/no form.  lmatch: left part (len1 chars) matched exactly ending at end1,
// `win` = chars of [end1, end1 + len2 + k).  rmatch: right part (len2 chars) matched exactly
// ending at end2, `win` = chars of [win_start, end2 - len2) with
// win_start = max(0, end2 - (len1+len2+k)).
bool lmatch_extend(const uint8_t *win, int64_t end1, int len1, const char *p2, int len2,
                   int esb, int eeb, const AlignParams &prm, AlignScratch &scr,
                   int64_t *end, int *value);
bool rmatch_extend(const uint8_t *win, int winlen, int64_t end2, const char *p1, int len1, int len2,
                   int esb, int eeb, const AlignParams &prm, AlignScratch &scr,
                   int64_t *end, int *value);

}  // namespace pm

// pm_extend.hip -- seed extension kernel: primer_alignment's banded global DP on the GPU.
//
// exact_halves (reference exact_halves.cc:140-190) extends every exact seed of a half pattern with
// primer_alignment_lmatch / _rmatch in yes/no form (primer_alignment.cc:568-617, 651-704), i.e.
// global_align (primer_alignment.cc:10-299): a DP of the partner half against len+k stream
// characters, band |t-p| <= k, unit costs, constraint violations priced at 5k+1, the end column
// chosen by the reference's tie rule (:253-280).  One lane runs one seed: the partner half is
// <= 16 characters and the band 2k+1 <= 5 cells wide, so two rolling rows live in registers and
// the whole DP is a few hundred integer ops; no traceback is needed in the yes/no form.
// Input: seed records {end = seed position, pid = inner id (2j+1 left half, 2j+2 right half)}.
// Output: for every seed whose extension succeeds one record
//   {end = seed position, pid = inner id, k = DP value, aux[0] = hit end - seed position}
// -- the order-dependent dedup of exact_halves.cc:163,178 stays on the host (it is a sequential
// rule over at most a few 10^5 records).
#include "pm_internal.h"

namespace pm {

namespace {

constexpr int XW = 8;          // band storage: 2k+2 <= 8 for k <= 3

struct ExtendArgs {
  const uint8_t *text;
  int64_t n;
  const pm_hit *seeds;
  unsigned long long nseeds;
  const uint8_t *half_codes;   // [2N][16] stream codes of the half patterns
  const uint8_t *half_len;     // [2N]
  const int32_t *esb, *eeb;    // [N] exact_start_bases / exact_end_bases of the whole patterns
  int k, eos_code;
  pm_hit *out;
  unsigned long long *counter;
  unsigned long long cap;
};

enum : uint32_t { F_EQ = 2, F_SUB = 8, F_INS = 16, F_DEL = 32, F_VIOL = 64 };

// global_align, yes/no form, indels on.  tc(t) = t-th text character in alignment direction
// (1-based), pc(p) = p-th pattern character.  Returns true and (matchlen,value).
template <typename TC, typename PC>
__device__ __forceinline__ bool global_align_dev(TC tc, int textlen, PC pc, int L, int lbexact, int rbexact,
                                                 int k, int eos, int *matchlen, int *value) {
  const int viol = 5 * k + 1, b = k;
  int prev[XW], cur[XW];
  uint32_t curf[XW];
#pragma unroll
  for (int o = 0; o < XW; ++o) { prev[o] = viol; cur[o] = viol; curf[o] = F_VIOL; }
  // row 0: cell t at o = t + b                                   (primer_alignment.cc:57, 88-112)
  prev[b] = 0;
  {
    const int ub = b < textlen ? b : textlen;
    int run = 0;
#pragma unroll
    for (int t = 1; t <= XW - 1; ++t) {
      if (t > ub) break;
      const int c = tc(t);
      if (0 < lbexact || 0 >= rbexact || c == eos) run = viol; else run = run + 1;
      if (t + b < XW) prev[t + b] = run;
    }
  }
  // column 0 (t = 0) of row p sits at o = b - p                    (primer_alignment.cc:64-82)
  int col0 = 0;
  for (int p = 1; p <= L; ++p) {
    const int lb = p - b > 1 ? p - b : 1, ub = p + b < textlen ? p + b : textlen;
    const int pch = pc(p);
    if (p <= b) col0 = (p < lbexact || p >= rbexact) ? viol : col0 + 1;
    int rowmin = viol;
#pragma unroll
    for (int o = 0; o < XW - 1; ++o) {
      const int t = p - b + o;
      int v = viol; uint32_t ac = F_VIOL;
      if (t >= lb && t <= ub) {
        const int c = tc(t);
        // diagonal: row p-1, same o; at t == 1 that is column 0 of row p-1 (or the origin)
        const int diag = prev[o];
        if (c == pch) { v = diag; ac = F_EQ; }
        else if (c == eos || p <= lbexact || p >= rbexact) { v = viol; ac = F_VIOL; }
        else { v = diag + 1; ac = F_SUB; }
        int v1; uint32_t ac1;
        if (c == eos || t <= lb || p < lbexact || p >= rbexact) { v1 = viol; ac1 = F_VIOL; }
        else { v1 = (o > 0 ? cur[o - 1] : viol) + 1; ac1 = F_INS; }
        if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
        if (t >= ub || p <= lbexact || p >= rbexact) { v1 = viol; ac1 = F_VIOL; }
        else { v1 = prev[o + 1] + 1; ac1 = F_DEL; }
        if (v1 < v) { v = v1; ac = ac1; } else if (v1 == v) ac |= ac1;
        rowmin = v < rowmin ? v : rowmin;
      } else if (t == 0 && p <= b) { v = col0; ac = (p < lbexact || p >= rbexact) ? F_VIOL : F_DEL; }
      cur[o] = v; curf[o] = ac;
    }
    if (rowmin > k) return false;                                 // :243-247
#pragma unroll
    for (int o = 0; o < XW; ++o) prev[o] = cur[o];
  }
  // end column (:252-280): start at L-b, later columns win on "<", or on "<=" when reached diagonally
  int best = L - b; if (textlen < best) best = textlen; if (best < 0) best = 0;
  int bestval = viol;
  {
    const int ub = L + b < textlen ? L + b : textlen;
    bool first = true;
#pragma unroll
    for (int o = 0; o < XW - 1; ++o) {
      const int t = L - b + o;
      if (t < best || t > ub) continue;
      const int v = cur[o];
      if (first) { bestval = v; best = t; first = false; }
      else if (v < bestval || (v <= bestval && (curf[o] & (F_EQ | F_SUB)))) { bestval = v; best = t; }
    }
  }
  if (best < L - b || best > L + b) return false;                 // :285-289
  *matchlen = best; *value = bestval;
  return true;
}

__global__ void pm_seed_extend(ExtendArgs a) {
  const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  if (i >= a.nseeds) return;
  const pm_hit s = a.seeds[i];
  const uint32_t hid = s.pid;                                     // 1-based inner id
  if (hid == PM_SEED_HOLE) return;                                // unused slot of a reserved output block (pm_seed.hip)
  const uint32_t j = (hid - 1) >> 1;
  const bool left = (hid & 1u) != 0;
  const int len1 = a.half_len[2 * j], len2 = a.half_len[2 * j + 1];
  const int esb = a.esb[j], eeb = a.eeb[j];
  const int k = a.k;
  int matchlen = 0, value = 0;
  bool ok;
  if (left) {
    // primer_alignment_lmatch (:568-617): partner = right half, text = [end1, end1+len2+k) forwards;
    // lmatch_-len(p1) is unsigned arithmetic landing in an int (:608): negative = no constraint
    const uint8_t *pat = a.half_codes + (size_t)(2 * j + 1) * 16;
    const int64_t base = s.end;
    const uint32_t lm = (uint32_t)esb - (uint32_t)len1;
    int lbexact = 0, rbexact = len2 + 1;
    if (lm > 0) lbexact = (int)lm;
    if (eeb > 0) rbexact = len2 + 1 - eeb;
    auto tc = [&](int t) -> int { const int64_t q = base + t - 1; return q < a.n ? a.text[q] : 0; };
    auto pc = [&](int p) -> int { return pat[p - 1]; };
    ok = global_align_dev(tc, len2 + k, pc, len2, lbexact, rbexact, k, a.eos_code, &matchlen, &value);
  } else {
    // primer_alignment_rmatch (:651-704): partner = left half, text = [textstart, end2-len2) backwards
    const uint8_t *pat = a.half_codes + (size_t)(2 * j) * 16;
    const int plen = len1 + len2 + k;
    const int64_t textstart = s.end > (int64_t)plen ? s.end - plen : 0;
    const int buflen = (int)(s.end - len2 - textstart);
    const int64_t last = s.end - len2;                            // one past the window
    const uint32_t rm = (uint32_t)eeb - (uint32_t)len2;
    int lbexact = 0, rbexact = len1 + 1;
    if (esb > 0) rbexact = len1 + 1 - esb;
    if (rm > 0) lbexact = (int)rm;
    auto tc = [&](int t) -> int { const int64_t q = last - t; return (q >= 0 && q < a.n) ? a.text[q] : 0; };
    auto pc = [&](int p) -> int { return pat[len1 - p]; };
    ok = buflen >= 0 && global_align_dev(tc, buflen, pc, len1, lbexact, rbexact, k, a.eos_code, &matchlen, &value);
    matchlen = 0;
  }
  if (!ok) return;
  const unsigned long long o = atomicAdd(a.counter, 1ull);
  if (o < a.cap) {
    pm_hit h;
    h.end = s.end; h.pid = hid; h.k = (uint8_t)value;
    h.aux[0] = (uint8_t)(left ? matchlen : 0); h.aux[1] = 1; h.aux[2] = 0;
    a.out[o] = h;
  }
}

}  // namespace

hipError_t extend_seeds(const uint8_t *d_text, int64_t n, const pm_hit *d_seeds, size_t nseeds,
                        const uint8_t *d_half_codes, const uint8_t *d_half_len, const int32_t *d_esb, const int32_t *d_eeb,
                        int k, int eos_code, pm_hit *d_out, unsigned long long *d_counter, size_t cap, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess || nseeds == 0) return e;
  ExtendArgs a;
  a.text = d_text; a.n = n; a.seeds = d_seeds; a.nseeds = nseeds; a.half_codes = d_half_codes; a.half_len = d_half_len;
  a.esb = d_esb; a.eeb = d_eeb; a.k = k; a.eos_code = eos_code; a.out = d_out; a.counter = d_counter; a.cap = cap;
  const int threads = 256;
  hipLaunchKernelGGL(pm_seed_extend, dim3((unsigned)((nseeds + threads - 1) / threads)), dim3(threads), 0, st, a);
  return hipGetLastError();
}

}  // namespace pm

// pm_cluster.hip -- device side of filter_bitvec's clustering for substitution-only search.
//
// filter_bitvec (reference filter_bitvec.cc:88-177) sorts the k-error candidates by position,
// chains same-pattern candidates that lie within 2k+1 of each other and verifies each chain once.
// For -K without exact-base constraints the verify reduces to "smallest level, left-most end"
// (DESIGN.md section 2), so the whole stage is a sort + a segmented pass:
//   1. pack every record into one 64-bit key  pattern(22) | end(40) | level(2)
//   2. rocPRIM/hipCUB radix sort of the keys (no payload)
//   3. one thread per key: a key whose predecessor is another pattern or more than 2k+1 away
//      starts a cluster; that thread walks its cluster (a handful of keys except on tandem
//      repeats), keeps the smallest level / left-most end, and appends one final hit.
// Clusters that could still grow (end of the scanned range) or that start inside the first L
// characters of the stream (the only place where the DP is not equivalent) are handed back
// unchanged for the host stage.
#include <hipcub/hipcub.hpp>

#include "pm_internal.h"

namespace pm {

namespace {

// One output slot for every lane that is active here, one atomic per wave: a counter shared by the
// whole grid serialises same-address atomics (~2.5 ns each, 9 ms for the 3.7 M clusters of a 3 Gbp -k 2 scan).
__device__ __forceinline__ unsigned long long wave_reserve_slot(unsigned long long *counter) {
  const unsigned long long bal = __ballot(1);
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(bal));
  base = __shfl(base, leader);
  return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
}

__global__ void pm_cluster_pack(const pm_hit *in, size_t n, uint64_t *keys) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  keys[i] = ((uint64_t)h.pid << 42) | (((uint64_t)h.end & 0xffffffffffull) << 2) | (uint64_t)(h.k & 3u);
}

// Position-sharded scans (OwnedRange, pm_internal.h): a chain the guard edge may have cut is an
// error when its hit could land in the owned range, and simply another shard's otherwise.
// Returns true when the chain [first, last] is to be processed here.
__device__ __forceinline__ bool chain_owned(const OwnedRange &own, int64_t first, int64_t last, int win, unsigned long long *cut_count) {
  if (!own.on) return true;
  const bool cut_l = own.guard_lo > 0 && first - win <= own.guard_lo;     // a predecessor may hide at <= guard_lo
  const bool cut_r = last > own.guard_hi - win;                           // a successor may follow guard_hi
  if ((cut_l && last > own.own_lo) || (cut_r && first <= own.own_hi)) { atomicAdd(cut_count, 1ull); return false; }
  return last > own.own_lo && first <= own.own_hi;                        // the hit ends inside [first, last]
}
__device__ __forceinline__ bool hit_owned(const OwnedRange &own, int64_t end) {
  return !own.on || (end > own.own_lo && end <= own.own_hi);
}

// One thread per sorted key; a chain's head walks it.  The final hits leave block by block: one atomic on the shared
// counter per 256 keys (one per wave -- 15k same-address atomics for the 10^6 hits of a 3 Gbp -K 2 scan -- took this
// kernel 0.19 ms; block by block: 0.05 ms, profiles/r03_kernel_stats_K2.csv).
__global__ __launch_bounds__(256) void pm_cluster_reduce(const uint64_t *keys, size_t n, int win, int64_t scanned_to, int last, int invalid_level,
                                                         const uint8_t *pat_len, const uint32_t *pat_id, OwnedRange own,
                                                         pm_hit *out, unsigned long long *out_count,
                                                         pm_hit *left, unsigned long long *left_count) {
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  pm_hit h;
  h.end = 0; h.pid = 0; h.k = 0; h.aux[0] = h.aux[1] = h.aux[2] = 0;
  // true: h is a final hit of this shard
  auto head_of_chain = [&]() -> bool {
    if (i >= n) return false;
    const uint64_t key = keys[i];
    const uint32_t pid = (uint32_t)(key >> 42);
    const int64_t end = (int64_t)((key >> 2) & 0xffffffffffull);
    if (i > 0) {
      const uint64_t pk = keys[i - 1];
      if ((uint32_t)(pk >> 42) == pid && end - (int64_t)((pk >> 2) & 0xffffffffffull) <= win) return false;   // not a head
    }
    int best = (int)(key & 3u);
    int64_t best_end = end, prev = end;
    size_t j = i + 1;
    for (; j < n; ++j) {
      const uint64_t nk = keys[j];
      const int64_t ne = (int64_t)((nk >> 2) & 0xffffffffffull);
      if ((uint32_t)(nk >> 42) != pid || ne - prev > win) break;
      const int lv = (int)(nk & 3u);
      if (lv < best) { best = lv; best_end = ne; }
      prev = ne;
    }
    if (!chain_owned(own, end, prev, win, out_count + 2)) return false;
    const bool incomplete = !last && scanned_to < prev + win;        // filter_bitvec.cc:118-121
    const bool needs_dp = end < (int64_t)pat_len[pid - 1];            // window not fully inside the stream
    if (incomplete || needs_dp) {
      const unsigned long long o = atomicAdd(left_count, (unsigned long long)(j - i));
      for (size_t t = i; t < j; ++t) {
        pm_hit x;
        x.pid = pid; x.end = (int64_t)((keys[t] >> 2) & 0xffffffffffull); x.k = (uint8_t)(keys[t] & 3u);
        x.aux[0] = x.aux[1] = x.aux[2] = 0;
        left[o + (t - i)] = x;
      }
      return false;
    }
    // (invalid_level: candidates whose substitutions touch an exact zone take part in the chain but
    // cannot be its hit -- the reference's constrained verify fails on them, pattern_alignment.cc:320-323)
    if (best == invalid_level || !hit_owned(own, best_end)) return false;
    h.pid = pat_id[pid - 1]; h.end = best_end; h.k = (uint8_t)best;
    return true;
  };
  const bool emit = head_of_chain();
  const unsigned long long bal = __ballot(emit);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = tot ? atomicAdd(out_count, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  if (emit) {
    unsigned long long o = s_base + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) o += s_cnt[w];
    out[o] = h;
  }
}

// ---- duplicate removal for the edit-distance seed plan --------------------------------------------
// Several seeds (combos, displacement patterns) lead to the same (pattern, end) candidate and the
// records come in blocks with unused slots: pack (holes get the largest key), sort, keep the first
// of every run of equal keys.
constexpr uint64_t DEDUP_HOLE = ~0ull;

__global__ void pm_dedup_pack(const pm_hit *in, size_t n, uint64_t *keys) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  keys[i] = h.pid == PM_SEED_HOLE ? DEDUP_HOLE
                                  : ((uint64_t)h.pid << 42) | (((uint64_t)h.end & 0xffffffffffull) << 2) | (uint64_t)(h.k & 3u);
}

// keys sorted: equal (pattern, end) are adjacent and the smallest level comes first.  The unique records leave block
// by block (one atomic per 256 keys; one per wave took 1.04 ms for the 7·10^6 records of a 3 Gbp -k 2 scan).
__global__ __launch_bounds__(256) void pm_dedup_unpack(const uint64_t *keys, size_t n, pm_hit *out, unsigned long long *count) {
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  bool head = false;
  uint64_t key = 0;
  if (i < n) {
    key = keys[i];
    head = key != DEDUP_HOLE && (i == 0 || (keys[i - 1] >> 2) != (key >> 2));
  }
  const unsigned long long bal = __ballot(head);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  if (head) {
    unsigned long long o = s_base + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) o += s_cnt[w];
    pm_hit h;
    h.pid = (uint32_t)(key >> 42); h.end = (int64_t)((key >> 2) & 0xffffffffffull); h.k = (uint8_t)(key & 3u);
    h.aux[0] = h.aux[1] = h.aux[2] = 0;
    out[o] = h;
  }
}

// ---- filter_bitvec with edits: clustering and the banded DP per cluster on the device ------------
// editdist_alignment::align (reference pattern_alignment.cc:117-705) as pm_align.cpp restates it,
// on stream codes (the patterns of the seed family are A,C,G,T, so code equality is character
// equality), band cells in thread-private memory as (value, flags) bytes.  Returns true and
// (*end, *value) when the alignment's value is <= k.
constexpr int DP_MAXL = 32, DP_MAXDELTA = 10, DP_MAXW = DP_MAXDELTA + 2 * 3 + 2;
enum : uint8_t { D_EQ = 2, D_SUB = 8, D_INS = 16, D_DEL = 32, D_VIOL = 64, D_END = 128 };

// The stream window (<= DP_MAXL + 3 + DP_MAXDELTA + 1 characters) and the pattern are copied into LDS first (byte i of
// thread t at i * DP_THREADS + t): read cell by cell from global memory, a dependent byte load per cell, they were
// most of the kernel's time.
constexpr int DP_THREADS = 256, DP_WIN = 48;
__device__ bool device_editdist(const uint8_t *text, int64_t n, int64_t end, int64_t end2, const uint8_t *pat, int L,
                                int lconst, int rconst, int k, bool indels, int eos, int64_t *out_end, int *out_value,
                                uint8_t *swin, uint8_t *spat) {
  uint8_t dp[(DP_MAXL + 1) * DP_MAXW], fl[(DP_MAXL + 1) * DP_MAXW];
  const int viol = 5 * k + 1, b = indels ? k : 0;
  int64_t ws = 0;
  if (end > (int64_t)L + k) ws = end - L - k;                      // :137-139
  const int buflen = (int)(end2 - ws), delta = (int)(end2 - end);
  const int W = delta + 2 * b + 2;
  auto at = [&](int p, int t) { return p * W + (t - (p - b)); };
  for (int i = 0; i <= buflen && i < DP_WIN; ++i) { const int64_t q = ws + i; swin[i * DP_THREADS] = q >= 0 && q < n ? text[q] : (uint8_t)0; }
  for (int i = 0; i < L; ++i) spat[i * DP_THREADS] = pat[i];
  pat = nullptr;
  auto tch = [&](int t) -> int { return (int)swin[(buflen - t) * DP_THREADS]; };   // win[buflen - t]
  int lbexact = 0, rbexact = L + 1;                                // :230-233
  if (lconst > 0) rbexact = L + 1 - lconst;
  if (rconst > 0) lbexact = rconst;
  dp[at(0, 0)] = 0; fl[at(0, 0)] = D_END;
  for (int p = 1, ub = b < L ? b : L; p <= ub; ++p) {              // column 0 (:253-268)
    const int i = at(p, 0);
    if (!indels || p < lbexact || p >= rbexact) { dp[i] = (uint8_t)viol; fl[i] = D_VIOL; }
    else { dp[i] = (uint8_t)(dp[at(p - 1, 0)] + 1); fl[i] = D_DEL; }
  }
  for (int t = 1, ub = buflen < delta + b ? buflen : delta + b; t <= ub; ++t) {   // row 0 (:276-294)
    const int i = at(0, t);
    if (t <= delta) { dp[i] = 0; fl[i] = D_END; }
    else if (!indels || lbexact > 0) { dp[i] = (uint8_t)viol; fl[i] = D_VIOL; }
    else { dp[i] = (uint8_t)(dp[at(0, t - 1)] + 1); fl[i] = D_INS; }
  }
  for (int p = 1; p <= L; ++p) {                                   // :296-437
    const int lb = p - b > 1 ? p - b : 1, ub = buflen < p + delta + b ? buflen : p + delta + b;
    const int pc = spat[(L - p) * DP_THREADS];
    const bool zone_sub = (p <= lbexact || p >= rbexact), zone_ins = (p < lbexact || p >= rbexact);
    int rowmin = viol;
    for (int t = lb; t <= ub; ++t) {
      const int tc = tch(t);
      int v, v1; uint8_t ac;
      if (tc == pc) { v = dp[at(p - 1, t - 1)]; ac = D_EQ; }
      else if (tc == eos || zone_sub) { v = viol; ac = D_VIOL; }
      else { v = dp[at(p - 1, t - 1)] + 1; ac = D_SUB; }
      if (tc == eos || !indels || t <= lb || zone_ins) {
        if (viol < v) { v = viol; ac = D_VIOL; }
      } else {
        v1 = dp[at(p, t - 1)] + 1;
        if (v1 < v) { v = v1; ac = D_INS; } else if (v1 == v) ac |= D_INS;
      }
      if (!indels || t >= ub || zone_sub) {
        if (viol < v) { v = viol; ac = D_VIOL; }
      } else {
        v1 = dp[at(p - 1, t)] + 1;
        if (v1 < v) { v = v1; ac = D_DEL; } else if (v1 == v) ac |= D_DEL;
      }
      const int i = at(p, t);
      dp[i] = (uint8_t)v; fl[i] = ac;
      rowmin = rowmin < v ? rowmin : v;
    }
    if (rowmin > k) return false;                                  // :425-436
  }
  int best = L - b < buflen ? L - b : buflen;                      // :443-475
  if (best < 0) best = 0;
  int bestval = dp[at(L, best)];
  for (int c = best + 1, ub = buflen < L + delta + b ? buflen : L + delta + b; c <= ub; ++c) {
    const int v = dp[at(L, c)];
    if (v < bestval || (v <= bestval && (fl[at(L, c)] & (D_EQ | D_SUB)))) { bestval = v; best = c; }
  }
  int p = L, t = best;
  if (t < p - b || t > p + b + delta) return false;                // :482-490
  int last = 0;                                                    // 0 none, 1 eq, 2 sub, 3 ins, 4 del
  for (int guard = 0; guard < 4 * (DP_MAXL + DP_MAXW); ++guard) {  // traceback (:514-590): only the end column matters here
    const uint8_t ac = fl[at(p, t)];
    if (ac & D_END) break;
    const bool match = ac & (D_EQ | D_SUB), sub = ac & D_SUB, ins = ac & D_INS, del = ac & D_DEL;
    if (match && !((last == 3 && ins) || (last == 4 && del))) {
      --p; --t;
      if ((ac & D_EQ) && !(last == 2 && sub)) last = 1;
      else if (sub) last = 2;
    } else if (del) { --p; last = 4; }
    else if (ins) { --t; last = 3; }
    else if (ac & D_VIOL) { p = 0; t = 0; break; }
    else return false;
  }
  *out_end = end2 - t;                                             // :603-610
  *out_value = bestval;
  return bestval <= k;
}

__global__ void pm_cluster_dp(const uint64_t *keys, size_t n, int k, int indels, int64_t scanned_to, int last,
                              const uint8_t *text, int64_t ntext, int eos,
                              const uint8_t *pat_codes, const uint8_t *pat_len, const int32_t *esb, const int32_t *eeb,
                              const uint32_t *pat_id, OwnedRange own, pm_hit *out, unsigned long long *out_count,
                              pm_hit *left, unsigned long long *left_count) {
  __shared__ uint8_t swin[DP_WIN * DP_THREADS], spat[DP_MAXL * DP_THREADS];   // device_editdist's copies of window and pattern
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == DEDUP_HOLE) return;
  const int win = 2 * k + 1;
  const uint32_t pid = (uint32_t)(key >> 42);
  const int64_t end = (int64_t)((key >> 2) & 0xffffffffffull);
  if (i > 0) {
    const uint64_t pk = keys[i - 1];
    if ((uint32_t)(pk >> 42) == pid && end - (int64_t)((pk >> 2) & 0xffffffffffull) <= win) return;   // not a cluster head
  }
  int64_t prev = end;
  size_t j = i + 1;
  for (; j < n; ++j) {                                             // chain: next candidate within 2k+1 (filter_bitvec.cc:103-116)
    const uint64_t nk = keys[j];
    const int64_t ne = (int64_t)((nk >> 2) & 0xffffffffffull);
    if ((uint32_t)(nk >> 42) != pid || ne - prev > win) break;
    prev = ne;
  }
  if (!chain_owned(own, end, prev, win, out_count + 2)) return;
  const int L = pat_len[pid - 1];
  const bool incomplete = !last && scanned_to < prev + win;        // :118-121
  if (incomplete || prev - end > DP_MAXDELTA || L > DP_MAXL || k > 3) {   // host stage: may still grow / long repeat cluster
    for (size_t t = i; t < j; ++t) {
      if (t > i && (keys[t] >> 2) == (keys[t - 1] >> 2)) continue;  // duplicate
      const unsigned long long o = atomicAdd(left_count, 1ull);
      pm_hit h;
      h.pid = pid; h.end = (int64_t)((keys[t] >> 2) & 0xffffffffffull); h.k = (uint8_t)(keys[t] & 3u);
      h.aux[0] = h.aux[1] = h.aux[2] = 0;
      left[o] = h;
    }
    return;
  }
  int64_t rend = 0; int rval = 0;
  if (device_editdist(text, ntext, end, prev, pat_codes + (size_t)(pid - 1) * 32, L, esb[pid - 1], eeb[pid - 1], k, indels != 0, eos, &rend, &rval, swin + threadIdx.x, spat + threadIdx.x) &&
      hit_owned(own, rend)) {
    const unsigned long long o = wave_reserve_slot(out_count);
    pm_hit h;
    h.pid = pat_id[pid - 1]; h.end = rend; h.k = (uint8_t)rval; h.aux[0] = h.aux[1] = h.aux[2] = 0;
    out[o] = h;
  }
}

}  // namespace

// Clustering + one DP per cluster for filter_bitvec with edits.  Records (two arrays: this range's and
// the ones an earlier range left undecided) in any order (duplicates and
// holes allowed); finals to d_out / d_counts[0], records the host stage must look at to d_left / d_counts[1],
// chains cut by the guard edge of an owned range counted in d_counts[2].
hipError_t cluster_dp_device(const pm_hit *d_in, size_t n1, const pm_hit *d_in2, size_t n2, int k, bool indels, int64_t scanned_to, bool last,
                             const uint8_t *d_text, int64_t ntext, int eos_code,
                             const uint8_t *d_pat_codes, const uint8_t *d_pat_len, const int32_t *d_esb, const int32_t *d_eeb,
                             const uint32_t *d_pat_id, const OwnedRange &own, uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                             pm_hit *d_out, pm_hit *d_left, unsigned long long *d_counts, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), st);
  const size_t n = n1 + n2;
  if (e != hipSuccess || n == 0) return e;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads);
  if (n1) hipLaunchKernelGGL(pm_dedup_pack, dim3((unsigned)((n1 + threads - 1) / threads)), dim3(threads), 0, st, d_in, n1, d_keys);
  if (n2) hipLaunchKernelGGL(pm_dedup_pack, dim3((unsigned)((n2 + threads - 1) / threads)), dim3(threads), 0, st, d_in2, n2, d_keys + n1);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys_alt, (int)n, 0, 64, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_cluster_dp, dim3(blocks), dim3(threads), 0, st, d_keys_alt, n, k, indels ? 1 : 0, scanned_to, last ? 1 : 0,
                     d_text, ntext, eos_code, d_pat_codes, d_pat_len, d_esb, d_eeb, d_pat_id, own, d_out, d_counts, d_left, d_counts + 1);
  return hipGetLastError();
}

// d_out may alias d_in (the keys are a copy).  *d_count receives the number of unique records.
hipError_t dedup_device(const pm_hit *d_in, size_t n, uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                        pm_hit *d_out, unsigned long long *d_count, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess || n == 0) return e;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads);
  hipLaunchKernelGGL(pm_dedup_pack, dim3(blocks), dim3(threads), 0, st, d_in, n, d_keys);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys_alt, (int)n, 0, 64, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_dedup_unpack, dim3(blocks), dim3(threads), 0, st, d_keys_alt, n, d_out, d_count);
  return hipGetLastError();
}

// ---- exact_halves: the per-pattern "end beyond the last kept end" rule on the device -------------
// exact_halves.cc:114-118,142,163,178: seed hits are visited in (position asc, inner id desc) order
// and a hit is kept when its end exceeds the pattern's last kept end (by more than 2k with edits).
// The rule only couples hits of one pattern, so: one sort key per seed
//   pattern(22) | seed position(40) | left half(1)     (right half = larger inner id = first)
// with (end - seed position, value) as the payload, a radix sort, and one thread per pattern
// walking its run.  Stateless: the caller guarantees a fresh engine state and a complete range.
namespace {

constexpr uint64_t HALVES_HOLE = ~0ull;

__device__ __forceinline__ uint64_t halves_key(uint32_t j, int64_t seedpos, bool left) {
  return ((uint64_t)j << 41) | (((uint64_t)seedpos & 0xffffffffffull) << 1) | (left ? 1u : 0u);
}

// whole-pattern Hamming candidates with clean-half flags (finalize_halves_flags): two seeds per record
__global__ void pm_halves_pack_flags(const pm_hit *in, size_t n, const uint8_t *pat_len, uint64_t *keys, uint32_t *vals) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  const int L = pat_len[h.pid - 1], len2 = L - L / 2;
  keys[2 * i] = (h.aux[0] & 1) ? halves_key(h.pid, h.end - len2, true) : HALVES_HOLE;
  vals[2 * i] = ((uint32_t)len2 << 8) | h.k;
  keys[2 * i + 1] = (h.aux[0] & 2) ? halves_key(h.pid, h.end, false) : HALVES_HOLE;
  vals[2 * i + 1] = h.k;
}

// extended half seeds (pm_extend.hip): {end = seed position, pid = inner id, k, aux[0] = end - position}
__global__ void pm_halves_pack_seeds(const pm_hit *in, size_t n, uint64_t *keys, uint32_t *vals) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  if (h.pid == PM_SEED_HOLE) { keys[i] = HALVES_HOLE; vals[i] = 0; return; }
  keys[i] = halves_key((h.pid + 1) / 2, h.end, (h.pid & 1u) != 0);
  vals[i] = ((uint32_t)h.aux[0] << 8) | h.k;
}

__global__ void pm_halves_rule(const uint64_t *keys, const uint32_t *vals, size_t n, int slack, const uint32_t *pat_id,
                               pm_hit *out, unsigned long long *out_count) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == HALVES_HOLE) return;
  const uint32_t j = (uint32_t)(key >> 41);
  if (i > 0 && (uint32_t)(keys[i - 1] >> 41) == j) return;         // not the first seed of its pattern
  int64_t lasthit = 0;                                             // fresh engine state (pm_api.cpp lasthit)
  for (size_t t = i; t < n; ++t) {
    const uint64_t kt = keys[t];
    if (kt == HALVES_HOLE || (uint32_t)(kt >> 41) != j) break;
    const uint32_t v = vals[t];
    const int64_t end = (int64_t)((kt >> 1) & 0xffffffffffull) + (int64_t)(v >> 8);
    if (end > lasthit + slack) {
      lasthit = end;
      const unsigned long long o = atomicAdd(out_count, 1ull);     // runs are short and heads are sparse: no aggregation to gain
      pm_hit h;
      h.pid = pat_id[j - 1]; h.end = end; h.k = (uint8_t)(v & 0xffu); h.aux[0] = h.aux[1] = h.aux[2] = 0;
      out[o] = h;
    }
  }
}

}  // namespace

size_t halves_temp_bytes(size_t n) {
  size_t bytes = 0;
  uint64_t *k = nullptr; uint32_t *v = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, v, v, (int)n);
  return bytes;
}

// flags != 0: records are whole-pattern candidates with clean-half flags (two seeds each, d_keys /
// d_vals need 2n slots); otherwise extended half seeds.  Finals to d_out, their number to d_counts[0].
hipError_t halves_rule_device(const pm_hit *d_in, size_t n, bool flags, int slack, const uint8_t *d_pat_len, const uint32_t *d_pat_id,
                              uint64_t *d_keys, uint64_t *d_keys_alt, uint32_t *d_vals, uint32_t *d_vals_alt, void *d_temp, size_t temp_bytes,
                              pm_hit *d_out, unsigned long long *d_counts, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess || n == 0) return e;
  const int threads = 256;
  const size_t m = flags ? 2 * n : n;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads), mblocks = (unsigned)((m + threads - 1) / threads);
  if (flags) hipLaunchKernelGGL(pm_halves_pack_flags, dim3(blocks), dim3(threads), 0, st, d_in, n, d_pat_len, d_keys, d_vals);
  else hipLaunchKernelGGL(pm_halves_pack_seeds, dim3(blocks), dim3(threads), 0, st, d_in, n, d_keys, d_vals);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_alt, d_vals, d_vals_alt, (int)m, 0, 64, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_halves_rule, dim3(mblocks), dim3(threads), 0, st, d_keys_alt, d_vals_alt, m, slack, d_pat_id, d_out, d_counts);
  return hipGetLastError();
}

namespace {
// pass-through engines on one shard of a position-sharded scan: keep the records that end in the
// owned range (the same records, compacted; order unspecified)
__global__ void pm_owned_filter(const pm_hit *in, size_t n, OwnedRange own, pm_hit *out, unsigned long long *out_count) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  if (!hit_owned(own, h.end)) return;
  out[wave_reserve_slot(out_count)] = h;
}
}  // namespace

hipError_t owned_filter_device(const pm_hit *d_in, size_t n, const OwnedRange &own, pm_hit *d_out, unsigned long long *d_count, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess || n == 0) return e;
  hipLaunchKernelGGL(pm_owned_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, n, own, d_out, d_count);
  return hipGetLastError();
}

// ---- final hits in (end, pid, k) order (pm_scan: PatternMatch::find_patterns hands its hits out in stream order,
// primer_match.cc:1118-1121; the reference's callers sort by key, sortedvector.t:490-510) -------------------------
// A final hit is (end, pattern, k) and nothing else, so the whole record fits the sort key: end | pattern index | k,
// a keys-only radix sort (a pass over 16-byte values costs three times a keys-only pass at 10^5 records) and one kernel
// that turns keys back into records.  Needs pattern ids that grow with the pattern index (the reference's callers add
// ids 1..N in order, primer_match.cc:1105-1107): index order is then id order, and the index of an id is found by
// bisection in the id table.  The number of records may be known only on the device (*d_count, written by the kernel in
// front on the same stream): the sort then runs over the host's upper bound, the slots beyond *d_count get a key above
// every real one and stay behind the real records.
namespace {
__global__ void pm_final_pack(const pm_hit *in, const unsigned long long *d_count, size_t n_upper, const uint32_t *pat_id, uint32_t npat,
                              int idxbits, int keybits, uint64_t *keys) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n_upper) return;
  const size_t n = d_count ? (size_t)*d_count : n_upper;
  uint64_t key = 1ull << keybits;                                   // padding: above every real key
  if (i < n) {
    const pm_hit h = in[i];
    uint32_t lo = 0, hi = npat;                                     // first index with pat_id[index] >= h.pid
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pat_id[mid] < h.pid) lo = mid + 1; else hi = mid; }
    key = ((uint64_t)h.end << (idxbits + 2)) | ((uint64_t)lo << 2) | (uint64_t)(h.k & 3u);
  }
  keys[i] = key;
}

__global__ void pm_final_unpack(const uint64_t *keys, const unsigned long long *d_count, size_t n_upper, const uint32_t *pat_id, int idxbits, pm_hit *out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t n = d_count ? (size_t)*d_count : n_upper;
  if (i >= n) return;
  const uint64_t key = keys[i];
  pm_hit h;
  h.end = (int64_t)(key >> (idxbits + 2));
  h.pid = pat_id[(uint32_t)(key >> 2) & ((1u << idxbits) - 1u)];
  h.k = (uint8_t)(key & 3u);
  h.aux[0] = h.aux[1] = h.aux[2] = 0;                               // engine-private bytes of candidate records do not leave
  out[i] = h;
}
}  // namespace

// keybits = bits of the largest end + idxbits + 2 (<= 63: the caller checks); the sort's temporary storage is that of
// cluster_temp_bytes(n_upper)
hipError_t sort_final_device(const pm_hit *d_in, const unsigned long long *d_count, size_t n_upper, const uint32_t *d_pat_id, uint32_t npat,
                             int idxbits, int keybits, uint64_t *d_keys, uint64_t *d_keys_alt, pm_hit *d_out, void *d_temp, size_t temp_bytes, hipStream_t st) {
  if (n_upper == 0) return hipSuccess;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n_upper + threads - 1) / threads);
  hipLaunchKernelGGL(pm_final_pack, dim3(blocks), dim3(threads), 0, st, d_in, d_count, n_upper, d_pat_id, npat, idxbits, keybits, d_keys);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int top = d_count ? keybits + 1 : keybits;                  // (the padding key has bit `keybits` set)
  if ((e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys_alt, (int)n_upper, 0, top, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_final_unpack, dim3(blocks), dim3(threads), 0, st, d_keys_alt, d_count, n_upper, d_pat_id, idxbits, d_out);
  return hipGetLastError();
}

size_t cluster_temp_bytes(size_t n) {
  size_t bytes = 0;
  uint64_t *p = nullptr;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, p, p, (int)n);
  return bytes;
}

hipError_t cluster_device(const pm_hit *d_in, size_t n1, const pm_hit *d_in2, size_t n2, int k, int64_t scanned_to, bool last, int invalid_level,
                          const uint8_t *d_pat_len, const uint32_t *d_pat_id, const OwnedRange &own,
                          uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                          pm_hit *d_out, pm_hit *d_left, unsigned long long *d_counts, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), st);
  const size_t n = n1 + n2;
  if (e != hipSuccess || n == 0) return e;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads);
  if (n1) hipLaunchKernelGGL(pm_cluster_pack, dim3((unsigned)((n1 + threads - 1) / threads)), dim3(threads), 0, st, d_in, n1, d_keys);
  if (n2) hipLaunchKernelGGL(pm_cluster_pack, dim3((unsigned)((n2 + threads - 1) / threads)), dim3(threads), 0, st, d_in2, n2, d_keys + n1);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys_alt, (int)n, 0, 64, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_cluster_reduce, dim3(blocks), dim3(threads), 0, st, d_keys_alt, n, 2 * k + 1, scanned_to, last ? 1 : 0, invalid_level,
                     d_pat_len, d_pat_id, own, d_out, d_counts, d_left, d_counts + 1);
  return hipGetLastError();
}

}  // namespace pm

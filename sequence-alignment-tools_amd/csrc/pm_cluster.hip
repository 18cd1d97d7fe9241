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

__global__ void pm_cluster_pack(const pm_hit *in, size_t n, uint64_t *keys) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pm_hit h = in[i];
  keys[i] = ((uint64_t)h.pid << 42) | (((uint64_t)h.end & 0xffffffffffull) << 2) | (uint64_t)(h.k & 3u);
}

__global__ void pm_cluster_reduce(const uint64_t *keys, size_t n, int win, int64_t scanned_to, int last,
                                  const uint8_t *pat_len, const uint32_t *pat_id,
                                  pm_hit *out, unsigned long long *out_count,
                                  pm_hit *left, unsigned long long *left_count) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  const uint32_t pid = (uint32_t)(key >> 42);
  const int64_t end = (int64_t)((key >> 2) & 0xffffffffffull);
  if (i > 0) {
    const uint64_t pk = keys[i - 1];
    if ((uint32_t)(pk >> 42) == pid && end - (int64_t)((pk >> 2) & 0xffffffffffull) <= win) return;   // not a head
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
  const bool incomplete = !last && scanned_to < prev + win;        // filter_bitvec.cc:118-121
  const bool needs_dp = end < (int64_t)pat_len[pid - 1];            // window not fully inside the stream
  if (incomplete || needs_dp) {
    const unsigned long long o = atomicAdd(left_count, (unsigned long long)(j - i));
    for (size_t t = i; t < j; ++t) {
      pm_hit h;
      h.pid = pid; h.end = (int64_t)((keys[t] >> 2) & 0xffffffffffull); h.k = (uint8_t)(keys[t] & 3u);
      h.aux[0] = h.aux[1] = h.aux[2] = 0;
      left[o + (t - i)] = h;
    }
  } else {
    const unsigned long long o = atomicAdd(out_count, 1ull);
    pm_hit h;
    h.pid = pat_id[pid - 1]; h.end = best_end; h.k = (uint8_t)best; h.aux[0] = h.aux[1] = h.aux[2] = 0;
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

// keys sorted: equal (pattern, end) are adjacent and the smallest level comes first
__global__ void pm_dedup_unpack(const uint64_t *keys, size_t n, pm_hit *out, unsigned long long *count) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  bool head = false;
  uint64_t key = 0;
  if (i < n) {
    key = keys[i];
    head = key != DEDUP_HOLE && (i == 0 || (keys[i - 1] >> 2) != (key >> 2));
  }
  const unsigned long long bal = __ballot(head);
  if (bal == 0) return;
  const int lane = threadIdx.x & 63;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(bal));
  base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base);
  if (head) {
    pm_hit h;
    h.pid = (uint32_t)(key >> 42); h.end = (int64_t)((key >> 2) & 0xffffffffffull); h.k = (uint8_t)(key & 3u);
    h.aux[0] = h.aux[1] = h.aux[2] = 0;
    out[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0))] = h;
  }
}

}  // namespace

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

size_t cluster_temp_bytes(size_t n) {
  size_t bytes = 0;
  uint64_t *p = nullptr;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, p, p, (int)n);
  return bytes;
}

hipError_t cluster_device(const pm_hit *d_in, size_t n, int k, int64_t scanned_to, bool last,
                          const uint8_t *d_pat_len, const uint32_t *d_pat_id,
                          uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                          pm_hit *d_out, pm_hit *d_left, unsigned long long *d_counts, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_counts, 0, 2 * sizeof(unsigned long long), st);
  if (e != hipSuccess || n == 0) return e;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads);
  hipLaunchKernelGGL(pm_cluster_pack, dim3(blocks), dim3(threads), 0, st, d_in, n, d_keys);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys_alt, (int)n, 0, 64, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(pm_cluster_reduce, dim3(blocks), dim3(threads), 0, st, d_keys_alt, n, 2 * k + 1, scanned_to, last ? 1 : 0,
                     d_pat_len, d_pat_id, d_out, d_counts, d_left, d_counts + 1);
  return hipGetLastError();
}

}  // namespace pm

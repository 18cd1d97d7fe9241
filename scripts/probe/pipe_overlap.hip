// Hardware probe (not product code): do the three pipes the pair-plan scan kernel loads -- VALU issue, random
// ds_read_b32, and the L1 miss path of a sparse gather from an L2-resident table -- overlap at 16 waves per CU
// (one 1024-thread workgroup per CU, LDS bound), or do their times add?  And what does the width of the
// gathered element cost?  Per loop step a lane does: [1 ds_read_b32 at a random LDS word] [1 global load of
// W bytes at a random table entry with probability p, else entry 0] [V dependent-free VALU ops].
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/pipe_overlap.hip -o scripts/probe/pipe_overlap && scripts/probe/pipe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int W, int V, bool LDS, bool GATHER>
__global__ __launch_bounds__(1024) void k(const char *table, uint32_t idx_mask, uint32_t thr, int iters, uint32_t *out) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 32768; i += 1024) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0, y = x ^ 0x5bd1e995u;
  for (int it = 0; it < iters; ++it) {
    uint32_t lv[8];
    uint32_t gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x * 1664525u + 1013904223u;
      if constexpr (LDS) lv[u] = lds[(x >> 9) & 32767u]; else lv[u] = x;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t key = ((lv[u] ^ x) >> 6) & idx_mask;
      const bool hit = ((lv[u] * 0x9e3779b1u) >> 24) < thr;
      const uint32_t off = (hit ? key : 0u) * W;
      gv[u] = 0;
      if constexpr (GATHER) {
        if constexpr (W == 2) gv[u] = *reinterpret_cast<const uint16_t *>(table + off);
        else if constexpr (W == 4) gv[u] = *reinterpret_cast<const uint32_t *>(table + off);
        else if constexpr (W == 8) { const u32x2 t = *reinterpret_cast<const u32x2 *>(table + off); gv[u] = t.x ^ t.y; }
        else { const u32x4 t = *reinterpret_cast<const u32x4 *>(table + off); gv[u] = t.x ^ t.y ^ t.z ^ t.w; }
      } else gv[u] = off;
#pragma unroll
      for (int v = 0; v < V; ++v) y = (y ^ (y >> 3)) + (uint32_t)v;     // 2 VALU ops per round, one dependent chain per lane
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += gv[u];
  }
  if ((acc ^ y) == 0x7fffffffu) out[0] = acc;
}

template <int W, int V, bool LDS, bool GATHER>
float run(const char *d_table, uint32_t idx_mask, uint32_t thr, uint32_t *d_out) {
  const int iters = 256, blocks = 256 * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<W, V, LDS, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<W, V, LDS, GATHER>), dim3(blocks), dim3(1024), 150 * 1024, 0, d_table, idx_mask, thr, iters, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)blocks * 16 * iters * 8;             // wave-level steps
  printf("W=%2d V=%2d lds=%d gather=%d table %5zu KiB p=%.3f: %7.3f ms  %6.2f CU-ns per wave-step\n", W, 2 * V, (int)LDS, (int)GATHER,
         ((size_t)idx_mask + 1) * W / 1024, thr / 256.0, ms, ms * 1e6 * 256 / steps);
  return ms;
}

int main() {
  const size_t bytes = 16u << 20;
  char *d; uint32_t *d_out;
  (void)hipMalloc(&d, bytes); (void)hipMalloc(&d_out, 64);
  (void)hipMemset(d, 3, bytes);
  const uint32_t M20 = (1u << 20) - 1, M19 = (1u << 19) - 1, M18 = (1u << 18) - 1;
  // element width at 2^20 entries and at a fixed 2 MiB footprint
  run<2, 0, false, true>(d, M20, 44, d_out);
  run<4, 0, false, true>(d, M20, 44, d_out);
  run<8, 0, false, true>(d, M20, 44, d_out);
  run<16, 0, false, true>(d, M20, 44, d_out);
  run<8, 0, false, true>(d, M19, 44, d_out);
  run<8, 0, false, true>(d, M18, 44, d_out);
  run<16, 0, false, true>(d, M18, 44, d_out);
  // the pipes alone
  run<2, 0, true, false>(d, M20, 44, d_out);
  run<2, 16, false, false>(d, M20, 44, d_out);
  run<2, 8, false, false>(d, M20, 44, d_out);
  // pairs and all three (2-byte elements, 2 MiB table)
  run<2, 16, true, false>(d, M20, 44, d_out);
  run<2, 16, false, true>(d, M20, 44, d_out);
  run<2, 0, true, true>(d, M20, 44, d_out);
  run<2, 16, true, true>(d, M20, 44, d_out);
  run<2, 8, true, true>(d, M20, 44, d_out);
  run<8, 16, true, true>(d, M18, 44, d_out);
  run<8, 16, true, true>(d, M19, 44, d_out);
  return 0;
}

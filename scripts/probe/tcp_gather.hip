// Hardware probe (not product code): what a wave-level 2-byte gather from an L2-resident table costs on
// gfx950, by how its 64 lanes' addresses are spread -- the per-window table lookup of pm_pair_scan /
// pm_edit_scan.  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/tcp_gather.hip -o scripts/probe/tcp_gather && scripts/probe/tcp_gather
// Modes (every lane of every wave issues `iters` x 16 loads, 16 in flight):
//   0  every lane reads entry 0                                   (all "dummy")
//   1  a lane reads a random entry with probability p, else entry 0, branch-free   (what pm_pair_scan does)
//   2  a lane reads a random entry with probability p, else does not load (exec mask)
//   3  every lane reads a random entry
//   4  as 2, but the four lanes of a quad decide together (p of the quads fully active)
//   5  as 1, but the dummy lanes read the entry of the nearest lower lane that has a real one (no line of their own)
// Output: CU-nanoseconds per wave-level load instruction (kernel time x CUs / instructions) and lane-lookups/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int MODE, typename T>
__global__ __launch_bounds__(1024) void gather(const T *table, uint32_t idx_mask, uint32_t thr, int iters, uint32_t *out) {
  extern __shared__ uint32_t lds[];
  uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t qx = ((blockIdx.x * 1024u + threadIdx.x) >> 2) * 2246822519u + 777u;     // quad-uniform stream
  int acc = 0;
  if (threadIdx.x == 5000) lds[0] = 1;                                              // keep the LDS block
  for (int it = 0; it < iters; ++it) {
    int v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x = x * 1664525u + 1013904223u;
      qx = qx * 1664525u + 1013904223u;
      const uint32_t key = (x >> 6) & idx_mask;
      const bool hit = (x >> 24) < thr;
      v[u] = 0;
      if constexpr (MODE == 0) v[u] = table[0];
      else if constexpr (MODE == 1) v[u] = table[hit ? key : 0u];
      else if constexpr (MODE == 2) { if (hit) v[u] = table[key]; }
      else if constexpr (MODE == 3) v[u] = table[key];
      else if constexpr (MODE == 4) { if ((qx >> 24) < thr) v[u] = table[key]; }
      else if constexpr (MODE == 5) {
        // dummy lanes copy the index of the nearest lower hit lane (lane 0: its own key)
        const unsigned long long bal = __ballot(hit) | 1ull;
        const int lane = threadIdx.x & 63;
        const unsigned long long below = bal & ((2ull << lane) - 1ull);
        const int src = 63 - __clzll((long long)below);
        const uint32_t k2 = __shfl(key, src);
        v[u] = table[k2];
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (acc == 0x7fffffff) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (uint32_t)acc;
}

template <int MODE, typename T>
void run(const char *what, const T *d_table, uint32_t idx_mask, uint32_t thr, int blocks, size_t lds, uint32_t *d_out, double lanes_frac) {
  const int iters = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void *>(gather<MODE, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((gather<MODE, T>), dim3(blocks), dim3(1024), lds, 0, d_table, idx_mask, thr, iters, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 16 * iters * 16;            // wave-level load instructions
  printf("mode %d %-46s table %4zu KiB x%zuB p=%.3f lds=%3zuK: %7.3f ms  %6.2f CU-ns/instr  %7.1f G lane-lookups/s (real %.1f G/s)  err=%s\n",
         MODE, what, ((size_t)idx_mask + 1) * sizeof(T) / 1024, sizeof(T), thr / 256.0, lds / 1024, ms, ms * 1e6 * 256 / instr,
         instr * 64 / ms / 1e6, instr * 64 * lanes_frac / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t n = 1 << 21;
  std::vector<int16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (int16_t)(i * 2654435761u >> 20);
  int16_t *d16; int32_t *d32; uint32_t *d_out;
  hipMalloc(&d16, n * 2); hipMalloc(&d32, n * 4); hipMalloc(&d_out, 64);
  hipMemcpy(d16, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemset(d32, 1, n * 4);
  const int blocks = 256 * 8;
  const uint32_t M20 = (1u << 20) - 1;
  for (size_t lds : {(size_t)150 * 1024, (size_t)64 * 1024}) {       // 1 or 2 workgroups of 16 waves per CU
    run<0>("all lanes entry 0", d16, M20, 44, blocks, lds, d_out, 0.0);
    run<1>("17% random, rest entry 0 (branch-free)", d16, M20, 44, blocks, lds, d_out, 44 / 256.0);
    run<2>("17% random, rest masked off", d16, M20, 44, blocks, lds, d_out, 44 / 256.0);
    run<4>("17% of the quads random, rest masked off", d16, M20, 44, blocks, lds, d_out, 44 / 256.0);
    run<5>("17% random, rest share a hit lane's entry", d16, M20, 44, blocks, lds, d_out, 44 / 256.0);
    run<3>("all lanes random", d16, M20, 44, blocks, lds, d_out, 1.0);
    run<1>("9% random, rest entry 0", d16, M20, 23, blocks, lds, d_out, 23 / 256.0);
    run<2>("9% random, rest masked off", d16, M20, 23, blocks, lds, d_out, 23 / 256.0);
    run<2>("2% random, rest masked off", d16, M20, 5, blocks, lds, d_out, 5 / 256.0);
    run<1>("17% random 4-byte entries, rest entry 0", d32, M20, 44, blocks, lds, d_out, 44 / 256.0);
    run<1>("17% random in a 256 KiB table, rest entry 0", d16, (1u << 17) - 1, 44, blocks, lds, d_out, 44 / 256.0);
    run<1>("17% random in a 32 KiB table, rest entry 0", d16, (1u << 14) - 1, 44, blocks, lds, d_out, 44 / 256.0);
    run<3>("all lanes random in a 16 KiB table", d16, (1u << 13) - 1, 44, blocks, lds, d_out, 1.0);
  }
  return 0;
}

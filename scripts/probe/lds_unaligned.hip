// Hardware probe (not product code): does gfx950 serve ds_read_b32 at byte addresses that are not
// multiples of four, and where does a kernel's dynamic LDS block start?  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/lds_unaligned.hip -o scripts/probe/lds_unaligned && scripts/probe/lds_unaligned
// Result on MI355X: every lane reads the four bytes at its byte address (base=0); the scan kernels
// nevertheless ran 2x slower with byte-granular filter blocks, so pm_seed.hip keeps dword blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
__global__ void k(uint32_t *out) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x)          // byte b of the block holds b & 255
    lds[i] = ((4u * i) & 255u) | (((4u * i + 1u) & 255u) << 8) | (((4u * i + 2u) & 255u) << 16) | (((4u * i + 3u) & 255u) << 24);
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(lds_u32 *)lds;
  const uint32_t addr = threadIdx.x;     // byte address, unaligned for most lanes
  out[threadIdx.x] = *(lds_u32 *)(uintptr_t)addr;
  if (threadIdx.x == 0) out[256] = base;
}
int main() {
  uint32_t *d; hipMalloc(&d, 4 * 300);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d);
  uint32_t h[300]; hipMemcpy(h, d, 4 * 300, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) { uint32_t want = 0; for (int b = 0; b < 4; ++b) want |= (uint32_t)((i + b) & 0xff) << (8 * b); if (h[i] != want) { if (bad < 5) printf("lane %d got %08x want %08x\n", i, h[i], want); ++bad; } }
  printf("base=%u bad=%d err=%s\n", h[256], bad, hipGetErrorString(hipGetLastError()));
  return 0;
}

// pm_util.hip -- measurement helper: the streaming-read rate this box sustains.
//
// SURVEY.md 8(d) asks for a measured read ceiling from the same box next to the vendor's 8 TB/s:
// every scan kernel reads the stream once with 16-byte loads per lane, and so does this kernel --
// with nothing else to do.  It is not on the product path; bench.py reports its rate beside the
// roofline fraction.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pm_gpu.h"
#include "pm_internal.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pm_stream_read(const u32x4 *p, size_t n16, uint32_t *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  // four independent 16-byte loads in flight per lane
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride),
                c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) {
    const u32x4 a = __builtin_nontemporal_load(p + i);
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x9e3779b9u && sink) sink[0] = acc;                  // keeps the loads alive; practically never taken
}

// which byte values occur in the stream: 256 flags (as 8 dwords), OR-reduced through LDS
__global__ __launch_bounds__(256) void pm_stream_presence(const uint8_t *p, size_t n, uint32_t *flags) {
  __shared__ uint32_t s[8];
  if (threadIdx.x < 8) s[threadIdx.x] = 0;
  __syncthreads();
  uint32_t loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint32_t b = p[i];
#pragma unroll
    for (int w = 0; w < 8; ++w) loc[w] |= (b >> 5) == (uint32_t)w ? 1u << (b & 31) : 0u;
  }
#pragma unroll
  for (int w = 0; w < 8; ++w) if (loc[w]) atomicOr(&s[w], loc[w]);
  __syncthreads();
  if (threadIdx.x < 8 && s[threadIdx.x]) atomicOr(&flags[threadIdx.x], s[threadIdx.x]);
}

}  // namespace

namespace pm {

hipError_t stream_presence(const uint8_t *d_text, int64_t n, bool present[256], hipStream_t st) {
  for (int i = 0; i < 256; ++i) present[i] = false;
  if (n <= 0) return hipSuccess;
  uint32_t *d = nullptr;
  hipError_t e = hipMalloc((void **)&d, 32);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d, 0, 32, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pm_stream_presence, dim3(256 * 8), dim3(256), 0, st, d_text, (size_t)n, d);
    e = hipGetLastError();
  }
  uint32_t hflags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(hflags, d, 32, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  for (int i = 0; i < 256; ++i) present[i] = (hflags[i >> 5] >> (i & 31)) & 1u;
  return e;
}

}  // namespace pm

extern "C" int pm_measure_stream_read(const void *d_buf, size_t bytes, int reps, void *stream, float *gbytes_per_s) {
  if (!d_buf || bytes < 16 || reps < 1 || !gbytes_per_s) return PM_E_INVALID;
  if ((uintptr_t)d_buf & 15) return PM_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return PM_E_HIP;
  uint32_t *sink = nullptr;
  if (hipMalloc((void **)&sink, 16) != hipSuccess) return PM_E_HIP;
  const size_t n16 = bytes / 16;
  const dim3 grid(256 * 8), block(256);
  hipLaunchKernelGGL(pm_stream_read, grid, block, 0, st, reinterpret_cast<const u32x4 *>(d_buf), n16, sink);   // warm-up
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(pm_stream_read, grid, block, 0, st, reinterpret_cast<const u32x4 *>(d_buf), n16, sink);
  (void)hipEventRecord(e1, st);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  if (e != hipSuccess || ms <= 0) return PM_E_HIP;
  *gbytes_per_s = (float)((double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9);
  return PM_OK;
}

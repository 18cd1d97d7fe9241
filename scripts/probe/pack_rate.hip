// pack_rate.hip -- how fast can the 1-byte-per-base stream be re-encoded at 2 bits per base (pm_pack_stream, pm_seed.hip)?
// 3 GB read + 0.75 GB written per pass; variants differ in bytes per thread, loads in flight and block size.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probe/pack_rate.hip -o scripts/probe/pack_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack4(uint32_t x, int sh) {
  uint32_t y = (x >> sh) & 0x03030303u;
  y |= y >> 6;
  return (y | (y >> 12)) & 0xffu;
}
__device__ __forceinline__ uint32_t pack16(const u32x4 &v, int sh) {
  return pack4(v.x, sh) | (pack4(v.y, sh) << 8) | (pack4(v.z, sh) << 16) | (pack4(v.w, sh) << 24);
}

// today's form: one 16-byte load, one dword store per thread
template <bool NT>
__global__ void k_one(const uint8_t *text, int sh, uint32_t *packed, int64_t npacked) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= npacked) return;
  const u32x4 *p = reinterpret_cast<const u32x4 *>(text) + i;
  const u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
  packed[i] = pack16(v, sh);
}

// U loads in flight per thread, the workgroup's loads side by side (coalesced), dword stores
template <int U, bool NT>
__global__ void k_unroll(const uint8_t *text, int sh, uint32_t *packed, int64_t npacked) {
  const int64_t base = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * blockDim.x;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(text) + (i < npacked ? i : 0);
    v[u] = NT ? __builtin_nontemporal_load(p) : *p;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * blockDim.x;
    if (i < npacked) packed[i] = pack16(v[u], sh);
  }
}

// 64 bytes per thread in one row: four 16-byte loads (lane stride 64 bytes), one 16-byte store
template <bool NT>
__global__ void k_row64(const uint8_t *text, int sh, uint32_t *packed, int64_t npacked) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;      // output u32x4 index
  if (4 * t + 3 >= npacked) return;
  const u32x4 *p = reinterpret_cast<const u32x4 *>(text) + 4 * t;
  u32x4 a, b, c, d;
  if (NT) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 1); c = __builtin_nontemporal_load(p + 2); d = __builtin_nontemporal_load(p + 3); }
  else { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
  u32x4 o;
  o.x = pack16(a, sh); o.y = pack16(b, sh); o.z = pack16(c, sh); o.w = pack16(d, sh);
  reinterpret_cast<u32x4 *>(packed)[t] = o;
}

// coalesced 16-byte loads, the workgroup transposes through LDS so that every lane stores 16 bytes
template <int U>
__global__ void k_lds(const uint8_t *text, int sh, uint32_t *packed, int64_t npacked) {
  __shared__ uint32_t s[256 * U];
  const int64_t blockbase = (int64_t)blockIdx.x * 256 * U;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = blockbase + u * 256 + threadIdx.x;
    v[u] = reinterpret_cast<const u32x4 *>(text)[i < npacked ? i : 0];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) s[u * 256 + threadIdx.x] = pack16(v[u], sh);
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * U; q += 256) {
    const int64_t i = blockbase + 4 * q;
    if (i + 3 < npacked) reinterpret_cast<u32x4 *>(packed)[i >> 2] = reinterpret_cast<u32x4 *>(s)[q];
    else for (int e = 0; e < 4; ++e) if (i + e < npacked) packed[i + e] = s[4 * q + e];
  }
}

__global__ void k_fill(uint8_t *text, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n / 4) return;
  uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  const char L[4] = {'A', 'C', 'G', 'T'};
  uint32_t w = 0;
  for (int b = 0; b < 4; ++b) w |= (uint32_t)L[(x >> (2 * b + 7)) & 3] << (8 * b);
  reinterpret_cast<uint32_t *>(text)[i] = w;
}

__global__ void k_sum(const uint32_t *p, int64_t n, unsigned long long *out) {
  unsigned long long s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (unsigned long long)p[i] * (unsigned long long)((i & 1023) + 1);
  atomicAdd(out, s);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int64_t n = 3000000000LL, np = n / 16;
  uint8_t *text; uint32_t *packed; unsigned long long *sum;
  CK(hipMalloc(&text, n + 64)); CK(hipMalloc(&packed, np * 4 + 64)); CK(hipMalloc(&sum, 8));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, text, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    CK(hipMemset(packed, 0, np * 4));
    launch();                                                      // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms;
    }
    CK(hipGetLastError());
    CK(hipMemset(sum, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, packed, np, sum);
    unsigned long long h; CK(hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost));
    printf("%-28s best %.3f ms  mean %.3f ms  %.2f TB/s (read + write)  check %016llx\n", name, best, tot / 5, (double)(n + np * 4) / best / 1e9, h);
  };
  const int sh = 1;
  run("one/256 nt", [&] { hipLaunchKernelGGL((k_one<true>), dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, text, sh, packed, np); });
  run("one/256", [&] { hipLaunchKernelGGL((k_one<false>), dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, text, sh, packed, np); });
  run("one/1024 nt", [&] { hipLaunchKernelGGL((k_one<true>), dim3((unsigned)((np + 1023) / 1024)), dim3(1024), 0, 0, text, sh, packed, np); });
  run("unroll2/256 nt", [&] { hipLaunchKernelGGL((k_unroll<2, true>), dim3((unsigned)((np + 511) / 512)), dim3(256), 0, 0, text, sh, packed, np); });
  run("unroll4/256 nt", [&] { hipLaunchKernelGGL((k_unroll<4, true>), dim3((unsigned)((np + 1023) / 1024)), dim3(256), 0, 0, text, sh, packed, np); });
  run("unroll4/256", [&] { hipLaunchKernelGGL((k_unroll<4, false>), dim3((unsigned)((np + 1023) / 1024)), dim3(256), 0, 0, text, sh, packed, np); });
  run("unroll8/256 nt", [&] { hipLaunchKernelGGL((k_unroll<8, true>), dim3((unsigned)((np + 2047) / 2048)), dim3(256), 0, 0, text, sh, packed, np); });
  run("unroll4/512 nt", [&] { hipLaunchKernelGGL((k_unroll<4, true>), dim3((unsigned)((np + 2047) / 2048)), dim3(512), 0, 0, text, sh, packed, np); });
  run("row64/256 nt", [&] { hipLaunchKernelGGL((k_row64<true>), dim3((unsigned)((np / 4 + 255) / 256)), dim3(256), 0, 0, text, sh, packed, np); });
  run("row64/256", [&] { hipLaunchKernelGGL((k_row64<false>), dim3((unsigned)((np / 4 + 255) / 256)), dim3(256), 0, 0, text, sh, packed, np); });
  run("lds4/256", [&] { hipLaunchKernelGGL((k_lds<4>), dim3((unsigned)((np + 1023) / 1024)), dim3(256), 0, 0, text, sh, packed, np); });
  run("lds8/256", [&] { hipLaunchKernelGGL((k_lds<8>), dim3((unsigned)((np + 2047) / 2048)), dim3(256), 0, 0, text, sh, packed, np); });
  // plain device-to-device copy of the same bytes for scale
  {
    uint8_t *dst; CK(hipMalloc(&dst, n));
    CK(hipMemcpy(dst, text, n, hipMemcpyDeviceToDevice));
    CK(hipEventRecord(e0, 0)); CK(hipMemcpyAsync(dst, text, n, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemcpy D2D 3 GB: %.3f ms  %.2f TB/s (read + write)\n", ms, 2.0 * n / ms / 1e9);
    CK(hipFree(dst));
  }
  return 0;
}

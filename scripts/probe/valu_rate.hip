// Hardware probe (not product code): issue rate of the integer / bitwise VALU instructions the scan kernels are
// made of, at the occupancy they run at (16 waves per CU = 4 per SIMD) -- cycles per wave-level instruction per
// SIMD, assuming the 2.4 GHz peak clock (the clock under load is lower: compare the rows with each other).
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/valu_rate.hip -o scripts/probe/valu_rate && scripts/probe/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define BODY(ASM) \
  for (int it = 0; it < iters; ++it) { \
    REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sc));) \
  }

#define KERNEL(NAME, ASM) \
__global__ __launch_bounds__(1024) void NAME(int iters, uint32_t *out) { \
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = a0 | 5, c = a0 ^ 0x5555; \
  const uint32_t sc = (uint32_t)iters | 0x55555u; \
  BODY(ASM) \
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) out[0] = a0; \
}

// eight independent instructions per asm block (one per accumulator)
#define I8(OP, ARGS) OP " %0, " ARGS("%0") "\n" OP " %1, " ARGS("%1") "\n" OP " %2, " ARGS("%2") "\n" OP " %3, " ARGS("%3") "\n" \
                     OP " %4, " ARGS("%4") "\n" OP " %5, " ARGS("%5") "\n" OP " %6, " ARGS("%6") "\n" OP " %7, " ARGS("%7") "\n"
#define A2(X) X ", %8"
#define A2S(X) "%10, " X
#define A3(X) X ", %8, %9"
#define A3S(X) X ", %8, %10"
#define A1(X) X
#define ASH(X) "3, " X
#define ABFE(X) X ", 3, 7"
#define AALN(X) X ", %8, 5"
#define ABIT(X) X ", %8, %10 bitop3:0xa8"
#define ALSA(X) X ", 3, %8"
#define ADPP(X) X " wave_shr:1"
#define ALIT(X) "0x1fffc, " X
#define ALITB(X) "0x55555555, " X
#define AVSH(X) "%8, " X
#define ASH17(X) "17, " X
#define ACND(X) X ", %8, vcc"
#define ACNDS(X) X ", %8, s[10:11]"
#define ACMP(X) "vcc, " X ", %8"
#define ASH64(X) "3, " X
#define ASDWA(X) "%8, " X " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"

KERNEL(k_xor, I8("v_xor_b32", A2))
KERNEL(k_and, I8("v_and_b32", A2))
KERNEL(k_add, I8("v_add_u32", A2))
KERNEL(k_and_s, I8("v_and_b32", A2S))
KERNEL(k_lshr, I8("v_lshrrev_b32", ASH))
KERNEL(k_lshrv, I8("v_lshrrev_b32", A2S))
KERNEL(k_mov, I8("v_mov_b32", A1))
KERNEL(k_bcnt, I8("v_bcnt_u32_b32", A2))
KERNEL(k_bfe, I8("v_bfe_u32", ABFE))
KERNEL(k_bfev, I8("v_bfe_u32", A3))
KERNEL(k_align, I8("v_alignbit_b32", AALN))
KERNEL(k_alignv, I8("v_alignbit_b32", A3))
KERNEL(k_bitop3, I8("v_bitop3_b32", ABIT))
KERNEL(k_bfi, I8("v_bfi_b32", A3))
KERNEL(k_min, I8("v_min_u32", A2))
KERNEL(k_min3, I8("v_min3_u32", A3))
KERNEL(k_or3, I8("v_or3_b32", A3))
KERNEL(k_add3, I8("v_add3_u32", A3))
KERNEL(k_xad, I8("v_xad_u32", A3))
KERNEL(k_lsa, I8("v_lshl_add_u32", ALSA))
KERNEL(k_mul24, I8("v_mul_u32_u24", A2))
KERNEL(k_mad24, I8("v_mad_u32_u24", A3))
KERNEL(k_mullo, I8("v_mul_lo_u32", A2))
KERNEL(k_perm, I8("v_perm_b32", A3))
KERNEL(k_dpp, I8("v_mov_b32_dpp", ADPP))
KERNEL(k_sdwa, I8("v_lshrrev_b32_sdwa", ASDWA))
KERNEL(k_fma, I8("v_fma_f32", A3))
KERNEL(k_pkadd, I8("v_pk_add_u16", A2))
KERNEL(k_andor, I8("v_and_or_b32", A3))
KERNEL(k_ashr, I8("v_ashrrev_i32", ASH))
KERNEL(k_ffbl, I8("v_ffbl_b32", A1))
KERNEL(k_cvt, I8("v_cvt_f32_u32", A1))
KERNEL(k_cnd_s, I8("v_cndmask_b32_e64", ACNDS))
KERNEL(k_mulhi, I8("v_mul_hi_u32", A2))
KERNEL(k_sad, I8("v_sad_u32", A3))
KERNEL(k_sadu8, I8("v_sad_u8", A3))
KERNEL(k_bfm, I8("v_bfm_b32", A2))
KERNEL(k_mbcnt, I8("v_mbcnt_lo_u32_b32", A2))
KERNEL(k_alignbyte, I8("v_alignbyte_b32", A3))
KERNEL(k_dot4, I8("v_dot4_i32_i8", A3))
KERNEL(k_dot8, I8("v_dot8_i32_i4", A3))
KERNEL(k_subrev, I8("v_subrev_u32", A2))
KERNEL(k_max3, I8("v_max3_u32", A3))
KERNEL(k_and_lit, I8("v_and_b32", ALIT))
KERNEL(k_and_lit2, I8("v_and_b32", ALITB))
KERNEL(k_xor_lit, I8("v_xor_b32", ALITB))
KERNEL(k_lshr_v, I8("v_lshrrev_b32", AVSH))
KERNEL(k_lshl_v, I8("v_lshlrev_b32", AVSH))
KERNEL(k_or, I8("v_or_b32", A2))
KERNEL(k_sub, I8("v_sub_u32", A2))
KERNEL(k_cnd, I8("v_cndmask_b32", ACND))
KERNEL(k_not, I8("v_not_b32", A1))
KERNEL(k_lshl17, I8("v_lshlrev_b32", ASH17))
KERNEL(k_max, I8("v_max_u32", A2))
KERNEL(k_addi, I8("v_add_u32", ASH17))

template <typename K>
void run(const char *name, K kern, uint32_t *d_out, int lds_bytes, int blocks_per_cu) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds_bytes, 0, iters, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)blocks_per_cu * 16 / 4 * iters * 64;          // wave-instructions per SIMD
  printf("%-22s %7.3f ms  %5.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
  uint32_t *d_out;
  (void)hipMalloc(&d_out, 64);
#define R(K) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run(#K, K, d_out, 150 * 1024, 4);
  R(k_xor) R(k_and) R(k_add) R(k_and_s) R(k_lshr) R(k_lshrv) R(k_mov) R(k_bcnt) R(k_bfe) R(k_bfev) R(k_align) R(k_alignv) R(k_bitop3) R(k_bfi)
  R(k_min) R(k_min3) R(k_or3) R(k_add3) R(k_xad) R(k_lsa) R(k_mul24) R(k_mad24) R(k_mullo) R(k_perm) R(k_dpp) R(k_sdwa) R(k_fma) R(k_pkadd) R(k_andor)
  R(k_cnd_s) R(k_mulhi) R(k_sad) R(k_sadu8) R(k_bfm) R(k_mbcnt) R(k_alignbyte) R(k_dot4) R(k_dot8) R(k_subrev) R(k_max3)
  R(k_ashr) R(k_ffbl) R(k_cvt) R(k_and_lit) R(k_and_lit2) R(k_xor_lit) R(k_lshr_v) R(k_lshl_v) R(k_or) R(k_sub) R(k_cnd) R(k_not) R(k_lshl17) R(k_max) R(k_addi)
  printf("-- two workgroups of 1024 per CU (8 waves per SIMD):\n");
#define R2(K) run(#K, K, d_out, 64 * 1024, 8);
  R2(k_xor) R2(k_alignv) R2(k_bitop3) R2(k_bcnt) R2(k_fma)
  return 0;
}

// pm_pair.hip -- "pair plan" scan kernels for gfx950: exact 20-bit key bitmap in LDS, a direct-mapped
// 2-byte table in L2 for the keys that occur, rank-indexed exact table for the few windows left.
//
// What it computes: every (end, pattern, d) with d = Hamming(stream window, pattern) <= k, k = 1 or 2,
// and no EOS inside the window -- the candidate set of the reference's substitution-only k-error
// automaton (shift_and_inexact.cc:249-352 with indels = false), on which filter_bitvec
// (filter_bitvec.cc:88-177) and exact_halves (exact_halves.cc:120-197) build.  Patterns are A,C,G,T
// strings of 20..32 characters; the plan looks at their last 20 bases.
//
// How.  The 40-bit packed window (2 bits per base) is four fields of 10 bits.  <= 2 substitutions
// leave >= 2 fields untouched: the window agrees with the pattern on one of the 6 field pairs
// (k = 1: on fields {0,1} or {2,3}).  A field pair is a 20-bit key, and 2^20 key bits are exactly a
// 128 KiB bitmap:
//   * a WORKGROUP owns one (field pair, stream chunk): it stages that pair's key bitmap (and a rank
//     directory over it) in LDS, then streams its chunk of the 2-bit packed stream;
//   * a LANE owns 16 consecutive window positions per block of 1024; its two predecessors' dwords
//     come in by two whole-wave DPP shifts.  Everything that runs for every window is straight-line
//     code with compile-time window offsets:
//       test     one v_alignbit (the key, pre-shifted by 2; field pairs that are apart: a second one
//                and a v_bfi), one v_and (LDS byte address: key bits 0..14 select the word), one
//                ds_read_b32, two shifts (key bits 15..19 select the bit; verdict funnelled into a
//                16-bit mask), then one 2-byte load from the pair's direct-mapped table in L2 (2 MiB,
//                index = key): six bases of the first pattern that has the key and a "several patterns
//                share this key" bit.  Windows whose key does not occur (83 % at 200k patterns) read
//                entry 0 -- one cached line for all of them -- so the load needs no branch;
//       consume  one block later, when the loads have landed: XOR + popcount of those six bases
//                against the window's other fields.  ~2 % of the windows stay suspicious;
//   * those are compacted per wave (ballot + mbcnt) into an LDS queue and resolved 64 at a time: the
//     key's RANK among the set bits of the bitmap (two-level directory + popcounts, all in LDS)
//     indexes a dense table with one 8-byte slot per distinct key -- the other 20 window bits of up
//     to three patterns -- whose load is consumed one batch later (XOR + popcount on all 20 bits);
//   * what is still within k there (1e-3 of the positions) goes to a suspect list; a second kernel,
//     pm_pair_verify, reads the raw stream bytes for the exact distance (N = mismatch, EOS = reject)
//     with every lane busy;
//   * a (window, pattern) pair that agrees on several field pairs is reported by the first of them
//     in the plan's list.
//
// Bounds (3 Gbp, 200k patterns, k = 2; PM_SEED_DEBUG stage switches, scripts/pair_stages.sh): the
// straight-line part is LDS-bank-conflict and VALU bound (8.5 ms for 1.8e10 tests: 5 + 4 + 9 VALU per
// test, random ds_read_b32 ~7 cycles per wave instruction), the 3.1e9 direct-table lookups add 4.9 ms
// (the L1 fill path moves one 128-byte line per lookup), the suspicious 2 % another 4 ms.
#include "pm_internal.h"
#include "pm_pair.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <type_traits>

namespace pm {

namespace {

constexpr int WAVES = PAIR_WAVES;
constexpr uint32_t SUPER_OFF = PAIR_BITMAP_WORDS * 4;                 // LDS byte offsets; the bitmap sits at LDS address 0
constexpr uint32_t REL_OFF = SUPER_OFF + PAIR_SUPER * 4;
constexpr uint32_t WAVE_OFF = PAIR_IMAGE_WORDS * 4;
constexpr uint32_t WAVE_BYTES = 2 * PAIR_QREGION * 8;
constexpr uint32_t F20 = 0xfffffu;
constexpr int PAIR_SUSPECT_BLOCK = 16;                               // suspect slots a wave reserves per atomic
constexpr uint32_t PAIR_SUSPECT_HOLE = 0xffffffffu;                  // rank field of a reserved slot that stayed unused
// Exact table, one 8-byte slot per distinct key: the other 20 window bits of up to three patterns that
// have the key (o1 | o2 << 20 | o3 << 40) and, in bits 60..63, how many there are (4 = more than three).
__device__ __host__ __forceinline__ uint64_t slot_pack(uint32_t o1, uint32_t o2, uint32_t o3, uint32_t count) {
  return (uint64_t)o1 | ((uint64_t)o2 << 20) | ((uint64_t)o3 << 40) | ((uint64_t)(count > 4 ? 4 : count) << 60);
}

struct PairArgs {
  const uint8_t *text;
  int64_t n, begin, end;                // owned hit ends: begin < end_pos <= end
  const uint32_t *packed;               // the stream, 2 bits per base, 16 bases per dword
  int64_t npacked;
  int64_t chunk0, chunk_len;            // first chunk index (absolute), positions per workgroup
  int nchunks, ncombos, group;
  int k, eos_code, debug;
  int fa[PAIR_MAX_COMBOS], fb[PAIR_MAX_COMBOS];
  const uint32_t *image;                // [combo][PAIR_IMAGE_WORDS]
  const int16_t *direct;                // [combo][2^20]: direct-mapped by key, see pair_build
  const uint2 *entries;                 // all combos; combo c starts at entries_off[c]
  const uint32_t *first_pat, *order;
  uint32_t entries_off[PAIR_MAX_COMBOS], first_off[PAIR_MAX_COMBOS];
  uint32_t np;
  const uint2 *pat40;
  const uint8_t *pat_len;
  const uint32_t *pat_id;
  const uint8_t *pat_codes;
  const uint32_t *pat_zone;             // per pattern: bit i = character i lies in an exact zone (exact_start_bases / exact_end_bases)
  int viol_level;                       // a mismatch inside an exact zone: > 0 = report the candidate with this level (filter_bitvec
                                        // chains every candidate and verifies the chain), 0 = not a candidate (exact_halves, exact_bases)
  pm_hit *out;
  unsigned long long *counter;
  unsigned long long cap;
  uint4 *susp;                          // suspect records for pm_pair_verify: {rank, other fields, position | combo << 40 | what << 44}
  unsigned long long *susp_count;
  unsigned long long susp_cap;
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u128;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u64;
__device__ __forceinline__ lds_u32 *lds32(uint32_t addr) { return reinterpret_cast<lds_u32 *>((uintptr_t)addr); }
__device__ __forceinline__ lds_u16 *lds16(uint32_t addr) { return reinterpret_cast<lds_u16 *>((uintptr_t)addr); }
__device__ __forceinline__ lds_u128 *lds128(uint32_t addr) { return reinterpret_cast<lds_u128 *>((uintptr_t)addr); }
__device__ __forceinline__ lds_u64 *lds64(uint32_t addr) { return reinterpret_cast<lds_u64 *>((uintptr_t)addr); }

__device__ __forceinline__ uint32_t load_packed(const uint32_t *packed, int64_t npacked, int64_t pos) {
  const int64_t i = pos >> 4;
  if (pos < 0 || i >= npacked) return 0u;
  return packed[i];
}

// 32 bits from bit O (compile time) of the 96-bit string prev2 : prev1 : cur (bit 0 = bit 0 of prev2)
template <int O>
__device__ __forceinline__ uint32_t bits_at(uint32_t p2, uint32_t p1, uint32_t cur) {
  static_assert(O >= 0 && O < 96, "offset");
  if constexpr (O == 0) return p2;
  else if constexpr (O < 32) return __builtin_amdgcn_alignbit(p1, p2, O);
  else if constexpr (O == 32) return p1;
  else if constexpr (O < 64) return __builtin_amdgcn_alignbit(cur, p1, O - 32);
  else return cur >> (O - 64);
}

// substitutions between two strings of 2-bit symbols
__device__ __host__ __forceinline__ int sym_distance(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc((x | (x >> 1)) & 0x55555555u);
#else
  return __builtin_popcount((x | (x >> 1)) & 0x55555555u);
#endif
}

// The 20-bit key of field pair (a, b) of a 40-bit window W and the other 20 bits (fields c < d).
// Bit layout of the LDS bitmap: key bits 0..14 = word, bits 15..19 = bit inside the word.
__device__ __host__ __forceinline__ uint32_t field_of(uint64_t W, int j) { return (uint32_t)(W >> (10 * j)) & 0x3ffu; }
__device__ __host__ __forceinline__ void other_fields(int a, int b, int *c, int *d) {
  int o[2], t = 0;
  for (int j = 0; j < 4; ++j) if (j != a && j != b) o[t++] = j;
  *c = o[0]; *d = o[1];
}

// Exact part of the verify (second kernel): (window ending at p, pattern pi) agree on this combo's key
// and are within k substitutions on the rest of the packed window; count mismatches on the raw stream
// codes over the whole pattern (N = mismatch, EOS = reject) and report -- once: only through the first
// combo of the plan whose two fields are clean.
__device__ __forceinline__ void pair_verify(const PairArgs &a, int combo, int64_t p, uint32_t pi) {
  const int L = a.pat_len[pi];
  const int64_t start = p + 1 - L;
  if (start < 0) return;
  // all 32 pattern codes and the 32 stream bytes from `start` at once (every load independent of the
  // others: this kernel is a chain of dependent loads as it is), per-byte verdicts by SWAR
  const uint4 *pcv = reinterpret_cast<const uint4 *>(a.pat_codes + (size_t)pi * 32);
  const uint4 pc0 = pcv[0], pc1 = pcv[1];
  uint32_t tw[8];
  if (start + 32 <= a.n) {
    uint4 t0, t1;
    __builtin_memcpy(&t0, a.text + start, 16);
    __builtin_memcpy(&t1, a.text + start + 16, 16);
    tw[0] = t0.x; tw[1] = t0.y; tw[2] = t0.z; tw[3] = t0.w; tw[4] = t1.x; tw[5] = t1.y; tw[6] = t1.z; tw[7] = t1.w;
  } else {                                            // the last bytes of the stream
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      tw[d] = 0;
      for (int b = 0; b < 4; ++b) { const int64_t q = start + 4 * d + b; if (q < a.n) tw[d] |= (uint32_t)a.text[q] << (8 * b); }
    }
  }
  const uint32_t pcw[8] = {pc0.x, pc0.y, pc0.z, pc0.w, pc1.x, pc1.y, pc1.z, pc1.w};
  const uint32_t eb = (uint32_t)(a.eos_code & 0xff) * 0x01010101u;
  uint32_t mism = 0, eos = 0;                         // bit i: stream byte i differs from the pattern / is EOS
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const uint32_t x = tw[d] ^ pcw[d], z = tw[d] ^ eb;
    const uint32_t y = (x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t e = ~(z | ((z & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    mism |= (((y >> 7) & 1u) | ((y >> 14) & 2u) | ((y >> 21) & 4u) | ((y >> 28) & 8u)) << (4 * d);
    eos |= (((e >> 7) & 1u) | ((e >> 14) & 2u) | ((e >> 21) & 4u) | ((e >> 28) & 8u)) << (4 * d);
  }
  const uint32_t lenmask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
  mism &= lenmask;
  if (a.eos_code >= 0 && (eos & lenmask)) return;     // EOS inside the window: never a candidate
  int ham = __popc(mism);                             // N (or any other code) = mismatch
  if (ham > a.k) return;
  // exact-base constraints (pattern_alignment.cc:320-323: a substitution inside an exact zone is a
  // constraint violation, the verify fails)
  if (mism & a.pat_zone[pi]) {
    if (a.viol_level <= 0) return;
    ham = a.viol_level;
  }
  const uint32_t tail = mism >> (L - 20);             // the 20 bases the plan looks at
  uint32_t dirty = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) if ((tail >> (5 * j)) & 31u) dirty |= 1u << j;
  int first = -1;
  for (int c = 0; c < a.ncombos && first < 0; ++c)
    if (!((dirty >> a.fa[c]) & 1u) && !((dirty >> a.fb[c]) & 1u)) first = c;
  if (first != combo) return;
  const int half = L / 2;
  const bool left_clean = (mism & ((1u << half) - 1u)) == 0, right_clean = (mism >> half) == 0;
  const unsigned long long o = atomicAdd(a.counter, 1ull);
  if (o < a.cap) {
    pm_hit hh;
    hh.end = p + 1; hh.pid = a.pat_id[pi]; hh.k = (uint8_t)ham;
    hh.aux[0] = (uint8_t)((left_clean ? 1 : 0) | (right_clean ? 2 : 0)); hh.aux[1] = hh.aux[2] = 0;
    a.out[o] = hh;
  }
}

// A suspect: a window whose exact-table slot says "within k on the other fields" for the key's first /
// second / third pattern (what & 1, 2, 4), or that the key has more than three (what & 8: walk the
// rest of the key's run of the sorted pattern list).
__device__ __forceinline__ void pair_resolve(const PairArgs &a, int combo, uint32_t rank, uint32_t what, uint32_t wo, int64_t p) {
  const uint32_t *fp = a.first_pat + a.first_off[combo];
  const uint32_t *ord = a.order + (size_t)combo * a.np;
  const uint32_t t0 = fp[rank];
  for (uint32_t j = 0; j < 3; ++j)
    if ((what >> j) & 1u) pair_verify(a, combo, p, ord[t0 + j]);
  if (what & 8u) {
    int c, d;
    other_fields(a.fa[combo], a.fb[combo], &c, &d);
    for (uint32_t t = t0 + 3; t < fp[rank + 1]; ++t) {
      const uint32_t pi = ord[t];
      const uint2 pp = a.pat40[pi];
      const uint64_t W = ((uint64_t)pp.y << 32) | pp.x;
      const uint32_t po = field_of(W, c) | (field_of(W, d) << 10);
      if (sym_distance(po ^ wo) <= a.k) pair_verify(a, combo, p, pi);
    }
  }
}

// Second kernel: the scan kernel's suspects (a few per thousand positions), one per lane, grid-stride;
// the count is read from device memory.  Here, with every lane busy, the dependent loads of the
// verify (pattern list, pattern, raw stream bytes) cost little; inside the scan kernel they ran with
// one or two live lanes per wave and held the wave for microseconds.
__global__ __launch_bounds__(256) void pm_pair_verify(PairArgs a) {
  unsigned long long n = *a.susp_count;
  if (n > a.susp_cap) n = a.susp_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint4 r = a.susp[i];
    if (r.x == PAIR_SUSPECT_HOLE) continue;
    const uint64_t pw = ((uint64_t)r.w << 32) | r.z;
    pair_resolve(a, (int)((pw >> 40) & 7u), r.x, (uint32_t)(pw >> 44) & 15u, r.y, (int64_t)(pw & 0xffffffffffull));
  }
}

// One (field pair, chunk) of the scan.  A, B: key fields (compile time: every window offset is an
// immediate).  See the file comment for the stages.
//
// Per block of 1024 positions (16 per lane) a wave runs three things, all but the last as straight,
// branch-free code over the lane's 16 windows:
//   test     key -> LDS bitmap bit (5 VALU + 1 ds_read_b32), then one 2-byte load per window from the
//            pair's direct-mapped table in L2 (index = key; windows whose key is absent read entry 0,
//            one cached line for all of them, so the load needs no branch and no exec juggling);
//   consume  (one block later, when those loads have landed) six bases of the first pattern that has
//            the key against the window's other fields, plus the entry's "several patterns share
//            this key" bit: only ~2 % of the windows stay suspicious;
//   resolve  those few are compacted (ballot + mbcnt) into a wave queue and resolved by whole waves:
//            rank of the key in the bitmap -> 8-byte entry with the other 20 bits of up to two
//            patterns -> XOR + popcount -> raw stream bytes for the exact distance.
template <int A, int B>
__device__ __forceinline__ void pair_scan_body(const PairArgs &a, const int combo, const int cj) {
  constexpr int C = (A != 0 && B != 0) ? 0 : ((A != 1 && B != 1) ? 1 : 2);
  constexpr int D = (A != 3 && B != 3) ? 3 : ((A != 2 && B != 2) ? 2 : 1);
  static_assert(A < B && C < D && C != A && C != B && D != A && D != B, "fields");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t QB = WAVE_OFF + (uint32_t)wave * WAVE_BYTES;       // this wave's queue of suspicious windows (8-byte entries)
  const int64_t sub = a.chunk_len / WAVES;
  const int64_t ws = (a.chunk0 + cj) * a.chunk_len + (int64_t)wave * sub;   // first position (window's last base) of this wave
  int64_t own_lo = ws > a.begin ? ws : a.begin;
  int64_t own_hi = ws + sub;
  if (own_hi > a.end) own_hi = a.end;
  if (own_hi > a.n) own_hi = a.n;
  if (own_lo < 19) own_lo = 19;                                   // the 20-base window must fit in the stream
  if (own_lo >= own_hi) return;

  const uint2 *entries = a.entries + a.entries_off[combo];
  const int16_t *direct = a.direct + ((size_t)combo << 20);
  // the 32 bases in front of the wave's range (wave-uniform; they enter lanes 0 and 1 through the DPP shifts)
  uint32_t carry1, carry2;
  {
    const uint32_t pk = load_packed(a.packed, a.npacked, ws - 32 + 16 * (lane & 1));
    carry2 = __builtin_amdgcn_readlane(pk, 0);
    carry1 = __builtin_amdgcn_readlane(pk, 1);
  }

  // The queue of suspicious windows has two regions of 64 entries {window low word, window bits
  // 32..39 | position << 8}: one fills up while the exact-table loads of the other are in flight.
  int qn = 0, pcount = 0;                                         // wave-uniform: fill of the filling region, entries of the one in flight
  uint32_t qr = 0;                                                // wave-uniform: region being filled
  uint32_t pex = 0, pey = 0, prank = 0;                           // the lane's lookup in flight
  unsigned long long sb_next = 0;                                 // wave-uniform: reserved suspect slots not yet used
  int sb_left = 0;

  // the region in flight: its loads have landed -- XOR + popcount against the window's other fields
  auto finish = [&]() __attribute__((always_inline)) {
    if (pcount == 0) return;
    const bool on = lane < pcount;
    const u32x2 e = *lds64(QB + (qr ^ 1u) * (PAIR_QREGION * 8) + 8 * lane);
    const uint64_t W = ((uint64_t)(e.y & 0xffu) << 32) | e.x;
    const uint32_t wo = field_of(W, C) | (field_of(W, D) << 10);
    const uint64_t S = ((uint64_t)pey << 32) | pex;
    const uint32_t cnt = pey >> 28;
    const bool hit1 = sym_distance(((uint32_t)S ^ wo) & F20) <= a.k;
    const bool hit2 = cnt >= 2 && sym_distance(((uint32_t)(S >> 20) ^ wo) & F20) <= a.k;
    const bool hit3 = cnt >= 3 && sym_distance(((uint32_t)(S >> 40) ^ wo) & F20) <= a.k;
    const uint32_t what = on ? ((hit1 ? 1u : 0u) | (hit2 ? 2u : 0u) | (hit3 ? 4u : 0u) | (cnt >= 4 ? 8u : 0u)) : 0u;
    pcount = 0;
    const unsigned long long bal = __ballot(what != 0);
    if (bal == 0) return;
    // suspects go to a list for the verify kernel.  Slots are reserved PAIR_SUSPECT_BLOCK at a time per
    // wave: one atomic per suspect batch on the one shared counter serialises the chip (same-address
    // atomics take ~2.5 ns each: 2 million of them cost 6 ms here).  Unused slots are marked as holes.
    const int c = __popcll(bal);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
    int done = 0;                                                   // wave-uniform: suspects of this batch already placed
    while (done < c) {
      if (sb_left == 0) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.susp_count, (unsigned long long)PAIR_SUSPECT_BLOCK);
        sb_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base);
        sb_left = PAIR_SUSPECT_BLOCK;
      }
      const int take = c - done < sb_left ? c - done : sb_left;
      if (what && before >= done && before < done + take) {
        const unsigned long long slot = sb_next + (unsigned long long)(before - done);
        const uint64_t pw = (uint64_t)(ws + (e.y >> 8)) | ((uint64_t)combo << 40) | ((uint64_t)what << 44);
        if (slot < a.susp_cap) a.susp[slot] = make_uint4(prank, wo, (uint32_t)pw, (uint32_t)(pw >> 32));
      }
      sb_next += take; sb_left -= take; done += take;
    }
  };
  // the region just filled: rank every key, start the exact table's loads
  auto issue = [&]() __attribute__((always_inline)) {
    if (lane >= qn) return;
    const u32x2 e = *lds64(QB + qr * (PAIR_QREGION * 8) + 8 * lane);
    const uint64_t W = ((uint64_t)(e.y & 0xffu) << 32) | e.x;
    const uint32_t key = field_of(W, A) | (field_of(W, B) << 10);
    // rank of the key among the set bits: superblock (2048 bits) + block (256 bits) + words + bit
    const uint32_t word = key & 0x7fffu, bit = key >> 15;
    const uint32_t sup = *lds32(SUPER_OFF + ((word >> 6) << 2));
    const uint32_t rel = *lds16(REL_OFF + ((word >> 3) << 1));
    const u32x4 b0 = *lds128((word >> 3) << 5), b1 = *lds128(((word >> 3) << 5) + 16);
    const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const uint32_t wq = word & 7u;
    uint32_t cnt = sup + rel, part = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) {
      cnt += t < wq ? (uint32_t)__popc(bw[t]) : 0u;
      part = t == wq ? bw[t] : part;
    }
    prank = cnt + __popc(__builtin_amdgcn_ubfe(part, 0, bit));
    if (a.debug & 2) { pex = prank; pey = key; return; }           // measurement: everything but the table load
    const uint2 ent = entries[prank];
    pex = ent.x; pey = ent.y;
  };
  auto drain = [&]() __attribute__((always_inline)) {
    if (a.debug & 1) { qn = 0; return; }
    finish();
    issue();
    pcount = qn; qr ^= 1u; qn = 0;
  };

  // four blocks of 1024 bases (one packed dword per lane each) in flight per wave
  uint32_t q0 = load_packed(a.packed, a.npacked, ws + 16 * lane);
  uint32_t q1 = ws + 1024 < own_hi ? load_packed(a.packed, a.npacked, ws + 1024 + 16 * lane) : 0u;
  uint32_t q2 = ws + 2048 < own_hi ? load_packed(a.packed, a.npacked, ws + 2048 + 16 * lane) : 0u;
  uint32_t q3 = ws + 3072 < own_hi ? load_packed(a.packed, a.npacked, ws + 3072 + 16 * lane) : 0u;

  const uint32_t m555 = 0x555u, moff = 0x1ffffeu;                  // (v_bitop3 takes no literal: constants in SGPRs)
  // consume stage of half a block (windows 8H .. 8H+7 of the block whose stream words are v2 : v1 : vc):
  // its table entries E have landed.  Returns the suspicious windows, bit j = window 8H + j.
  auto consume = [&](auto HALF, uint32_t v2, uint32_t v1, uint32_t vc, const int32_t (&E)[8]) __attribute__((always_inline)) -> uint32_t {
    constexpr int H = decltype(HALF)::value;
    uint32_t sacc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * H + j;
      uint32_t w = 0;                                               // six bases of the window's other fields at bits 0..11
#define PM_PAIR_OTHER(I)                                                                                        \
      if (i == I) {                                                                                                 \
        const uint32_t X = bits_at<2 * I + 26 + 10 * C>(v2, v1, vc);                                                \
        if constexpr (D == C + 1) w = X;                                                                            \
        else { const uint32_t Y = bits_at<2 * I + 26 + 10 * D - 10>(v2, v1, vc); w = (X & 0x3ffu) | (Y & ~0x3ffu); } \
      }
      PM_PAIR_OTHER(0) PM_PAIR_OTHER(1) PM_PAIR_OTHER(2) PM_PAIR_OTHER(3) PM_PAIR_OTHER(4) PM_PAIR_OTHER(5) PM_PAIR_OTHER(6) PM_PAIR_OTHER(7)
      PM_PAIR_OTHER(8) PM_PAIR_OTHER(9) PM_PAIR_OTHER(10) PM_PAIR_OTHER(11) PM_PAIR_OTHER(12) PM_PAIR_OTHER(13) PM_PAIR_OTHER(14) PM_PAIR_OTHER(15)
#undef PM_PAIR_OTHER
      const uint32_t x = (uint32_t)E[j] ^ w;
      // substitutions on the six bases, minus k + 1 (the entry's top four bits, sign-extended: -(k+1), or
      // -8 when several patterns share the key): negative = suspicious; the sign bits are funnelled
      // into sacc (v_bitop3 (a | b) & c, v_bcnt with accumulator, v_alignbit: no compare)
      const int z = __popc(__builtin_amdgcn_bitop3_b32(x, x >> 1, m555, 0xa8)) + (E[j] >> 12);
      sacc = __builtin_amdgcn_alignbit(sacc, (uint32_t)z, 31);
    }
    return __brev(sacc) >> 24;
  };

  // the suspicious windows `rem` of the block at bbase (stream words v2 : v1 : vc) into the wave's
  // queue: one per lane and round (ballot + mbcnt give the slots)
  auto compact = [&](uint32_t v2, uint32_t v1, uint32_t vc, int64_t bbase, uint32_t rem) __attribute__((always_inline)) {
    if (a.debug & 4) return;
    const uint32_t prel0 = (uint32_t)(bbase - ws) + 16 * (uint32_t)lane;
    for (;;) {
      const unsigned long long bal = __ballot(rem != 0);
      if (bal == 0) break;
      const int s = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (rem != 0 && s < PAIR_QREGION) {                          // lanes the region has no room for keep their window for the next round
        const int i = __ffs(rem) - 1;
        rem &= rem - 1;
        const uint32_t sft = 2 * (uint32_t)i + 26;                   // 26 .. 56: first window bit in v2 : v1 : vc
        const uint32_t x0 = __builtin_amdgcn_alignbit(v1, v2, sft), x1 = __builtin_amdgcn_alignbit(vc, v1, sft), x2 = vc >> (sft & 31u);
        const uint32_t wlo = sft < 32 ? x0 : x1, whi = (sft < 32 ? x1 : x2) & 0xffu;
        u32x2 e;
        e.x = wlo; e.y = whi | ((prel0 + (uint32_t)i) << 8);
        *lds64(QB + qr * (PAIR_QREGION * 8) + 8 * s) = e;
      }
      qn += __popcll(bal);
      if (qn >= PAIR_QREGION) { qn = PAIR_QREGION; drain(); }
    }
  };

  // test stage of half a block: keys, bitmap bits (into acc, funnelled from the top), and the direct
  // table's loads into E
  auto test = [&](auto HALF, uint32_t prev2, uint32_t prev1, uint32_t cur, int32_t (&E)[8], uint32_t &acc) __attribute__((always_inline)) {
    constexpr int H = decltype(HALF)::value;
    uint32_t ks[8], wd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * H + j;
      uint32_t K4 = 0;                                              // the key at bits 2..21
      // window i starts at bit 2i + 26 of prev2 : prev1 : cur; field f at + 10f
#define PM_PAIR_KEY(I)                                                                                         \
      if (i == I) {                                                                                                \
        const uint32_t X = bits_at<2 * I + 26 + 10 * A - 2>(prev2, prev1, cur);                                    \
        if constexpr (B == A + 1) K4 = X;                                                                          \
        else { const uint32_t Y = bits_at<2 * I + 26 + 10 * B - 12>(prev2, prev1, cur); K4 = (X & 0xffcu) | (Y & ~0xffcu); } \
      }
      PM_PAIR_KEY(0) PM_PAIR_KEY(1) PM_PAIR_KEY(2) PM_PAIR_KEY(3) PM_PAIR_KEY(4) PM_PAIR_KEY(5) PM_PAIR_KEY(6) PM_PAIR_KEY(7)
      PM_PAIR_KEY(8) PM_PAIR_KEY(9) PM_PAIR_KEY(10) PM_PAIR_KEY(11) PM_PAIR_KEY(12) PM_PAIR_KEY(13) PM_PAIR_KEY(14) PM_PAIR_KEY(15)
#undef PM_PAIR_KEY
      ks[j] = K4;
      wd[j] = *lds32(K4 & 0x1fffcu);
    }
    __builtin_amdgcn_sched_barrier(0);                              // issue the eight reads before the first verdict waits
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t v = wd[j] >> ((ks[j] >> 17) & 31u);            // bits 15..19 of the key pick the bit
      acc = __builtin_amdgcn_alignbit(v, acc, 1);
      // table index = key for a survivor, 0 otherwise (byte offset 2 * key from the key at bits 2..21)
      const uint32_t off = __builtin_amdgcn_bitop3_b32(ks[j] >> 1, (uint32_t)__builtin_amdgcn_sbfe((int)v, 0, 1), moff, 0x80);   // three-way AND
      E[j] = *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(direct) + off);   // plain load: nontemporal ran 3x, sc1 1.7x slower
    }
  };

  // Software pipeline, one block deep: the entries a half block's test stage loads are consumed
  // after the same half of the NEXT block has been tested -- about 300 instructions of this wave (and
  // as many of each of the three other waves of its SIMD) later; with half a block of distance the
  // table loads (the L1 fill path runs saturated: one 128-byte line per lookup) were still waited for.
  // Block parity is compile time (two entry buffers per half, two sets of block state).
  int32_t E[2][2][8];                                               // [block parity][half][window]
  uint32_t sv2[2] = {0, 0}, sv1[2] = {0, 0}, svc[2] = {0, 0}, srem[2] = {0, 0};   // per parity: stream words, bitmap survivors
  int64_t sbb[2] = {ws, ws};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 8; ++i) E[q][hh][i] = 0;
  const std::integral_constant<int, 0> H0;
  const std::integral_constant<int, 1> H1;
  int64_t bb = ws;
  bool have_prev = false;                                           // wave-uniform
  auto block = [&](auto PAR) __attribute__((always_inline)) {
    constexpr int P = decltype(PAR)::value, O = P ^ 1;
    const uint32_t cur = q0;
    q0 = q1; q1 = q2; q2 = q3;
    if (bb + 4096 < own_hi) q3 = load_packed(a.packed, a.npacked, bb + 4096 + 16 * lane);
    // the two dwords in front of every lane's own: whole-wave shifts by one lane (lane 0 takes the carry)
    const uint32_t prev1 = __builtin_amdgcn_update_dpp(carry1, cur, 0x138, 0xf, 0xf, false);       // wave_shr:1
    const uint32_t prev2 = __builtin_amdgcn_update_dpp(carry2, prev1, 0x138, 0xf, 0xf, false);     // lane 1 takes carry1 from lane 0 of prev1
    carry2 = __builtin_amdgcn_readlane(cur, 62);
    carry1 = __builtin_amdgcn_readlane(cur, 63);
    uint32_t own = 0xffffu;
    if (bb < own_lo || bb + 1024 > own_hi) {                       // wave-uniform: edge blocks only
      const int64_t pbase = bb + 16 * lane;
      const int64_t lo = own_lo - pbase, hi = own_hi - pbase;
      const uint32_t l = lo <= 0 ? 0u : (lo >= 16 ? 16u : (uint32_t)lo), hh = hi <= 0 ? 0u : (hi >= 16 ? 16u : (uint32_t)hi);
      own = ((1u << hh) - 1u) & ~((1u << l) - 1u);
    }
    uint32_t acc = 0, slow = 0;
    test(H0, prev2, prev1, cur, E[P][0], acc);
    if (have_prev) slow = consume(H0, sv2[O], sv1[O], svc[O], E[O][0]);
    test(H1, prev2, prev1, cur, E[P][1], acc);
    if (have_prev) {
      slow |= consume(H1, sv2[O], sv1[O], svc[O], E[O][1]) << 8;
      compact(sv2[O], sv1[O], svc[O], sbb[O], slow & srem[O]);
    }
    sv2[P] = prev2; sv1[P] = prev1; svc[P] = cur; srem[P] = (acc >> 16) & own; sbb[P] = bb;
    have_prev = true;
    bb += 1024;
  };
  const std::integral_constant<int, 0> P0;
  const std::integral_constant<int, 1> P1;
  bool last_odd = false;                                            // parity of the last block tested
  while (bb < own_hi) {
    block(P0); last_odd = false;
    if (bb >= own_hi) break;
    block(P1); last_odd = true;
  }
  if (have_prev) {                                                  // the last block's entries
    if (last_odd) {
      const uint32_t slow = consume(H0, sv2[1], sv1[1], svc[1], E[1][0]) | (consume(H1, sv2[1], sv1[1], svc[1], E[1][1]) << 8);
      compact(sv2[1], sv1[1], svc[1], sbb[1], slow & srem[1]);
    } else {
      const uint32_t slow = consume(H0, sv2[0], sv1[0], svc[0], E[0][0]) | (consume(H1, sv2[0], sv1[0], svc[0], E[0][1]) << 8);
      compact(sv2[0], sv1[0], svc[0], sbb[0], slow & srem[0]);
    }
  }
  if (qn) drain();
  finish();
  if (lane < sb_left && sb_next + lane < a.susp_cap) a.susp[sb_next + lane] = make_uint4(PAIR_SUSPECT_HOLE, 0, 0, 0);
}

__global__ __launch_bounds__(PAIR_THREADS) void pm_pair_scan(PairArgs a) {
  extern __shared__ uint32_t lds[];
  // blockIdx -> (combo, chunk): runs of `group` chunks share a combo, all combos of a superchunk
  // follow each other (the superchunk's stream is re-read from MALL, a combo's table stays in L2)
  const int per_super = a.group * a.ncombos;
  const int sc = blockIdx.x / per_super;
  const int rem = blockIdx.x - sc * per_super;
  int combo = rem / a.group;
  int cj = sc * a.group + (rem - combo * a.group);
  const int full = (a.nchunks / a.group) * a.group;               // last, shorter superchunk
  if (sc * a.group >= full) {
    const int tail = a.nchunks - full;
    const int r2 = blockIdx.x - (full / a.group) * per_super;
    combo = r2 / tail;
    cj = full + (r2 - combo * tail);
  }
  if (cj >= a.nchunks || combo >= a.ncombos) return;
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.image + (size_t)combo * PAIR_IMAGE_WORDS);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < PAIR_IMAGE_WORDS / 4; i += PAIR_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const int fa = a.fa[combo], fb = a.fb[combo];
  switch (fa * 4 + fb) {                                            // wave-uniform
    case 1: pair_scan_body<0, 1>(a, combo, cj); break;
    case 2: pair_scan_body<0, 2>(a, combo, cj); break;
    case 3: pair_scan_body<0, 3>(a, combo, cj); break;
    case 6: pair_scan_body<1, 2>(a, combo, cj); break;
    case 7: pair_scan_body<1, 3>(a, combo, cj); break;
    case 11: pair_scan_body<2, 3>(a, combo, cj); break;
    default: break;
  }
}

}  // namespace

// ---- host side: tables, launch -------------------------------------------------------------------

std::string pair_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids, const Alphabet &alpha, int k,
                       int eos_code, PairTables *out) {
  PairTables &t = *out;
  t = PairTables();
  if (k < 1 || k > 2) return "the pair plan is built for k = 1 and k = 2";
  const bool norm = alpha.nch['A'] == 0 && alpha.nch['C'] == 1 && alpha.nch['G'] == 2 && alpha.nch['T'] == 3;
  const bool ascii = alpha.size == 256 && alpha.nch['A'] == 'A' && alpha.nch['C'] == 'C' && alpha.nch['G'] == 'G' && alpha.nch['T'] == 'T';
  if (!norm && !ascii) return "stream alphabet is neither A,C,G,T-normalized nor raw ASCII";
  t.ascii = ascii && !norm;
  t.k = k;
  t.eos_code = (eos_code >= 0 && eos_code < 256) ? eos_code : -1;
  // 2-bit code of a base as pack_stream (pm_seed.hip) derives it from the stream byte
  auto base2 = [&](unsigned char ch) -> int {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return t.ascii ? 3 : 2; case 'T': return t.ascii ? 2 : 3; }
    return -1;
  };
  const size_t np = pats.size();
  if (np == 0) return "no patterns";
  if (np >= ((size_t)1 << 30)) return "too many patterns";
  t.pat40.resize(np); t.pat_len.resize(np); t.pat_id.resize(np); t.pat_codes.assign(np * 32, 0); t.pat_zone.assign(np, 0);
  for (size_t j = 0; j < np; ++j) {
    const std::string &s = pats[j].s;
    const int L = (int)s.size();
    if (L < 20 || L > 32) return "the pair plan needs patterns of 20..32 characters";
    uint64_t w = 0;
    for (int i = 0; i < 20; ++i) {
      const int b2 = base2((unsigned char)s[L - 20 + i]);
      if (b2 < 0) return "pattern with characters other than A,C,G,T";
      w |= (uint64_t)b2 << (2 * i);
    }
    for (int i = 0; i < L; ++i) {
      if (base2((unsigned char)s[i]) < 0) return "pattern with characters other than A,C,G,T";
      t.pat_codes[j * 32 + i] = (uint8_t)alpha.nch[(unsigned char)s[i]];
    }
    t.pat40[j] = w; t.pat_len[j] = (uint8_t)L; t.pat_id[j] = ids[j];
    {
      const int es = std::max(0, std::min(L, pats[j].esb)), ee = std::max(0, std::min(L, pats[j].eeb));
      uint32_t z = 0;
      for (int i = 0; i < L; ++i) if (i < es || i >= L - ee) z |= 1u << i;
      t.pat_zone[j] = z;
    }
    t.maxlen = std::max(t.maxlen, L);
  }
  // field pairs in the order that decides who reports a pair found several times
  if (k == 1) { t.ncombos = 2; t.fa[0] = 0; t.fb[0] = 1; t.fa[1] = 2; t.fb[1] = 3; }
  else { t.ncombos = 0; for (int x = 0; x < 4; ++x) for (int y = x + 1; y < 4; ++y) { t.fa[t.ncombos] = x; t.fb[t.ncombos] = y; ++t.ncombos; } }
  const int C = t.ncombos;
  t.image.assign((size_t)C * PAIR_IMAGE_WORDS, 0);
  t.order.assign((size_t)C * np, 0);
  t.direct.assign((size_t)C << 20, 0);
  std::vector<std::vector<uint32_t>> ent(C), fp(C);
  auto build_combo = [&](int ci) {
    const int fa = t.fa[ci], fb = t.fb[ci];
    int fc, fd;
    other_fields(fa, fb, &fc, &fd);
    // sort key = position of the key's bit in the bitmap (word major), then pattern index
    std::vector<uint64_t> srt(np);
    for (size_t j = 0; j < np; ++j) {
      const uint32_t key = field_of(t.pat40[j], fa) | (field_of(t.pat40[j], fb) << 10);
      const uint32_t bitpos = ((key & 0x7fffu) << 5) | (key >> 15);
      srt[j] = ((uint64_t)bitpos << 32) | (uint64_t)j;
    }
    std::sort(srt.begin(), srt.end());
    uint32_t *img = &t.image[(size_t)ci * PAIR_IMAGE_WORDS];
    uint32_t *ord = &t.order[(size_t)ci * np];
    int16_t *dir = &t.direct[(size_t)ci << 20];
    std::vector<uint32_t> &e = ent[ci], &f = fp[ci];
    e.reserve(2 * np); f.reserve(np + 1);
    for (size_t j = 0; j < np;) {
      const uint32_t bitpos = (uint32_t)(srt[j] >> 32);
      size_t j2 = j;
      while (j2 < np && (uint32_t)(srt[j2] >> 32) == bitpos) ++j2;
      img[bitpos >> 5] |= 1u << (bitpos & 31u);
      f.push_back((uint32_t)j);
      uint32_t o[3] = {0, 0, 0};
      for (size_t q = j; q < j2; ++q) {
        const uint32_t pi = (uint32_t)srt[q];
        ord[q] = pi;
        if (q - j < 3) o[q - j] = field_of(t.pat40[pi], fc) | (field_of(t.pat40[pi], fd) << 10);
      }
      const uint64_t slot = slot_pack(o[0], o[1], o[2], (uint32_t)std::min<size_t>(j2 - j, 4));
      const uint32_t ex = (uint32_t)slot;
      e.push_back(ex); e.push_back((uint32_t)(slot >> 32));
      // direct-mapped by key: six bases of the first pattern's other fields; top four bits = -(k + 1), or -8 when
      // further patterns share the key (what the consume stage adds to its mismatch count: negative = suspicious)
      const uint32_t key = (bitpos >> 5) | ((bitpos & 31u) << 15);
      dir[key] = (int16_t)(uint16_t)((ex & 0xfffu) | (j2 - j > 1 ? 0x8000u : ((uint32_t)(16 - (k + 1)) << 12)));
      j = j2;
    }
    f.push_back((uint32_t)np);
    // rank directory: set bits before every 2048-bit superblock (u32) and, inside it, before every 256-bit block (u16)
    uint32_t *sup = img + PAIR_BITMAP_WORDS;
    uint16_t *rel = reinterpret_cast<uint16_t *>(img + PAIR_BITMAP_WORDS + PAIR_SUPER);
    uint32_t run = 0;
    for (int sb = 0; sb < PAIR_SUPER; ++sb) {
      sup[sb] = run;
      uint32_t in = 0;
      for (int b = 0; b < 8; ++b) {
        rel[sb * 8 + b] = (uint16_t)in;
        for (int w = 0; w < 8; ++w) in += (uint32_t)__builtin_popcount(img[sb * 64 + b * 8 + w]);
      }
      run += in;
    }
  };
  if (np * (size_t)C < 20000) for (int ci = 0; ci < C; ++ci) build_combo(ci);
  else {
    std::vector<std::thread> th;
    for (int ci = 0; ci < C; ++ci) th.emplace_back([&, ci]() { build_combo(ci); });
    for (std::thread &x : th) x.join();
  }
  for (int ci = 0; ci < C; ++ci) {
    t.entries_off[ci] = t.entries.size() / 2;
    t.entries.insert(t.entries.end(), ent[ci].begin(), ent[ci].end());
    t.first_off[ci] = t.first_pat.size();
    t.first_pat.insert(t.first_pat.end(), fp[ci].begin(), fp[ci].end());
  }
  return "";
}

hipError_t pair_upload(const PairTables &t, PairDevice *d, hipStream_t st) {
  pair_free(d);
  d->k = t.k; d->maxlen = t.maxlen; d->ncombos = t.ncombos; d->eos_code = t.eos_code; d->ascii = t.ascii; d->np = t.pat40.size();
  for (int c = 0; c < PAIR_MAX_COMBOS; ++c) { d->fa[c] = t.fa[c]; d->fb[c] = t.fb[c]; d->entries_off[c] = t.entries_off[c]; d->first_off[c] = t.first_off[c]; }
  auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
  };
  hipError_t e;
  if ((e = up(t.image.data(), t.image.size() * 4, (void **)&d->image)) != hipSuccess) return e;
  if ((e = up(t.entries.data(), t.entries.size() * 4, (void **)&d->entries)) != hipSuccess) return e;
  if ((e = up(t.direct.data(), t.direct.size() * 2, (void **)&d->direct)) != hipSuccess) return e;
  if ((e = up(t.first_pat.data(), t.first_pat.size() * 4, (void **)&d->first_pat)) != hipSuccess) return e;
  if ((e = up(t.order.data(), t.order.size() * 4, (void **)&d->order)) != hipSuccess) return e;
  if ((e = up(t.pat40.data(), t.pat40.size() * 8, (void **)&d->pat40)) != hipSuccess) return e;
  if ((e = up(t.pat_len.data(), t.pat_len.size(), (void **)&d->pat_len)) != hipSuccess) return e;
  if ((e = up(t.pat_id.data(), t.pat_id.size() * 4, (void **)&d->pat_id)) != hipSuccess) return e;
  if ((e = up(t.pat_codes.data(), t.pat_codes.size(), (void **)&d->pat_codes)) != hipSuccess) return e;
  if ((e = up(t.pat_zone.data(), t.pat_zone.size() * 4, (void **)&d->pat_zone)) != hipSuccess) return e;
  // the kernel addresses the bitmap from LDS address 0: it must have no static LDS in front of the dynamic block
  hipFuncAttributes fa;
  if ((e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(pm_pair_scan))) != hipSuccess) return e;
  if (fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(pm_pair_scan), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES)) != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

void pair_free(PairDevice *d) {
  void *ptrs[] = {d->image, d->entries, d->direct, d->first_pat, d->order, d->pat_id, d->pat40, d->pat_len, d->pat_codes, d->pat_zone};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  *d = PairDevice();
}

ScanGeometry pair_geometry(const PairDevice &d, int64_t begin, int64_t end) {
  ScanGeometry g;
  int64_t chunk = 1 << 19;                                         // 512 Ki positions per workgroup
  // large ranges: 2 Mi positions per workgroup amortise staging the 146 KiB LDS image
  if ((end - begin) / ((int64_t)1 << 21) * d.ncombos >= 256 * 8) chunk = (int64_t)1 << 21;
  if (const char *env = getenv("PM_SEED_CHUNK")) {                 // test knob (shared with the seed kernels)
    const int64_t v = atoll(env);
    if (v >= 1024 * WAVES) chunk = v / (1024 * WAVES) * (1024 * WAVES);
  }
  g.seg_len = chunk;
  const int64_t c_lo = begin / chunk, c_hi = end > begin ? (end - 1) / chunk : c_lo - 1;
  g.nseg = (int)(c_hi - c_lo + 1);
  g.threads = PAIR_THREADS;
  g.blocks = g.nseg * d.ncombos;
  return g;
}

hipError_t pair_launch(const PairDevice &d, const uint8_t *d_text, const uint32_t *d_packed, int64_t n, int64_t begin, int64_t end,
                       pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, void *d_susp, unsigned long long *d_susp_count, uint64_t susp_cap,
                       hipStream_t st, ScanGeometry *geo_out) {
  if (!d_packed) return hipErrorInvalidValue;
  if (end > n) end = n;
  ScanGeometry g = pair_geometry(d, begin, end);
  if (geo_out) *geo_out = g;
  if (g.blocks <= 0 || d.np == 0) return hipSuccess;
  PairArgs a;
  memset(&a, 0, sizeof(a));
  a.text = d_text; a.n = n; a.begin = begin; a.end = end;
  a.packed = d_packed; a.npacked = (n + 15) / 16;
  a.chunk_len = g.seg_len; a.chunk0 = begin / g.seg_len; a.nchunks = g.nseg; a.ncombos = d.ncombos;
  a.group = 256;
  if (const char *env = getenv("PM_SEED_GROUP")) { const int v = atoi(env); if (v > 0) a.group = v; }
  a.k = d.k; a.eos_code = d.eos_code;
  if (const char *env = getenv("PM_SEED_DEBUG")) a.debug = atoi(env);
  for (int c = 0; c < PAIR_MAX_COMBOS; ++c) {
    a.fa[c] = d.fa[c]; a.fb[c] = d.fb[c];
    a.entries_off[c] = (uint32_t)d.entries_off[c]; a.first_off[c] = (uint32_t)d.first_off[c];
  }
  a.image = d.image; a.direct = d.direct; a.entries = reinterpret_cast<const uint2 *>(d.entries); a.first_pat = d.first_pat; a.order = d.order;
  a.np = (uint32_t)d.np;
  a.pat40 = reinterpret_cast<const uint2 *>(d.pat40); a.pat_len = d.pat_len; a.pat_id = d.pat_id; a.pat_codes = d.pat_codes; a.pat_zone = d.pat_zone; a.viol_level = d.viol_level;
  a.out = d_out; a.counter = d_counter; a.cap = cap;
  if (!d_susp || !d_susp_count) return hipErrorInvalidValue;
  a.susp = reinterpret_cast<uint4 *>(d_susp); a.susp_count = d_susp_count; a.susp_cap = susp_cap;   // *d_susp_count zeroed by the caller (stream order)
  hipLaunchKernelGGL(pm_pair_scan, dim3(g.blocks), dim3(PAIR_THREADS), PAIR_LDS_BYTES, st, a);
  hipError_t ce = hipGetLastError();
  if (ce != hipSuccess) return ce;
  hipLaunchKernelGGL(pm_pair_verify, dim3(256 * 16), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace pm

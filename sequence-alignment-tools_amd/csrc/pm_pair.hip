// pm_pair.hip -- "pair plan" scan kernels for gfx950: exact 20-bit key bitmap in LDS, and ONE 8-byte
// lookup in an L2-resident slot table that settles a key hit.
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
//   * a WORKGROUP owns one (field pair, stream chunk): it stages that pair's key bitmap in LDS (key
//     bits 0..14 = ROW = dword, bits 15..19 = bit), then streams its chunk of the 2-bit packed stream;
//   * a LANE owns 16 consecutive window positions per block of 1024; its two predecessors' dwords
//     come in by two whole-wave DPP shifts;
//       pass A   for every window, straight-line code with compile-time offsets: key (v_lshrrev /
//                v_alignbit, v_bfi for separated fields), one ds_read_b32 of the key's ROW of the bitmap,
//                the key's bit funnelled into a mask of the lane's pending key hits (5 instructions);
//       pass B   only for the key hits (17 % of the windows at 200k patterns), in ROUNDS: per round every
//                lane takes its oldest pending hit (hits of two blocks are pending at a time).  The row
//                gives, besides the key's bit, its rank inside the row (popcount of the lower bits): the
//                keys that occur own consecutive 8-byte slots of a table in L2 -- `stride` slots per row --
//                so (row, rank in row) addresses the key's slot without any rank directory: one
//                global_load_dwordx2.  The slot holds the OTHER 20 window bits of up to three patterns
//                with this key; three XOR + popcount against the window's own other fields decide the
//                window, five rounds later.  What survives is a true candidate on the packed bases
//                (1e-3 of the positions), a key with more than three patterns, or a key beyond its
//                row's slots (0.3 % of the keys at 200k patterns: row occupancy is Poisson);
//   * those few are queued per wave in LDS and leave in batches for a suspect list; a second kernel,
//     pm_pair_verify, works out rank and slot again and reads the raw stream bytes for the exact
//     distance (N = mismatch, EOS = reject) with every lane busy;
//   * a (window, pattern) pair that agrees on several field pairs is reported by the first of them
//     in the plan's list.
//
// What bounds it (3 Gbp, 200k patterns, k = 2; scripts/probe/, profiles/r03_*): VALU issue.  Most of the
// integer instructions this code is made of (every VOP3 form, v_bcnt, v_bfe, v_alignbit, left shifts,
// anything with an SGPR operand) issue at one wave64 instruction per ~4 cycles and SIMD, half the rate of
// v_and / v_add / right shifts: 7.2e9 of them per launch = 0.78 of that capacity.  Beside it every key hit
// (17.4 % of 1.8e10 tests) is a random 128-byte line from L2 into a CU's L1 -- that path moves one line per
// ~2 cycles and CU, 269 G lines/s for the chip, whatever the bytes asked for: 3.1e9 lines = 0.68 of it.
// Round 2's form paid a second such line (rank-indexed exact table) for the 2.3 % of the windows its 2-byte
// direct table could not settle, and did the rank / address / compare work branch-free for all windows.
#include "pm_internal.h"
#include "pm_pair.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <type_traits>
#include <utility>

namespace pm {

namespace {

constexpr int WAVES = PAIR_WAVES;
constexpr uint32_t F20 = 0xfffffu;
constexpr uint32_t WHAT_REST = 8u, WHAT_ALL = 16u;                   // suspect: walk the key's patterns from the fourth / from the first
// Slot of a key, 8 bytes: the other 20 window bits of up to three patterns that have the key
// (o1 | o2 << 20 | o3 << 40; with fewer than three patterns the free fields repeat o1) and bit 63 =
// "not settled by these three": more patterns share the key, or -- last slot of every row -- the key
// lies beyond its row's slots.  It is the high dword's sign bit: the consume stage ORs that dword into its
// three "within k" verdicts, whose sign is what it tests (bits 60..62 stay zero).  (Until round 3's fuzz the flag
// went into the third count as -8: with -K 1 a window whose other ten bases all differ from the slot's third
// pattern -- every window without an A, on a row's overflow marker -- came out at 10 - 8 - 2 = 0, not suspicious.)
__device__ __host__ __forceinline__ uint64_t slot_pack(uint32_t o1, uint32_t o2, uint32_t o3, bool walk) {
  return (uint64_t)o1 | ((uint64_t)o2 << 20) | ((uint64_t)o3 << 40) | ((uint64_t)(walk ? 8 : 0) << 60);
}

struct PairArgs {
  const uint8_t *text;
  int64_t n, begin, end;                // owned hit ends: begin < end_pos <= end
  const uint32_t *packed;               // the stream, 2 bits per base, 16 bases per dword
  int64_t npacked;
  int64_t chunk0, chunk_len;            // first chunk index (absolute), positions per workgroup
  int nchunks, ncombos, group;
  int k, eos_code, debug;
  int stride;                           // slots per row of the slot table (the last one is the row's overflow marker)
  int fa[PAIR_MAX_COMBOS], fb[PAIR_MAX_COMBOS];
  const uint32_t *image;                // [combo][PAIR_BITMAP_WORDS]: key bitmap
  const uint2 *slots;                   // [combo][PAIR_BITMAP_WORDS * stride]
  const uint32_t *row_base;             // [combo][PAIR_BITMAP_WORDS + 1]: distinct keys in front of every row
  const uint32_t *first_pat, *order;    // [combo][distinct keys + 1] (by rank), [combo][np]: patterns sorted by key
  const uint32_t *olist;                // [combo][np]: the OTHER 20 window bits of order[]'s patterns, in the same sequence (a key's run is contiguous)
  uint32_t first_off[PAIR_MAX_COMBOS];
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
  uint4 *susp;                          // suspect records for pm_pair_verify: {key, other fields, position | combo << 40}
  unsigned long long *susp_count;
  unsigned long long susp_cap;
  uint64_t *seed_out;                   // edit plan (pm_pair_edit_resolve): 8-byte seed records pattern index << 40 | position for pm_edits_verify
  unsigned long long *seed_count;
  unsigned long long seed_cap;
  unsigned long long *stats;            // measurement (debug & 32): [0] blocks of 1024 positions, [1] rounds, [2] key hits, over waves and field pairs
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u128;
__device__ __forceinline__ lds_u128 *lds128(uint32_t addr) { return reinterpret_cast<lds_u128 *>((uintptr_t)addr); }
__device__ __forceinline__ lds_u32 *lds32(uint32_t addr) { return reinterpret_cast<lds_u32 *>((uintptr_t)addr); }

__device__ __forceinline__ uint32_t load_packed(const uint32_t *packed, int64_t npacked, int64_t pos) {
  const int64_t i = pos >> 4;
  if (pos < 0 || i >= npacked) return 0u;
  return packed[i];
}

// f(integral_constant<int, 0>()), f(integral_constant<int, 1>()), ... in order
template <int... Is, typename F>
__device__ __forceinline__ void for_windows(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>()), ...); }

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


// ---- edit distance on the pair geometry -------------------------------------------------------
// A match with <= 2 edits (substitution, insertion, deletion) of a pattern's last 20 bases leaves two of the four fields
// untouched; they sit in the text 5 (B - A) + d bases apart, d = net insertions between them, |d| <= 2.  14 tests
// (A, B, d) cover every placement of two edits (scripts/edit_pair_cover.py).  A key hit of test (A, B, d) fixes a frame --
// field B where the window has it, field A displaced by -d -- and what remains is a decision about the OTHER ten bases:
// can they be brought to the text with the edits that are left?  `edit_cost` is a necessary condition for that, cheap enough
// to run for every key hit; the automaton (pm_edits_verify, shift_and_inexact.cc:249-352) decides exactly afterwards.
//
// In the frame, pattern base j sits at text base j + s(j).  s = 0 on field B and -d on field A; an insertion raises s by one
// for the bases behind it (towards the pattern's end), a deletion lowers it and leaves the deleted base without a partner, a
// substitution leaves one base without a partner.  The other fields fall into regions: L (in front of A), M (between A and
// B), R (behind B).  Within a region with i insertions and e deletions the bases take i + e + 1 consecutive shifts, and at
// most (substitutions + e) of them match at none of those shifts.  So, ORDER IGNORED (which only lets more through):
//     edits in the region >= i + max(e, u),   u = bases of the region that match at no shift of the set,
// minimised over the shift sets the remaining budget allows, and the regions' minima add up (the budget is one sum).
// Mixed sets (an insertion and a deletion in one region) are dominated by the one-sided sets of the same span.
//   R, base shift 0:            {0}: u | {0,+1}: 1 + u | {0,-1}: max(1, u) | {0,+1,+2}: 2 + u | {0,-1,-2}: 2 if two_deletions
//   L, base shift -d (mirror):  {0}: u | {+1}: max(1, u) | {-1}: 1 + u | {+1,+2}: 2 if two_deletions | {-1,-2}: 2 + u   (relative to -d)
//   M, from -d at A to 0 at B:  d = 0: u | 1 + max(1, u{0,+1}) | 1 + max(1, u{0,-1});  d = 1: 1 + u{-1,0};  d = -1: max(1, u{1,0});
//                               d = 2: 2 + u{-2,-1,0};  d = -2: 2 if two_deletions
// (two deletions in one region leave no budget, so their ORDER can be checked: see two_deletions)
// Window bits: base t of the text, t = -2 .. 21 relative to the 20-base window whose last base is the lane's position, at bits
// 2 (t + 2) of whi : wlo (2 bits per base; N and end-of-sequence alias to a base there, which only lets more through).
template <int OFF>
__device__ __forceinline__ uint32_t wbits(uint32_t wlo, uint32_t whi) {
  static_assert(OFF >= 0 && OFF < 48, "window offset");
  if constexpr (OFF == 0) return wlo;
  else if constexpr (OFF < 32) return __builtin_amdgcn_alignbit(whi, wlo, OFF);
  else return whi >> (OFF - 32);
}
// one flag per base (even bit positions): NB pattern bases p against the text bases T .. T + NB - 1
template <int NB, int T>
__device__ __forceinline__ uint32_t mism(uint32_t p, uint32_t wlo, uint32_t whi) {
  static_assert(T >= -2 && T + NB <= 22, "text base outside the window");
  const uint32_t x = p ^ wbits<2 * (T + 2)>(wlo, whi);
  return (x | (x >> 1)) & (NB == 5 ? 0x155u : 0x55555u);
}
enum { REG_L = 0, REG_M = 1, REG_R = 2 };
// Two DELETIONS in one region use up the budget, so there the order is known and cheap to check: the region's bases form
// three runs -- at the shift of its low end, one less, two less (towards the high end) -- with the two deleted bases between
// them.  `lowm`, `midm`, `highm` = mismatch flags at those three shifts.  The longest clean run from the low end and the
// longest from the high end leave the smallest middle: it must be clean at the middle shift.  (Without this the tier was
// "at most two bases match at none of three shifts": 14 % of all random ten-base regions, half of all suspects.)
template <int NB>
__device__ __forceinline__ bool two_deletions(uint32_t lowm, uint32_t midm, uint32_t highm) {
  const int lo = __builtin_ctz(lowm | (1u << (2 * NB)));           // first base (its flag bit) that does not match at the low end's shift: a deleted one
  const int hi = 31 - __builtin_clz(highm | 1u);                    // last base that does not match at the high end's shift (bit 0: none or base 0)
  const uint32_t between = ((1u << hi) - 1u) & ~((2u << lo) - 1u);  // flag bits strictly between them (empty when hi <= lo)
  return (midm & between) == 0;
}
// cost of one region: NB bases from pattern base T0 on, base shift BASE, remaining budget BUDGET (shift sets that need more
// indels than that are not tried -- they would also reach outside the window)
template <int KIND, int NB, int T0, int BASE, int DSP, int BUDGET>
__device__ __forceinline__ int region_cost(uint32_t p, uint32_t wlo, uint32_t whi) {
  const uint32_t m0 = mism<NB, T0 + BASE>(p, wlo, whi);
  if constexpr (KIND == REG_M) {
    if constexpr (DSP == 0) {
      int c = __popc(m0);
      if constexpr (BUDGET >= 2) {
        const int up = 1 + max(1, __popc(m0 & mism<NB, T0 + 1>(p, wlo, whi))), dn = 1 + max(1, __popc(m0 & mism<NB, T0 - 1>(p, wlo, whi)));
        c = min(c, min(up, dn));
      }
      return c;
    }
    else if constexpr (DSP == 1) return 1 + __popc(m0 & mism<NB, T0 - 1>(p, wlo, whi));
    else if constexpr (DSP == -1) return max(1, __popc(m0 & mism<NB, T0 + 1>(p, wlo, whi)));
    else if constexpr (DSP == 2) return 2 + __popc(m0 & mism<NB, T0 - 1>(p, wlo, whi) & mism<NB, T0 - 2>(p, wlo, whi));
    else return two_deletions<NB>(mism<NB, T0 + 2>(p, wlo, whi), mism<NB, T0 + 1>(p, wlo, whi), m0) ? 2 : 9;   // (A's side, the low end, is two ahead)
  } else {
    // R: +1 = insertion, -1 = deletion; L (scanned away from A, towards the pattern's start): +1 = deletion, -1 = insertion
    int c = __popc(m0);
    if constexpr (BUDGET >= 1) {
      const uint32_t rp1 = mism<NB, T0 + BASE + 1>(p, wlo, whi), rn1 = mism<NB, T0 + BASE - 1>(p, wlo, whi);
      const uint32_t mp1 = m0 & rp1, mn1 = m0 & rn1;
      const int up = __popc(mp1), dn = __popc(mn1);
      c = min(c, KIND == REG_R ? min(1 + up, max(1, dn)) : min(max(1, up), 1 + dn));
      if constexpr (BUDGET >= 2) {
        const uint32_t rp2 = mism<NB, T0 + BASE + 2>(p, wlo, whi), rn2 = mism<NB, T0 + BASE - 2>(p, wlo, whi);
        if constexpr (KIND == REG_R) {                               // two insertions: every base matches at 0, +1 or +2; two deletions: in order, from B's side 0, -1, -2
          const int ins2 = 2 + __popc(mp1 & rp2);
          c = min(c, min(ins2, two_deletions<NB>(m0, rn1, rn2) ? 2 : 9));
        } else {                                                     // in front of A (its high end): deletions raise the shift towards the low end
          const int ins2 = 2 + __popc(mn1 & rn2);
          c = min(c, min(ins2, two_deletions<NB>(rp2, rp1, m0) ? 2 : 9));
        }
      }
    }
    return c;
  }
}
// lower bound of the edits a match of test (A, B, DSP) needs on the two fields outside the key, o = their 20 pattern bits
// (lower field in bits 0..9); the forced indels are part of region M's cost: a candidate needs edit_cost <= 2.
// LOOSE (the scan kernel's form; the resolve kernel runs the full one): where the other fields are two regions and nothing
// is forced (DSP = 0), a shift set that costs 2 in one region can only pay when the other region costs 0 -- its five bases
// match in place, one key hit in a thousand.  So the scan tries the sets that cost <= 1 in each region (three shifts instead
// of five, no order check) and lets everything through whose one region is exact: 0.2 % more suspects from these three tests,
// which were 40 % of the compare's instructions.
template <int A, int B, int DSP, bool LOOSE = false>
__device__ __forceinline__ int edit_cost(uint32_t o, uint32_t wlo, uint32_t whi) {
  constexpr int C = (A != 0 && B != 0) ? 0 : ((A != 1 && B != 1) ? 1 : 2);
  constexpr int D = (A != 3 && B != 3) ? 3 : ((A != 2 && B != 2) ? 2 : 1);
  constexpr int AD = DSP < 0 ? -DSP : DSP, BUD = 2 - AD;
  constexpr int BUD2 = LOOSE && DSP == 0 ? 1 : BUD;                  // two regions, nothing forced: see above
  static_assert(B > A + 1 || DSP == 0, "adjacent key fields are not displaced");
  if constexpr (A == 0 && B == 1) return region_cost<REG_R, 10, 10, 0, 0, BUD>(o, wlo, whi);                       // fields 2, 3 behind B
  else if constexpr (A == 2 && B == 3) return region_cost<REG_L, 10, 0, 0, 0, BUD>(o, wlo, whi);                   // fields 0, 1 in front of A
  else if constexpr (A == 0 && B == 3) return region_cost<REG_M, 10, 5, 0, DSP, BUD>(o, wlo, whi);                 // fields 1, 2 between
  else {
    int c1, c2;
    if constexpr (A == 1 && B == 2) {                                                                               // field 0 in front, field 3 behind
      c1 = region_cost<REG_L, 5, 0, 0, 0, BUD2>(o, wlo, whi);
      c2 = region_cost<REG_R, 5, 15, 0, 0, BUD2>(o >> 10, wlo, whi);
    }
    else if constexpr (A == 0 && B == 2) {                                                                          // field 1 between, field 3 behind
      c1 = region_cost<REG_M, 5, 5, 0, DSP, BUD2>(o, wlo, whi);
      c2 = region_cost<REG_R, 5, 15, 0, 0, BUD2>(o >> 10, wlo, whi);
    }
    else {                                                                                                          // (1, 3): field 0 in front of A (shift -DSP), field 2 between
      static_assert(A == 1 && B == 3 && C == 0 && D == 2, "fields");
      c1 = region_cost<REG_L, 5, 0, -DSP, 0, BUD2>(o, wlo, whi);
      c2 = region_cost<REG_M, 5, 10, 0, DSP, BUD2>(o >> 10, wlo, whi);
    }
    if constexpr (LOOSE && DSP == 0) return min(c1, c2) == 0 ? 0 : c1 + c2;
    else return c1 + c2;
  }
}
// the 20-bit key of test (A, B, DSP) from the window's 48 bits (field B in place, field A displaced by -DSP bases)
template <int A, int B, int DSP>
__device__ __forceinline__ uint32_t edit_key(uint32_t wlo, uint32_t whi) {
  return (wbits<4 + 10 * A - 2 * DSP>(wlo, whi) & 0x3ffu) | ((wbits<4 + 10 * B>(wlo, whi) & 0x3ffu) << 10);
}
// the 14 tests in table order; f(integral constants A, B, DSP) is called for variant v
template <typename F>
__device__ __forceinline__ void edit_variant(int v, F &&f) {
  using std::integral_constant;
  switch (v) {
    case 0: f(integral_constant<int, 0>(), integral_constant<int, 1>(), integral_constant<int, 0>()); break;
    case 1: f(integral_constant<int, 0>(), integral_constant<int, 2>(), integral_constant<int, -1>()); break;
    case 2: f(integral_constant<int, 0>(), integral_constant<int, 2>(), integral_constant<int, 0>()); break;
    case 3: f(integral_constant<int, 0>(), integral_constant<int, 2>(), integral_constant<int, 1>()); break;
    case 4: f(integral_constant<int, 0>(), integral_constant<int, 3>(), integral_constant<int, -2>()); break;
    case 5: f(integral_constant<int, 0>(), integral_constant<int, 3>(), integral_constant<int, -1>()); break;
    case 6: f(integral_constant<int, 0>(), integral_constant<int, 3>(), integral_constant<int, 0>()); break;
    case 7: f(integral_constant<int, 0>(), integral_constant<int, 3>(), integral_constant<int, 1>()); break;
    case 8: f(integral_constant<int, 0>(), integral_constant<int, 3>(), integral_constant<int, 2>()); break;
    case 9: f(integral_constant<int, 1>(), integral_constant<int, 2>(), integral_constant<int, 0>()); break;
    case 10: f(integral_constant<int, 1>(), integral_constant<int, 3>(), integral_constant<int, -1>()); break;
    case 11: f(integral_constant<int, 1>(), integral_constant<int, 3>(), integral_constant<int, 0>()); break;
    case 12: f(integral_constant<int, 1>(), integral_constant<int, 3>(), integral_constant<int, 1>()); break;
    case 13: f(integral_constant<int, 2>(), integral_constant<int, 3>(), integral_constant<int, 0>()); break;
    default: break;
  }
}
__device__ __host__ __forceinline__ int edit_variant_table(int var) { return var == 0 ? 0 : (var <= 3 ? 1 : (var <= 8 ? 2 : (var == 9 ? 3 : (var <= 12 ? 4 : 5)))); }

// Exact part of the verify (second kernel): (window ending at p, pattern pi) agree on this combo's key
// and are within k substitutions on the rest of the packed window; count mismatches on the raw stream
// codes over the whole pattern (N = mismatch, EOS = reject) and report -- once: only through the first
// combo of the plan whose two fields are clean.
// Returns whether (p, pi) is a candidate this combo reports; *hh is then its record.
__device__ __forceinline__ bool pair_verify(const PairArgs &a, int combo, int64_t p, uint32_t pi, pm_hit *hh) {
  const int L = a.pat_len[pi];
  const int64_t start = p + 1 - L;
  if (start < 0) return false;
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
  if (a.eos_code >= 0 && (eos & lenmask)) return false;   // EOS inside the window: never a candidate
  int ham = __popc(mism);                             // N (or any other code) = mismatch
  if (ham > a.k) return false;
  // exact-base constraints (pattern_alignment.cc:320-323: a substitution inside an exact zone is a
  // constraint violation, the verify fails)
  if (mism & a.pat_zone[pi]) {
    if (a.viol_level <= 0) return false;
    ham = a.viol_level;
  }
  const uint32_t tail = mism >> (L - 20);             // the 20 bases the plan looks at
  uint32_t dirty = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) if ((tail >> (5 * j)) & 31u) dirty |= 1u << j;
  int first = -1;
  for (int c = 0; c < a.ncombos && first < 0; ++c)
    if (!((dirty >> a.fa[c]) & 1u) && !((dirty >> a.fb[c]) & 1u)) first = c;
  if (first != combo) return false;
  const int half = L / 2;
  const bool left_clean = (mism & ((1u << half) - 1u)) == 0, right_clean = (mism >> half) == 0;
  hh->end = p + 1; hh->pid = a.pat_id[pi]; hh->k = (uint8_t)ham;
  hh->aux[0] = (uint8_t)((left_clean ? 1 : 0) | (right_clean ? 2 : 0)); hh->aux[1] = hh->aux[2] = 0;
  return true;
}

// Output of the verify kernel.  The record list's end is ONE counter for the whole grid and same-address atomics
// serialise (~10 ns each under load), so a workgroup collects its records in LDS and appends them in batches: one atomic
// per ~1500 records instead of one per wave and call (hit-dense streams -- tandem repeats, 0.4 candidates per base --
// spent most of this kernel waiting for that counter).  Called by whichever lanes of a wave are executing together.
constexpr int VSTAGE = 2048;                                        // records a workgroup stages (32 KiB)
struct VerifyStage { pm_hit *rec; uint32_t *fill, *valid; };        // LDS: records, reserved slots, first slot that was refused

__device__ __forceinline__ void pair_emit(const PairArgs &a, const VerifyStage &vs, bool ok, const pm_hit &hh) {
  const unsigned long long bal = __ballot(ok);
  if (bal == 0) return;
  const int leader = __ffsll((long long)bal) - 1;
  const int lane = threadIdx.x & 63;
  const uint32_t cnt = (uint32_t)__popcll(bal), mine = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
  uint32_t pos = 0;
  if (lane == leader) pos = atomicAdd(vs.fill, cnt);
  pos = __builtin_amdgcn_readlane(pos, leader);
  if (pos + cnt <= (uint32_t)VSTAGE) {
    if (ok) vs.rec[pos + mine] = hh;
    return;
  }
  // no room (a workgroup whose suspects give thousands of records in one trip): this batch goes straight to the list;
  // every later reservation of the trip is refused as well (fill stays above VSTAGE), the flush takes the slots in front
  if (lane == leader) atomicMin(vs.valid, pos);
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(a.counter, (unsigned long long)cnt);
  const uint32_t blo = __builtin_amdgcn_readlane((uint32_t)base, leader), bhi = __builtin_amdgcn_readlane((uint32_t)(base >> 32), leader);
  if (ok) {
    const unsigned long long o = (((unsigned long long)bhi << 32) | blo) + (unsigned long long)mine;
    if (o < a.cap) a.out[o] = hh;
  }
}

// A suspect: a window whose slot says "within k on the other fields" for the key's first / second /
// third pattern (what & 1, 2, 4), that the key has more than three (WHAT_REST: walk the rest of the
// key's run of the sorted pattern list), or whose key has no slot (WHAT_ALL: walk the whole run).
__device__ __forceinline__ void pair_resolve(const PairArgs &a, const VerifyStage &vs, int combo, uint32_t rank, uint32_t what, uint32_t wo, int64_t p, pm_hit *first, bool *have) {
  const uint32_t *fp = a.first_pat + a.first_off[combo];
  const uint32_t *ord = a.order + (size_t)combo * a.np;
  const uint32_t *ol = a.olist + (size_t)combo * a.np;
  const uint32_t t0 = fp[rank], t1 = fp[rank + 1];
  pm_hit hh;
  // the suspect's first record stays in registers (the kernel writes those block by block), further ones -- a
  // window within k of two patterns that share the key -- go out one by one
  auto take = [&](bool ok) __attribute__((always_inline)) {
    const bool extra = ok && *have;
    if (ok && !*have) { *first = hh; *have = true; }
    pair_emit(a, vs, extra, hh);
  };
  for (uint32_t j = 0; j < 3; ++j)
    take(((what >> j) & 1u) && t0 + j < t1 && pair_verify(a, combo, p, ord[t0 + j], &hh));   // (free slot fields repeat the first pattern)
  if (what & (WHAT_REST | WHAT_ALL)) {
    // the rest of the key's run: its patterns' other fields lie next to each other (one or two cache lines for a run of
    // twenty; ord[] -> pat40[] was two dependent random loads per pattern -- keys shared by many primers are what a
    // skewed stream with primers cut from it is made of)
    for (uint32_t t = t0 + ((what & WHAT_ALL) ? 0u : 3u); t < t1; ++t)
      take(sym_distance(ol[t] ^ wo) <= a.k && pair_verify(a, combo, p, ord[t], &hh));
  }
}

// Second kernel: the scan kernel's suspects (a few per thousand positions), one per thread; the count is read
// from device memory.  Here, with every lane busy, the dependent loads of the verify (bitmap row, slot, pattern
// list, pattern, raw stream bytes) cost little; inside the scan kernel they ran with one or two live lanes per
// wave and held the wave for microseconds.  The records leave block by block: one atomic on the shared counter per
// 256 suspects -- one per wave and emit call, 10^5 to 10^6 per launch on one address at ~2 ns each, was 1.7 of
// this kernel's 2.0 ms.
__global__ __launch_bounds__(256) void pm_pair_verify(PairArgs a) {
  __shared__ pm_hit s_rec[VSTAGE];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_fill, s_valid, s_full;
  const VerifyStage vs = {s_rec, &s_fill, &s_valid};
  unsigned long long n = *a.susp_count;
  if (n > a.susp_cap) n = a.susp_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  if (threadIdx.x == 0) { s_fill = 0; s_valid = (uint32_t)VSTAGE; }
  __syncthreads();
  // the staged records join the list: one atomic for all of them (block-uniform call)
  auto flush = [&]() __attribute__((always_inline)) {
    const uint32_t cnt = min(s_fill, s_valid);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = cnt ? atomicAdd(a.counter, (unsigned long long)cnt) : 0ull; s_fill = 0; s_valid = (uint32_t)VSTAGE; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) if (s_base + i < a.cap) a.out[s_base + i] = s_rec[i];
    __syncthreads();
  };
  for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < n; base += stride) {   // block-uniform trip count
    const unsigned long long i = base + threadIdx.x;
    pm_hit first;
    bool have = false;
    if (i < n) {
      const uint4 r = a.susp[i];
      const uint64_t pw = ((uint64_t)r.w << 32) | r.z;
      if (r.x != 0xffffffffu) {                                       // (a slot its wave reserved and did not need)
      const int combo = (int)((pw >> 40) & 7u);
      // the key's row of the bitmap (global copy) -> rank; its slot -> which of its patterns are within k on the other fields
      const uint32_t key = r.x, wo = r.y, row = key & 0x7fffu, bit = key >> 15;
      const uint32_t wd = a.image[(size_t)combo * PAIR_BITMAP_WORDS + row];
      if ((wd >> bit) & 1u) {
        const uint32_t lr = (uint32_t)__popc(wd & ((1u << bit) - 1u));
        const uint32_t rank = a.row_base[(size_t)combo * (PAIR_BITMAP_WORDS + 1) + row] + lr;
        uint32_t what = WHAT_ALL;
        if (lr < (uint32_t)a.stride - 1u) {
          const uint2 sl = a.slots[((size_t)combo * PAIR_BITMAP_WORDS + row) * (size_t)a.stride + lr];
          const uint64_t S = ((uint64_t)sl.y << 32) | sl.x;
          what = (sym_distance(((uint32_t)S ^ wo) & F20) <= a.k ? 1u : 0u) | (sym_distance(((uint32_t)(S >> 20) ^ wo) & F20) <= a.k ? 2u : 0u) |
                 (sym_distance(((uint32_t)(S >> 40) ^ wo) & F20) <= a.k ? 4u : 0u) | ((sl.y >> 31) ? WHAT_REST : 0u);
        }
        pair_resolve(a, vs, combo, rank, what, wo, (int64_t)(pw & 0xffffffffffull), &first, &have);
      }
      }
    }
    if (a.debug & 16) have = false;
    pair_emit(a, vs, have, first);                                  // (every lane is here)
    __syncthreads();
    if (threadIdx.x == 0) s_full = s_fill > (uint32_t)(VSTAGE - 512);   // one thread decides: a wave that runs ahead into the next trip moves s_fill
    __syncthreads();
    if (s_full) flush();
  }
  flush();
}

// One (field pair, chunk) of the scan.  A, B: key fields (compile time: every window offset of pass A is an
// immediate).  Per block of 1024 positions (16 per lane) a wave runs pass A -- the 16 membership tests of every
// lane, straight-line -- and then pass B, rounds over the lanes' pending key hits (a three-stage pipeline over a
// ring of register entries: pick the window and re-read its bitmap row; rank in the row, slot address, 8-byte
// load; compare the slot's three patterns).  What the compare cannot dismiss (1e-3 of the windows) is queued in LDS
// and leaves in batches for pm_pair_verify.  See the file comment and the comments at the stages.
template <typename F>
__device__ __forceinline__ void static_for5(F &&f) { for_windows(std::make_integer_sequence<int, 5>(), f); }

// DSP, FLOOR: measurement instantiations (pm_pair_edit_scan below): the pair geometry as the first stage of an
// EDIT-distance plan tests field A displaced by DSP bases against field B (two clean fields of a match with <= 2 edits sit
// 5 (B - A) + d bases apart, |d| <= 2; 14 (pair, d) tests cover every placement, scripts/edit_pair_cover.py).  FLOOR = 1
// keeps the substitution compare of the slot's three patterns (a lower bound of such a kernel's cost), FLOOR = 2 runs, for
// two patterns of the slot, the cheapest necessary condition for "<= 2 edits on the other ten bases" that was found: mismatch
// masks at five shifts, tiered AND / popcount tests.  Neither produces a hit list: they exist to be timed.  With DSP = 0 and
// FLOOR = 0 (the product's instantiations) every `if constexpr` below takes the branch that was there before.
template <int A, int B, int DSP = 0, int FLOOR = 0>
__device__ __forceinline__ void pair_scan_body(const PairArgs &a, const int combo, const int cj, const int variant = 0) {
  constexpr int C = (A != 0 && B != 0) ? 0 : ((A != 1 && B != 1) ? 1 : 2);
  constexpr int D = (A != 3 && B != 3) ? 3 : ((A != 2 && B != 2) ? 2 : 1);
  static_assert(A < B && C < D && C != A && C != B && D != A && D != B, "fields");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t sub = a.chunk_len / WAVES;
  const int64_t ws = (a.chunk0 + cj) * a.chunk_len + (int64_t)wave * sub;   // first position (window's last base) of this wave
  int64_t own_lo = ws > a.begin ? ws : a.begin;
  int64_t own_hi = ws + sub;
  if (own_hi > a.end) own_hi = a.end;
  if (own_hi > a.n) own_hi = a.n;
  if (own_lo < 19) own_lo = 19;                                   // the 20-base window must fit in the stream
  if (own_lo >= own_hi) return;

  const uint32_t stride = (uint32_t)a.stride, kmax = stride - 1u;   // slots per row; rank of a row's overflow marker
  const char *slots = reinterpret_cast<const char *>(a.slots + (size_t)combo * PAIR_BITMAP_WORDS * stride);
  // the 32 bases in front of the wave's range (wave-uniform; they enter lanes 0 and 1 through the DPP shifts)
  uint32_t carry1, carry2;
  {
    const uint32_t pk = load_packed(a.packed, a.npacked, ws - 32 + 16 * (lane & 1));
    carry2 = __builtin_amdgcn_readlane(pk, 0);
    carry1 = __builtin_amdgcn_readlane(pk, 1);
  }
  const uint32_t QB = PAIR_BITMAP_WORDS * 4 + (uint32_t)wave * (PAIR_QUEUE * 16);   // this wave's suspect queue (16-byte entries) behind the bitmap
  int qn = 0;                                                     // wave-uniform: entries in it
  uint32_t keep = 0;                                              // measurement (debug & 1): verdicts folded, never emitted

  // four blocks of 1024 bases (one packed dword per lane each) in flight per wave
  // (the edit plan's windows look two bases beyond themselves: its waves also fetch the block behind their range)
  const int64_t pre_hi = FLOOR == 3 ? own_hi + 1024 : own_hi;
  uint32_t q0 = load_packed(a.packed, a.npacked, ws + 16 * lane);
  uint32_t q1 = ws + 1024 < pre_hi ? load_packed(a.packed, a.npacked, ws + 1024 + 16 * lane) : 0u;
  uint32_t q2 = ws + 2048 < pre_hi ? load_packed(a.packed, a.npacked, ws + 2048 + 16 * lane) : 0u;
  uint32_t q3 = ws + 3072 < pre_hi ? load_packed(a.packed, a.npacked, ws + 3072 + 16 * lane) : 0u;
  const uint32_t *pk_wave = a.packed + (ws >> 4);                 // wave-uniform: the wave's first dword (ws is a multiple of 1024)
  uint32_t pf_idx = 4096u / 16u + (uint32_t)lane;                  // dword of this lane in the block four ahead

  const int negk1 = -(a.k + 1);

  // ---- suspects ----------------------------------------------------------------------------------------
  // Windows the consume stage cannot dismiss become suspect records {key, other fields, position |
  // field pair << 40} for pm_pair_verify, which works out rank and slot again with every lane busy.
  // They are collected in a queue of this wave in LDS and leave in batches of >= 64, one atomic each
  // (same-address atomics serialise at ~2.5 ns; a dependent table read on this path held the whole
  // wave for a microsecond per suspect: 7 ms per 3 Gbp).
  // The list's end is ONE counter for the whole grid, and same-address atomics serialise (~10 ns each under load): on a
  // stream where a third of the windows are suspects (skewed composition, primers cut from the stream: 7e8 suspects per
  // 300 Mbp) one atomic per batch of 64 was 1e7 atomics = the whole kernel (110 ms against 1.9 on uniform text).  So a
  // wave that comes back for more reserves ahead: its first batch takes exactly what it holds (a wave of a uniform
  // stream has 2.6 suspects in all: no unused slots), every later reservation doubles, up to 4096 slots; slots left
  // over when the wave ends are marked (key all ones) and skipped by the verify kernel.
  unsigned long long blk_at = 0;                                  // wave-uniform: next free slot of the wave's reserved run
  uint32_t blk_left = 0, blk_next = 0;                            // slots left in it; size of the next reservation
  auto flush = [&]() __attribute__((always_inline)) {
    int done = 0;
    while (done < qn) {
      if (blk_left == 0) {
        const uint32_t want = max((uint32_t)(qn - done), blk_next);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.susp_count, (unsigned long long)want);
        blk_at = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base);
        blk_left = want;
        blk_next = min(2u * max(want, 64u), 4096u);
      }
      const int take = min(qn - done, (int)blk_left);
      for (int e = lane; e < take; e += 64) {
        const u32x4 q = *lds128(QB + 16 * (uint32_t)(done + e));
        const uint64_t pw = (uint64_t)(ws + q.z) | ((uint64_t)(FLOOR == 3 ? variant : combo) << 40);
        if (blk_at + (unsigned long long)e < a.susp_cap) a.susp[blk_at + (unsigned long long)e] = make_uint4(q.x, q.y, (uint32_t)pw, (uint32_t)(pw >> 32));
      }
      blk_at += (unsigned long long)take; blk_left -= (uint32_t)take; done += take;
    }
    qn = 0;
  };
  auto give_back = [&]() __attribute__((always_inline)) {          // the wave is done: mark what it reserved and did not use
    for (uint32_t e = (uint32_t)lane; e < blk_left; e += 64)
      if (blk_at + e < a.susp_cap) a.susp[blk_at + e] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    blk_left = 0;
  };
  auto enqueue = [&](bool mine, uint32_t key, uint32_t wo, uint32_t prel) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(mine);
    if (bal == 0) return;
    if (a.debug & 1) { keep ^= key; return; }
    if (mine) {
      const int s_ = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      u32x4 q;
      q.x = key; q.y = wo; q.z = prel; q.w = 0;
      *lds128(QB + 16 * (uint32_t)s_) = q;
    }
    qn += __popcll(bal);
    if (qn >= 64) flush();                                         // (a call adds at most 64: the queue holds 128)
  };

  // ---- pass A: every window's key against the bitmap -----------------------------------------------------
  // window I starts at bit 2 I + 26 of prev2 : prev1 : cur; field f at + 10 f.  The key comes out at bits 2..21.
  auto key_of = [&](auto WIN, uint32_t prev2, uint32_t prev1, uint32_t cur) __attribute__((always_inline)) -> uint32_t {
    constexpr int I = decltype(WIN)::value;
    const uint32_t X = bits_at<2 * I + 26 + 10 * A - 2 - 2 * DSP>(prev2, prev1, cur);
    if constexpr (B == A + 1 && DSP == 0) return X;
    else { const uint32_t Y = bits_at<2 * I + 26 + 10 * B - 12>(prev2, prev1, cur); return (X & 0xffcu) | (Y & ~0xffcu); }
  };

  // ---- pass B: the windows whose key occurs, one per lane and round ---------------------------------------
  // 17 % of the windows are key hits; what is done for them -- rank in the row, slot address, the 8-byte load,
  // three XOR + popcount against the slot's patterns -- costs ~60 vector-instruction slots, four times the
  // membership test, and VALU issue is what this kernel is bound by: most of the integer instructions it is
  // made of (every three-operand form, v_bcnt, v_bfe, v_alignbit, anything with an SGPR operand) issue at HALF the
  // rate of v_and / v_xor / v_add / shifts by a constant on gfx950 (scripts/probe/valu_rate.hip).  Done branch-free
  // for all 16 windows of a lane the kernel took 17.7 ms per 3 Gbp.  So the hits are processed in ROUNDS instead:
  // per round every lane takes its oldest pending hit.  Pending hits are a 32-bit mask per lane -- bits 0..15 the
  // previous block's windows, 16..31 this block's -- so a lane with many hits in one block works them off while
  // its neighbours idle only for what exceeds two blocks' worth: a block takes RMIN rounds, more only while some
  // lane still has hits of the previous block (whose stream words are about to be dropped).
  // A round is a three-stage software pipeline over the lane's hits (registers: see NR below):
  //   stage 1  next pending window -> its 40 bits (two v_alignbit by 2 i on the block's pre-shifted words) -> key,
  //            other fields, position; read the key's row of the bitmap again (LDS);
  //   stage 2  (one round later) rank inside the row -> slot address -> global_load_dwordx2;
  //   stage 3  (five rounds later) the slot's three patterns against the window's other 20 bits; the few that
  //            stay suspicious (or need the key's pattern list walked) are queued for the verify kernel.
  // Lanes without a pending hit run along with key 0 (row 0: one cached line) and their verdict masked.
  // The pipeline's registers are a ring of NR entries with compile-time indices (round<PH> is instantiated for
  // PH = 0 .. NR-1 and rounds run in pairs, so nothing is ever moved): round t works on entry t mod NR three times --
  // stage 3 consumes it (filled NR rounds ago, its slot loaded NR - 1 rounds ago), stage 1 refills it; stage 2 loads the
  // slot of the entry filled last round.  NR - 1 = 5 slot loads in flight per lane.
#ifndef PM_PAIR_NR
#define PM_PAIR_NR 6
#endif
  constexpr int NR = PM_PAIR_NR;
  static_assert(NR % 2 == 0 && NR >= 4 && NR <= 12, "ring size");
  uint32_t P = 0;                                                   // pending key hits: bits 0..15 previous block, 16..31 current block
  uint32_t pu1 = 0, pu2 = 0, pu3 = 0, cu1 = 0, cu2 = 0, cu3 = 0;       // stream bits 26..57 / 58..89 / 90..95 of the previous / current block's word triple
  uint32_t tagbase = 0xfffffff0u;                                   // 16 (block - 1): a pending window's tag = tagbase + i = 16 x its block + its index in the block
  u32x2 e_sl[NR];                                                   // slot
  uint32_t e_key[NR], e_wo[NR], e_rel[NR], e_okm[NR], e_wd[NR];       // key, other fields, position, validity (sign bit), the key's row of the bitmap
  uint32_t e_hi[FLOOR >= 2 ? NR : 1];                                // (FLOOR 2, 3: e_wo / e_hi = the window's 48 bits from two bases in front of it)
  for (int d = 0; d < (FLOOR >= 2 ? NR : 1); ++d) e_hi[d] = 0;
#pragma unroll
  for (int d = 0; d < NR; ++d) { e_sl[d].x = 0; e_sl[d].y = 0; e_key[d] = e_wo[d] = e_rel[d] = e_okm[d] = e_wd[d] = 0; }
  uint32_t m5v, f20v, rowmask;                                      // constants in VGPRs: an SGPR or literal operand halves the issue rate of v_and
  asm volatile("v_mov_b32 %0, 0x55555\n v_mov_b32 %1, 0xfffff\n v_mov_b32 %2, 0x7fff" : "=v"(m5v), "=v"(f20v), "=v"(rowmask));

  auto round = [&](auto PHASE) __attribute__((always_inline)) {
    constexpr int PH = decltype(PHASE)::value, LD = (PH + NR - 1) % NR;
    // stage 3: entry PH, whose slot was loaded NR - 1 rounds ago: substitutions against each of the slot's three
    // patterns minus k + 1, a negative one = within k; the slot's "walk" flag is the sign bit of its high dword;
    // any sign bit = suspicious (if the entry holds a window at all)
    if constexpr (FLOOR == 3) {
      // the slot's two patterns (the table of the edit plan holds two per key; more: walk flag): can the other ten bases be
      // brought to the text with the edits that are left?  cost - 3 < 0 = yes as far as edit_cost can tell
      const uint32_t lo = e_wo[PH], hi = e_hi[PH];
      const int d1 = edit_cost<A, B, DSP, true>(e_sl[PH].x & f20v, lo, hi) - 3;
      const int d2 = edit_cost<A, B, DSP, true>(__builtin_amdgcn_alignbit(e_sl[PH].y, e_sl[PH].x, 20) & f20v, lo, hi) - 3;
      const bool susp = (int)(((uint32_t)(d1 | d2) | e_sl[PH].y) & e_okm[PH]) < 0;
      enqueue(susp, lo, hi, ((e_rel[PH] >> 4) << 10) + 16u * (uint32_t)lane + (e_rel[PH] & 15u));
    } else if constexpr (FLOOR == 2) {
      // the other ten pattern bases (contiguous from field C in this cost model) against the text at shifts -2 .. 2
      constexpr int OC = 4 + 10 * C;
      const uint32_t lo = e_wo[PH], hi = e_hi[PH];
      uint32_t T[5];
      static_for5([&](auto SI) __attribute__((always_inline)) { constexpr int si = decltype(SI)::value; T[si] = __builtin_amdgcn_alignbit(hi, lo, OC + 2 * (si - 2)) & f20v; });
      auto score = [&](uint32_t o) __attribute__((always_inline)) -> int {
        uint32_t m[5];
        static_for5([&](auto SI) __attribute__((always_inline)) { constexpr int si = decltype(SI)::value; const uint32_t x = o ^ T[si]; m[si] = (x | (x >> 1)) & m5v; });
        const int c0 = __popc(m[2]), c1 = __popc(m[2] & m[3]) + 1, c1n = __popc(m[2] & m[1]) + 1;
        const int c2 = __popc(m[2] & m[3] & m[4]) + 2, c2n = __popc(m[2] & m[1] & m[0]) + 2;
        return min(min(c0, min(c1, c1n)), min(c2, c2n)) - 3;          // negative = within two edits as far as this test can tell
      };
      const int d1 = score(e_sl[PH].x & f20v), d2 = score(__builtin_amdgcn_alignbit(e_sl[PH].y, e_sl[PH].x, 20) & f20v);
      const bool susp = (int)(((uint32_t)(d1 | d2) | e_sl[PH].y) & e_okm[PH]) < 0;
      enqueue(susp, e_key[PH], lo, ((e_rel[PH] >> 4) << 10) + 16u * (uint32_t)lane + (e_rel[PH] & 15u));
    } else {
      const uint32_t w = e_wo[PH];
      const uint32_t x1 = e_sl[PH].x ^ w, x2 = __builtin_amdgcn_alignbit(e_sl[PH].y, e_sl[PH].x, 20) ^ w, x3 = (e_sl[PH].y >> 8) ^ w;
      const int d1 = __popc((x1 | (x1 >> 1)) & m5v) + negk1;
      const int d2 = __popc((x2 | (x2 >> 1)) & m5v) + negk1;
      const int d3 = __popc((x3 | (x3 >> 1)) & m5v) + negk1;
      const bool susp = (int)(((uint32_t)(d1 | d2 | d3) | e_sl[PH].y) & e_okm[PH]) < 0;
      // (position relative to ws = 1024 x block + 16 x lane + window index, from the tag)
      enqueue(susp, e_key[PH], w, ((e_rel[PH] >> 4) << 10) + 16u * (uint32_t)lane + (e_rel[PH] & 15u));
    }
    // stage 2: entry LD, filled last round: rank of its key inside the row -> slot (keys beyond the row's slots: its
    // last slot, the overflow marker)
    {
      const uint32_t bit = e_key[LD] >> 15;
      const uint32_t lr = min((uint32_t)__popc(__builtin_amdgcn_ubfe(e_wd[LD], 0, bit)), kmax);
      const uint32_t off = (__umul24(e_key[LD] & rowmask, stride) + lr) << 3;
      e_sl[LD] = *reinterpret_cast<const u32x2 *>(slots + off);       // plain load: nontemporal ran 3x, sc1 1.7x slower
    }
    // stage 1: the lane's oldest pending hit into entry PH
    {
      uint32_t i;                                                     // lowest pending window, -1 when nothing is pending
      asm("v_ffbl_b32 %0, %1" : "=v"(i) : "v"(P));
      P &= P - 1u;
      const uint32_t okm = ~i;                                        // sign bit: the entry holds a window
      const bool cb = (i & 16u) != 0;                                 // window of the current block?
      const uint32_t j2 = i + i;                                      // (v_alignbit takes the low five bits: 2 (i & 15); an add issues at twice a left shift's rate)
      const uint32_t u1 = cb ? cu1 : pu1, u2 = cb ? cu2 : pu2, u3 = cb ? cu3 : pu3;
      const uint32_t wlo = __builtin_amdgcn_alignbit(u2, u1, j2), whi = __builtin_amdgcn_alignbit(u3, u2, j2);   // window bits 0..31, 32..39 (+ junk above)
      // 20 bits from window bit O (O = 0, 10, 20; junk above bit 19 unless masked)
      auto from = [&](int O) __attribute__((always_inline)) -> uint32_t { return O == 0 ? wlo : (O == 10 ? wlo >> 10 : __builtin_amdgcn_alignbit(whi, wlo, 20)); };
      uint32_t key, wo;
      if constexpr (FLOOR != 0) {
        // (measurement builds: the block's words start two bases in front of the window, see cu1 below; field A displaced)
        constexpr int OA = 4 + 10 * A - 2 * DSP, OB = 4 + 10 * B - 10, OCC = 4 + 10 * C, ODD = 4 + 10 * D - 10;
        key = (__builtin_amdgcn_alignbit(whi, wlo, OA) & 0x3ffu) | (__builtin_amdgcn_alignbit(whi, wlo, OB) & 0xffc00u);
        if constexpr (FLOOR >= 2) { wo = wlo; e_hi[PH] = whi; }
        else wo = (__builtin_amdgcn_alignbit(whi, wlo, OCC) & 0x3ffu) | (__builtin_amdgcn_alignbit(whi, wlo, ODD) & 0xffc00u);
      } else {
      if constexpr (B == A + 1) key = from(10 * A) & f20v; else key = (from(10 * A) & 0x3ffu) | (from(10 * B - 10) & 0xffc00u);
      if constexpr (D == C + 1) wo = from(10 * C) & f20v; else wo = (from(10 * C) & 0x3ffu) | (from(10 * D - 10) & 0xffc00u);
      }
      key &= (uint32_t)((int)okm >> 31);                              // no window: key 0 (row 0: one cached line)
      e_key[PH] = key; e_wo[PH] = wo; e_okm[PH] = okm;
      e_rel[PH] = tagbase + i;
      e_wd[PH] = *lds32((key & rowmask) << 2);
    }
  };
  int ph = 0;                                                       // wave-uniform: next pair of ring entries
  int quiet = NR + 1;                                               // wave-uniform: rounds run since a lane last had a hit pending (> NR: the pipeline is empty)
  auto two_rounds = [&]() __attribute__((always_inline)) {
    bool done = false;
    for_windows(std::make_integer_sequence<int, NR / 2>(), [&](auto Q) __attribute__((always_inline)) {
      constexpr int q = decltype(Q)::value;
      if (!done && ph == q) {
        round(std::integral_constant<int, 2 * q>()); round(std::integral_constant<int, 2 * q + 1>());
        ph = q + 1 == NR / 2 ? 0 : q + 1;
        done = true;
      }
    });
  };

  const bool stats = (a.debug & 32) != 0 && a.stats != nullptr;     // wave-uniform measurement switch (bench.py --stream-style)
  uint32_t st_blocks = 0, st_rounds = 0, st_hits = 0;
  int64_t bb = ws;
  const int rmin = (a.debug >> 8) & 15 ? (a.debug >> 8) & 15 : 2;   // rounds per block at least (measurement knob in the debug word)
  while (bb < own_hi) {
    const uint32_t cur = q0;
    q0 = q1; q1 = q2; q2 = q3;
    if (bb + 4096 < pre_hi) q3 = pk_wave[pf_idx];                    // (the packed stream is padded: no per-lane bounds check)
    pf_idx += 64u;
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
    // pass A: 16 membership tests (key, row of the bitmap, bit), the bits funnelled into acc from the top
    uint32_t acc = 0;
    {
      uint32_t ks[16], wd[16];
      for_windows(std::make_integer_sequence<int, 16>(), [&](auto WIN) __attribute__((always_inline)) {
        constexpr int I = decltype(WIN)::value;
        ks[I] = key_of(WIN, prev2, prev1, cur);
        wd[I] = *lds32(ks[I] & 0x1fffcu);
      });
      for_windows(std::make_integer_sequence<int, 16>(), [&](auto WIN) __attribute__((always_inline)) {
        constexpr int I = decltype(WIN)::value;
        acc = __builtin_amdgcn_alignbit(wd[I] >> ((ks[I] >> 17) & 31u), acc, 1);   // bits 15..19 of the key pick the bit; window 0 ends up at bit 16
      });
    }
    P |= acc & (own << 16);
    if (stats) { st_hits += (uint32_t)__popc(acc & (own << 16)); ++st_blocks; }
    constexpr int WB = FLOOR != 0 ? 22 : 26;                          // (measurement builds keep two more bases in front of the windows)
    cu1 = __builtin_amdgcn_alignbit(prev1, prev2, WB); cu2 = __builtin_amdgcn_alignbit(cur, prev1, WB); cu3 = cur >> WB;
    if constexpr (FLOOR == 3) {
      // a lane's last windows reach two bases into the next lane's dword (shifts + 1, + 2 behind the window): whole-wave
      // shift the other way, lane 63 takes the next block's first dword (prefetched: q0 already is the next block)
      const uint32_t nfirst = __builtin_amdgcn_readfirstlane(q0);
      const uint32_t nxt = __builtin_amdgcn_update_dpp(nfirst, cur, 0x130, 0xf, 0xf, false);       // wave_shl:1
      cu3 |= nxt << 10;
    }
    // pass B (skipped while no lane has a hit pending and the pipeline has run empty: sparse pattern sets)
    if (__ballot(P != 0)) quiet = 0;
    if (quiet <= NR) {
      int r = 0;
      do { two_rounds(); r += 2; } while (r < rmin || __ballot((P & 0xffffu) != 0));
      quiet += r;
      st_rounds += (uint32_t)r;
    }
    P >>= 16; pu1 = cu1; pu2 = cu2; pu3 = cu3; tagbase += 16u;
    bb += 1024;
  }
  while (__ballot(P != 0)) two_rounds();                            // what the last block left
  for (int d = 0; d < NR / 2 + 1; ++d) two_rounds();                // and what is still in the pipeline
  if (keep == 0x9e3779b9u) a.susp[0] = make_uint4(keep, 0, 0, 0);   // (measurement switch: the verdicts must stay alive)
  flush();
  give_back();
  if (stats) {
    for (int o = 32; o > 0; o >>= 1) st_hits += (uint32_t)__shfl_xor((int)st_hits, o);
    if (lane == 0) { atomicAdd(a.stats, (unsigned long long)st_blocks); atomicAdd(a.stats + 1, (unsigned long long)st_rounds); atomicAdd(a.stats + 2, (unsigned long long)st_hits); }
  }
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
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.image + (size_t)combo * PAIR_BITMAP_WORDS);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < PAIR_BITMAP_WORDS / 4; i += PAIR_THREADS) dst[i] = src[i];
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

// The pair geometry as the first stage of the edit-distance plan: 14 (field pair, displacement) tests per window, one
// workgroup per (test, chunk), the tests of one field pair next to each other in the grid (they share that pair's slot
// table in L2).  FLOOR = 3 is the product (-k 2: edit_cost, suspects for pm_pair_edit_resolve); FLOOR = 1, 2 are the
// measurement builds the plan was decided on (profiles/r04_edit_pair_floor.json): they write suspect records and nothing
// else.  a.ncombos = 14.
template <int FLOOR>
__global__ __launch_bounds__(PAIR_THREADS) void pm_pair_edit_scan(PairArgs a) {
  extern __shared__ uint32_t lds[];
  const int per_super = a.group * a.ncombos;
  const int sc = blockIdx.x / per_super;
  const int rem = blockIdx.x - sc * per_super;
  int var = rem / a.group;
  int cj = sc * a.group + (rem - var * a.group);
  const int full = (a.nchunks / a.group) * a.group;
  if (sc * a.group >= full) {
    const int tail = a.nchunks - full;
    const int r2 = blockIdx.x - (full / a.group) * per_super;
    var = r2 / tail;
    cj = full + (r2 - var * tail);
  }
  if (cj >= a.nchunks || var >= a.ncombos) return;
  // variants in table order: (0,1,0); (0,2,-1..1); (0,3,-2..2); (1,2,0); (1,3,-1..1); (2,3,0)
  const int table = edit_variant_table(var);
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.image + (size_t)table * PAIR_BITMAP_WORDS);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < PAIR_BITMAP_WORDS / 4; i += PAIR_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  switch (var) {                                                    // wave-uniform
    case 0: pair_scan_body<0, 1, 0, FLOOR>(a, table, cj, 0); break;
    case 1: pair_scan_body<0, 2, -1, FLOOR>(a, table, cj, 1); break;
    case 2: pair_scan_body<0, 2, 0, FLOOR>(a, table, cj, 2); break;
    case 3: pair_scan_body<0, 2, 1, FLOOR>(a, table, cj, 3); break;
    case 4: pair_scan_body<0, 3, -2, FLOOR>(a, table, cj, 4); break;
    case 5: pair_scan_body<0, 3, -1, FLOOR>(a, table, cj, 5); break;
    case 6: pair_scan_body<0, 3, 0, FLOOR>(a, table, cj, 6); break;
    case 7: pair_scan_body<0, 3, 1, FLOOR>(a, table, cj, 7); break;
    case 8: pair_scan_body<0, 3, 2, FLOOR>(a, table, cj, 8); break;
    case 9: pair_scan_body<1, 2, 0, FLOOR>(a, table, cj, 9); break;
    case 10: pair_scan_body<1, 3, -1, FLOOR>(a, table, cj, 10); break;
    case 11: pair_scan_body<1, 3, 0, FLOOR>(a, table, cj, 11); break;
    case 12: pair_scan_body<1, 3, 1, FLOOR>(a, table, cj, 12); break;
    case 13: pair_scan_body<2, 3, 0, FLOOR>(a, table, cj, 13); break;
    default: break;
  }
}

// Second kernel of the edit plan: a suspect = {window bits, position | test << 40}.  Every pattern that has the key of the
// test goes through edit_cost again (here every lane is busy and a key's run of patterns is one contiguous list); what
// passes leaves as a seed record (pattern index, position) for the automaton (pm_edits_verify), whose last five steps read
// the ends position - 1 .. position + 3 -- exactly where a match whose frame has field B in place can end.  Records are
// staged in LDS and join the list in batches (one atomic per few thousand: the list's end is one counter for the grid).
constexpr int ESTAGE = 2048;                                        // 8-byte records a workgroup stages (16 KiB)
constexpr int EWORK = 3072;                                         // (suspect, pattern of its key's run) items a workgroup queues per trip
__global__ __launch_bounds__(256) void pm_pair_edit_resolve(PairArgs a) {
  __shared__ uint64_t s_rec[ESTAGE];
  __shared__ uint4 s_susp[256];                                     // this trip's suspects: {window lo, window hi, position lo, position hi | test << 8}
  __shared__ uint32_t s_work[EWORK];                                // items: suspect of the trip << 24 | ... no: local suspect id (8 bits) << 24 | offset into the run (24 bits)
  __shared__ uint32_t s_t0[256];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_fill, s_valid, s_full, s_nwork;
  unsigned long long n = *a.susp_count;
  if (n > a.susp_cap) n = a.susp_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { s_fill = 0; s_valid = (uint32_t)ESTAGE; s_nwork = 0; }
  __syncthreads();
  auto emit = [&](bool ok, uint64_t rec) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(ok);
    if (bal == 0) return;
    const int leader = __ffsll((long long)bal) - 1;
    const uint32_t cnt = (uint32_t)__popcll(bal), mine = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    uint32_t pos = 0;
    if (lane == leader) pos = atomicAdd(&s_fill, cnt);
    pos = __builtin_amdgcn_readlane(pos, leader);
    if (pos + cnt <= (uint32_t)ESTAGE) { if (ok) s_rec[pos + mine] = rec; return; }
    // no room in this trip's stage (every later reservation of the trip is refused too: s_fill stays above ESTAGE; the
    // flush takes the slots in front of the first refused position): straight to the list
    unsigned long long base = 0;
    if (lane == leader) { atomicMin(&s_valid, pos); base = atomicAdd(a.seed_count, (unsigned long long)cnt); }
    const uint32_t blo = __builtin_amdgcn_readlane((uint32_t)base, leader), bhi = __builtin_amdgcn_readlane((uint32_t)(base >> 32), leader);
    if (ok) { const unsigned long long o = (((unsigned long long)bhi << 32) | blo) + mine; if (o < a.seed_cap) a.seed_out[o] = rec; }
  };
  auto flush = [&]() __attribute__((always_inline)) {               // block-uniform call
    __syncthreads();
    const uint32_t cnt = min(s_fill, s_valid);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = cnt ? atomicAdd(a.seed_count, (unsigned long long)cnt) : 0ull; s_fill = 0; s_valid = (uint32_t)ESTAGE; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) if (s_base + i < a.seed_cap) a.seed_out[s_base + i] = s_rec[i];
    __syncthreads();
  };
  // one (suspect, pattern) item: the pattern's other fields through edit_cost, a seed record when it passes
  auto item = [&](uint32_t wlo, uint32_t whi, int var, uint64_t pos40, uint32_t t, bool live) __attribute__((always_inline)) {
    bool ok = false;
    uint64_t rec = 0;
    if (live) {
      const int table = edit_variant_table(var);
      const uint32_t o = a.olist[(size_t)table * a.np + t];
      int c = 9;
      edit_variant(var, [&](auto A_, auto B_, auto D_) __attribute__((always_inline)) { c = edit_cost<decltype(A_)::value, decltype(B_)::value, decltype(D_)::value>(o, wlo, whi); });
      ok = c <= 2;
      if (ok) rec = ((uint64_t)a.order[(size_t)table * a.np + t] << 40) | pos40;
    }
    emit(ok, rec);
  };
  for (unsigned long long base = (unsigned long long)blockIdx.x * blockDim.x; base < n; base += stride) {   // block-uniform trip count
    const unsigned long long i = base + threadIdx.x;
    uint32_t t0 = 0, t1 = 0;
    uint4 r = make_uint4(0, 0, 0, 0xffffffffu);
    if (i < n) r = a.susp[i];
    if (r.w != 0xffffffffu) {                                       // (all ones: a slot its wave reserved and did not need; a real record's test number is < 14)
      const int var = (int)((r.w >> 8) & 15u);
      uint32_t key = 0;
      edit_variant(var, [&](auto A_, auto B_, auto D_) __attribute__((always_inline)) { key = edit_key<decltype(A_)::value, decltype(B_)::value, decltype(D_)::value>(r.x, r.y); });
      const int table = edit_variant_table(var);
      const uint32_t row = key & 0x7fffu, bit = key >> 15;
      const uint32_t wd = a.image[(size_t)table * PAIR_BITMAP_WORDS + row];
      if ((wd >> bit) & 1u) {
        const uint32_t rank = a.row_base[(size_t)table * (PAIR_BITMAP_WORDS + 1) + row] + (uint32_t)__popc(wd & ((1u << bit) - 1u));
        const uint32_t *fp = a.first_pat + a.first_off[table];
        t0 = fp[rank]; t1 = fp[rank + 1];
      }
    }
    // Runs are one pattern long for most suspects and three or more for the flagged ones (keys with more patterns than a
    // slot holds): lanes walking their own runs idled for most of the trip.  The (suspect, pattern) items of the whole
    // workgroup go into one queue instead and every thread takes its share.
    s_susp[threadIdx.x] = r;
    s_t0[threadIdx.x] = t0;
    const uint32_t len = t1 - t0;
    uint32_t at = len ? atomicAdd(&s_nwork, len) : 0u;
    uint32_t q = 0;
    for (; q < len && at + q < (uint32_t)EWORK; ++q) s_work[at + q] = ((uint32_t)threadIdx.x << 24) | q;
    __syncthreads();
    const uint32_t nw = min(s_nwork, (uint32_t)EWORK);
    for (uint32_t w0 = 0; w0 < nw; w0 += blockDim.x) {              // block-uniform
      const uint32_t w = w0 + threadIdx.x;
      const bool live = w < nw;
      const uint32_t it = live ? s_work[w] : 0u;
      const uint4 sr = s_susp[it >> 24];
      item(sr.x, sr.y, (int)((sr.w >> 8) & 15u), ((uint64_t)(sr.w & 0xffu) << 32) | sr.z, s_t0[it >> 24] + (it & 0xffffffu), live);
    }
    // what did not fit the queue (a trip with a key shared by thousands of patterns): its owner walks it
    {
      uint32_t t = t0 + q;
      while (__ballot(t < t1)) { item(r.x, r.y, (int)((r.w >> 8) & 15u), ((uint64_t)(r.w & 0xffu) << 32) | r.z, t, t < t1); if (t < t1) ++t; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_full = s_fill > (uint32_t)(ESTAGE - 1024); s_nwork = 0; }
    __syncthreads();
    if (s_full) flush();
  }
  flush();
}

}  // namespace

// ---- host side: tables, launch -------------------------------------------------------------------

std::string pair_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids, const Alphabet &alpha, int k,
                       int eos_code, PairTables *out, int stride_knob, int slot_patterns) {
  const size_t SP = slot_patterns == 2 ? 2 : 3;                     // patterns a slot settles (the edit plan compares two: its compare is three times the substitution plan's)
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
  // Slots per row of the slot table: rows (32 keys each) hold a Poisson number of the keys that occur, 5.5 on
  // average at 200k patterns; the last slot of a row is its overflow marker.  12 slots = 96 bytes per row keep
  // a field pair's table at 3 MiB -- inside an XCD's 4 MiB of L2 -- with 0.3 % of the keys beyond their row
  // (3 Gbp x 200k patterns: 12 slots 14.8 ms, 14: 15.2, 16: 16.9 -- L2 misses; 10: 21.8 -- overflow suspects);
  // fuller tiles (the 250k-pattern tiles of a 10^6-primer set: 6.9 keys per row) take 14.
  t.stride = stride_knob >= 2 && stride_knob <= 33 ? stride_knob : (np <= 230000 ? 12 : (np <= 300000 ? 14 : 16));
  const size_t ST = (size_t)t.stride, KMAX = ST - 1;
  t.image.assign((size_t)C * PAIR_BITMAP_WORDS, 0);
  t.order.assign((size_t)C * np, 0);
  t.olist.assign((size_t)C * np, 0);
  t.slots.assign((size_t)C * PAIR_BITMAP_WORDS * ST, 0);
  t.row_base.assign((size_t)C * (PAIR_BITMAP_WORDS + 1), 0);
  std::vector<std::vector<uint32_t>> fp(C);
  auto build_combo = [&](int ci) {
    const int fa = t.fa[ci], fb = t.fb[ci];
    int fc, fd;
    other_fields(fa, fb, &fc, &fd);
    // sort key = position of the key's bit in the bitmap (row major), then pattern index
    std::vector<uint64_t> srt(np);
    for (size_t j = 0; j < np; ++j) {
      const uint32_t key = field_of(t.pat40[j], fa) | (field_of(t.pat40[j], fb) << 10);
      const uint32_t bitpos = ((key & 0x7fffu) << 5) | (key >> 15);
      srt[j] = ((uint64_t)bitpos << 32) | (uint64_t)j;
    }
    std::sort(srt.begin(), srt.end());
    uint32_t *img = &t.image[(size_t)ci * PAIR_BITMAP_WORDS];
    uint32_t *ord = &t.order[(size_t)ci * np];
    uint32_t *ol = &t.olist[(size_t)ci * np];
    uint64_t *slot = &t.slots[(size_t)ci * PAIR_BITMAP_WORDS * ST];
    uint32_t *rb = &t.row_base[(size_t)ci * (PAIR_BITMAP_WORDS + 1)];
    std::vector<uint32_t> &f = fp[ci];
    f.reserve(np + 1);
    for (size_t r = 0; r < (size_t)PAIR_BITMAP_WORDS; ++r) slot[r * ST + KMAX] = slot_pack(0, 0, 0, true);   // overflow markers
    uint32_t cur_row = 0, in_row = 0;
    for (size_t j = 0; j < np;) {
      const uint32_t bitpos = (uint32_t)(srt[j] >> 32);
      size_t j2 = j;
      while (j2 < np && (uint32_t)(srt[j2] >> 32) == bitpos) ++j2;
      const uint32_t row = bitpos >> 5;
      while (cur_row < row) { rb[++cur_row] = (uint32_t)f.size(); in_row = 0; }
      img[row] |= 1u << (bitpos & 31u);
      f.push_back((uint32_t)j);
      uint32_t o[3] = {0, 0, 0};
      for (size_t q = j; q < j2; ++q) {
        const uint32_t pi = (uint32_t)srt[q];
        ord[q] = pi;
        ol[q] = field_of(t.pat40[pi], fc) | (field_of(t.pat40[pi], fd) << 10);
        if (q - j < SP) o[q - j] = field_of(t.pat40[pi], fc) | (field_of(t.pat40[pi], fd) << 10);
      }
      for (size_t q = std::min(j2 - j, SP); q < 3; ++q) o[q] = o[0];   // free fields repeat the first pattern
      if (in_row < KMAX) slot[(size_t)row * ST + in_row] = slot_pack(o[0], o[1], o[2], j2 - j > SP);
      ++in_row;
      j = j2;
    }
    while (cur_row < (uint32_t)PAIR_BITMAP_WORDS) rb[++cur_row] = (uint32_t)f.size();
    f.push_back((uint32_t)np);
  };
  if (np * (size_t)C < 20000) for (int ci = 0; ci < C; ++ci) build_combo(ci);
  else {
    std::vector<std::thread> th;
    for (int ci = 0; ci < C; ++ci) th.emplace_back([&, ci]() { build_combo(ci); });
    for (std::thread &x : th) x.join();
  }
  for (int ci = 0; ci < C; ++ci) {
    t.first_off[ci] = t.first_pat.size();
    t.first_pat.insert(t.first_pat.end(), fp[ci].begin(), fp[ci].end());
  }
  return "";
}

hipError_t pair_upload(const PairTables &t, PairDevice *d, hipStream_t st) {
  pair_free(d);
  d->k = t.k; d->maxlen = t.maxlen; d->ncombos = t.ncombos; d->eos_code = t.eos_code; d->ascii = t.ascii; d->np = t.pat40.size(); d->stride = t.stride;
  for (int c = 0; c < PAIR_MAX_COMBOS; ++c) { d->fa[c] = t.fa[c]; d->fb[c] = t.fb[c]; d->first_off[c] = t.first_off[c]; }
  auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
  };
  hipError_t e;
  if ((e = up(t.image.data(), t.image.size() * 4, (void **)&d->image)) != hipSuccess) return e;
  if ((e = up(t.slots.data(), t.slots.size() * 8, (void **)&d->slots)) != hipSuccess) return e;
  if ((e = up(t.row_base.data(), t.row_base.size() * 4, (void **)&d->row_base)) != hipSuccess) return e;
  if ((e = up(t.first_pat.data(), t.first_pat.size() * 4, (void **)&d->first_pat)) != hipSuccess) return e;
  if ((e = up(t.order.data(), t.order.size() * 4, (void **)&d->order)) != hipSuccess) return e;
  if ((e = up(t.olist.data(), t.olist.size() * 4, (void **)&d->olist)) != hipSuccess) return e;
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
  void *ptrs[] = {d->image, d->slots, d->row_base, d->first_pat, d->order, d->olist, d->pat_id, d->pat40, d->pat_len, d->pat_codes, d->pat_zone};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  *d = PairDevice();
}

ScanGeometry pair_geometry(const PairDevice &d, int64_t begin, int64_t end) {
  ScanGeometry g;
  int64_t chunk = 1 << 19;                                         // 512 Ki positions per workgroup
  // larger ranges: 1 Mi or 2 Mi positions per workgroup amortise staging the 128 KiB bitmap and the pipeline's fill and
  // drain, as long as the grid keeps >= 16 workgroups per CU for its tail (3 Gbp: -K 2, 6 field pairs, 2 Mi = 8586
  // workgroups; -K 1, 2 field pairs: 5.23 ms with 1 Mi against 5.38 with 2 Mi and 5.36 with 512 Ki; a 375 Mbp shard
  // of a position-sharded -K 2 scan: 2.09 ms with 512 Ki, 2.31 with 1 Mi, 2.50 with 2 Mi)
  const int64_t range = end - begin;
  if (range / ((int64_t)1 << 21) * d.ncombos >= 256 * 16) chunk = (int64_t)1 << 21;
  else if (range / ((int64_t)1 << 20) * d.ncombos >= 256 * 16) chunk = (int64_t)1 << 20;
  if (d.knobs.seed_chunk >= 1024 * WAVES) chunk = d.knobs.seed_chunk / (1024 * WAVES) * (1024 * WAVES);   // test knob (shared with the seed kernels)
  g.seg_len = chunk;
  const int64_t c_lo = begin / chunk, c_hi = end > begin ? (end - 1) / chunk : c_lo - 1;
  g.nseg = (int)(c_hi - c_lo + 1);
  g.threads = PAIR_THREADS;
  g.blocks = g.nseg * d.ncombos;
  return g;
}

hipError_t pair_launch(const PairDevice &d, const uint8_t *d_text, const uint32_t *d_packed, int64_t n, int64_t begin, int64_t end,
                       pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, void *d_susp, unsigned long long *d_susp_count, uint64_t susp_cap,
                       hipStream_t st, ScanGeometry *geo_out, unsigned long long *d_stats, int floor_mode,
                       uint64_t *seed_out, unsigned long long *seed_count, uint64_t seed_cap) {
  if (!d_packed) return hipErrorInvalidValue;
  if (end > n) end = n;
  if (floor_mode && (d.k != 2 || d.ncombos != 6)) return hipErrorInvalidValue;
  if (floor_mode == 3 && (!seed_out || !seed_count)) return hipErrorInvalidValue;
  ScanGeometry g = pair_geometry(d, begin, end);
  if (floor_mode) g.blocks = g.nseg * 14;
  if (geo_out) *geo_out = g;
  if (g.blocks <= 0 || d.np == 0) return hipSuccess;
  PairArgs a;
  memset(&a, 0, sizeof(a));
  a.text = d_text; a.n = n; a.begin = begin; a.end = end;
  a.packed = d_packed; a.npacked = (n + 15) / 16;
  a.chunk_len = g.seg_len; a.chunk0 = begin / g.seg_len; a.nchunks = g.nseg; a.ncombos = d.ncombos;
  a.group = 256;
  if (d.knobs.seed_group > 0) a.group = d.knobs.seed_group;
  a.k = d.k; a.eos_code = d.eos_code;
  a.debug = d.knobs.seed_debug;
  for (int c = 0; c < PAIR_MAX_COMBOS; ++c) { a.fa[c] = d.fa[c]; a.fb[c] = d.fb[c]; a.first_off[c] = (uint32_t)d.first_off[c]; }
  a.stride = d.stride;
  a.image = d.image; a.slots = reinterpret_cast<const uint2 *>(d.slots); a.row_base = d.row_base; a.first_pat = d.first_pat; a.order = d.order; a.olist = d.olist;
  a.np = (uint32_t)d.np;
  a.pat40 = reinterpret_cast<const uint2 *>(d.pat40); a.pat_len = d.pat_len; a.pat_id = d.pat_id; a.pat_codes = d.pat_codes; a.pat_zone = d.pat_zone; a.viol_level = d.viol_level;
  a.out = d_out; a.counter = d_counter; a.cap = cap;
  if (!d_susp || !d_susp_count) return hipErrorInvalidValue;
  a.susp = reinterpret_cast<uint4 *>(d_susp); a.susp_count = d_susp_count; a.susp_cap = susp_cap;   // *d_susp_count zeroed by the caller (stream order)
  a.stats = d_stats;
  a.seed_out = seed_out; a.seed_count = seed_count; a.seed_cap = seed_cap;
  if (floor_mode) {                                                 // measurement: 14 (pair, displacement) variants, no verify kernel
    a.ncombos = 14;
    hipError_t fe;
    if (floor_mode == 1) {
      if ((fe = hipFuncSetAttribute(reinterpret_cast<const void *>(pm_pair_edit_scan<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES)) != hipSuccess) return fe;
      hipLaunchKernelGGL(pm_pair_edit_scan<1>, dim3(g.blocks), dim3(PAIR_THREADS), PAIR_LDS_BYTES, st, a);
    } else if (floor_mode == 2) {
      if ((fe = hipFuncSetAttribute(reinterpret_cast<const void *>(pm_pair_edit_scan<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES)) != hipSuccess) return fe;
      hipLaunchKernelGGL(pm_pair_edit_scan<2>, dim3(g.blocks), dim3(PAIR_THREADS), PAIR_LDS_BYTES, st, a);
    } else {                                                        // the edit plan: scan, then suspects -> seed records
      if ((fe = hipFuncSetAttribute(reinterpret_cast<const void *>(pm_pair_edit_scan<3>), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES)) != hipSuccess) return fe;
      hipLaunchKernelGGL(pm_pair_edit_scan<3>, dim3(g.blocks), dim3(PAIR_THREADS), PAIR_LDS_BYTES, st, a);
      if ((fe = hipGetLastError()) != hipSuccess) return fe;
      hipLaunchKernelGGL(pm_pair_edit_resolve, dim3(16384), dim3(256), 0, st, a);
    }
    return hipGetLastError();
  }
  hipLaunchKernelGGL(pm_pair_scan, dim3(g.blocks), dim3(PAIR_THREADS), PAIR_LDS_BYTES, st, a);
  hipError_t ce = hipGetLastError();
  if (ce != hipSuccess) return ce;
  // one suspect per thread for up to 8 Mi of them (the count is on the device): the verify is a chain of ~8 dependent
  // table and stream reads per suspect; six suspects per thread one after the other made every wave live 0.9 ms
  hipLaunchKernelGGL(pm_pair_verify, dim3(32768), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace pm

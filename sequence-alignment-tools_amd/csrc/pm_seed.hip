// pm_seed.hip -- seed-filter scan kernel for gfx950: 2-bit packed windows, per-combo Bloom filter
// in LDS, ballot/popcount compaction of survivors, hash-table verify.
//
// What it computes: every (end, pattern, d) with d = Hamming(stream window, pattern) <= k and no
// EOS inside the window -- the hit set of the reference's exact engines for k = 0
// (keyword_tree.t:427-486, shift_and.cc:208-255) and the candidate set of its substitution-only
// k-error automaton (shift_and_inexact.cc:249-352 with indels=false), on which filter_bitvec and
// exact_halves build.  Patterns are A/C/G/T strings of <= 32 characters.
//
// How (pigeonhole, DESIGN.md "seed family"): the last Lw characters of every pattern are cut into
// m = k + r pieces of pb characters; <= k substitutions leave >= r pieces untouched, so a true
// candidate agrees with the pattern on at least one of the C(m,r) piece combinations ("combos").
//   * a WORKGROUP owns one (combo, stream chunk): it stages that combo's 128 KiB blocked Bloom
//     filter (3 bits per key inside one dword) in LDS, then streams its chunk;
//   * a LANE loads 16 stream bytes with one 16-byte load (a wave reads 1 KiB contiguous), packs
//     them to 2 bits per base, borrows the previous 32 bases from its neighbours with two wave
//     shuffles, and tests its 16 windows: funnel shift -> combo key -> one ds_read_b32;
//   * survivors (~9 % per test) are compacted wave-wide with ballot + mbcnt into a per-wave LDS
//     queue, so the second stage runs on full waves: probe the combo's open-addressing table
//     (8-byte slots, L2 resident: consecutive workgroups work on the same combo), XOR + popcount
//     the packed window against the pattern, and only then re-read the raw stream bytes for the
//     exact distance (N = mismatch, EOS = reject);
//   * a (window, pattern) pair reachable through several combos is reported by exactly one: the
//     combo made of its first r clean pieces.
//
// Cost model: ~25 VALU + 1 LDS read per (position, combo), ~0.9*C/10 L2 probes per position.
// Bound: LDS/VALU issue, then L2 request rate; HBM traffic is 1 byte per base per MALL pass.
//
// In this file, in order: the helpers all seed kernels share (packed loads, hashes, the partner test of
// exact_halves, the exact verify), pm_seed_scan<LW,MODE,HALVES,EDITS> (the plan above: k = 0, windows
// shorter than 20, and the round-1 forms of the two plans below), pm_edit_scan (first stage of the
// edit-distance plan: tabulated piece hashes, key map in L2), pm_half_scan (exact_halves -k on a key
// bitmap + rank directory), the dense second kernels pm_edits_verify / pm_half_verify / pm_bases_verify,
// pm_pack_stream, and the host side (edit_cover, seed_build, seed_upload, seed_launch).
#include "pm_internal.h"
#include "pm_seed.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <type_traits>
#include <utility>

namespace pm {

namespace {

constexpr int QCAP = SEED_QCAP;           // queue entries per wave (8 B each)
constexpr int WAVES = SEED_THREADS / 64;
constexpr uint32_t EMPTY = 0xffffffffu;
constexpr uint32_t HASH_LO = 0x9E3779B1u, HASH_HI = 0x9E3779u, HASH_SLOT = 0x85EBCA6Bu, HASH_SEL = 0xC2B2AFu;

struct SeedArgs;
struct SeedArgs {
  const uint8_t *text;
  int64_t n, begin, end;                // owned hit ends: begin < end_pos <= end
  int64_t chunk0;                       // first chunk index (absolute, chunk_len aligned)
  int64_t chunk_len;                    // bytes per workgroup, multiple of 1024*WAVES
  int nchunks, ncombos, group;          // group = chunks per (superchunk, combo) run
  int k, Lw, pb, r, ascii, debug, maxlen;
  uint32_t mask_lo[SEED_MAX_COMBOS];    // window bits (2 per base) that belong to the combo's pieces
  uint32_t mask_hi[SEED_MAX_COMBOS];
  uint32_t perm_sel[SEED_MAX_COMBOS];   // byte-aligned plans (pb == 4): v_perm selector gathering the pieces
  const uint32_t *bloom;                // [combo][SEED_BLOOM_STRIDE]
  const uint4 *buckets;                 // [combo][nbuckets][2]: 8 slots of (fingerprint << idx_bits | pattern index)
  uint32_t bucket_shift, idx_bits;      // bucket = h2 >> bucket_shift; nbuckets = 2^(32-bucket_shift)
  const uint32_t *bitmap2;              // [combo][2^(lb2-5)] second-level one-bit filter (L2 resident)
  uint32_t lb2;                         // log2 of its size in bits
  const uint2 *pat40;                   // packed last Lw bases of every pattern
  const uint8_t *pat_len;
  const uint32_t *pat_id;
  const uint8_t *pat_codes;             // stream codes of every pattern char, 32 bytes per pattern
  const uint8_t *cmap;                  // stream code -> 1 for EOS, else 0
  pm_hit *out;
  unsigned long long *counter;
  unsigned long long cap;
  int eos_code;                         // stream code of the end-of-sequence character, -1 = none
  int halves, hk;                       // exact_halves -k: patterns are halves, partner prefilter for hk edits
  int edits;                            // > 0: filter_bitvec / shift_and_inexact with indels; pat_codes holds 32-byte automaton records
  uint64_t *seed_out;                   // EDITS scan: 8-byte seed records (pattern index << 40 | position), ~0 = unused slot
  uint32_t emask_a[SEED_MAX_COMBOS], emask_b[SEED_MAX_COMBOS];   // byte masks (low window word) of the combo's first and second piece
  uint32_t evar[SEED_MAX_COMBOS];       // bit v: displacement pattern v is tested for this combo (edit_cover on the host)
  int exact_filter;                     // halves with keys of <= 20 bits (one piece of <= 10 bases): the LDS filter is the exact
                                        // key bitmap (bit = key), no false positives, and the second-level bitmap is skipped
  int hfast;                            // > 0: every pattern has this length and half j lies on side j & 1, so the
                                        // partner's stream window is known before the half's record is read
  const uint32_t *part32;               // partner half, 2 bits per base (<= 16 bases)
  const uint8_t *part_len, *part_side;  // its length; 0 = partner lies to the right of the seed, 1 = to the left
  const uint32_t *packed;               // the stream, 2 bits per base, 16 bases per dword (pack_stream); the first stage reads this
  const uint8_t *etable;                // pm_edit_scan: [combo][2^et_bytes_log bytes] bit map of the key hashes
  const uint32_t *eidx;                 // pm_edit_scan: pattern index of every bucket slot
  const uint32_t *hr_image;             // pm_half_scan: key bitmap + rank directory (HR_IMAGE_WORDS)
  const uint2 *hr_slots, *hr_more;      // one slot per distinct key (by rank); further halves of a key
  const uint32_t *hr_first, *hr_order;  // pm_half_verify: halves of every key (first index by rank; pattern indices sorted by key)
  int et_shift, et_bytes_log;           // the map: H >> et_shift = byte offset of the key's dword (& et_mask), 2^et_bytes_log bytes per combo
  uint32_t et_mask;
  int64_t npacked;                      // dwords in `packed`
  const SeedArgs *self;                 // device copy of this struct, for the out-of-line rare paths
};

__device__ __forceinline__ uint32_t pack4(uint32_t x, int sh) {
  // 4 stream bytes -> 8 bits, 2 bits per base (byte 0 in bits 0-1)
  uint32_t y = (x >> sh) & 0x03030303u;
  y |= y >> 6;
  return (y | (y >> 12)) & 0xffu;
}

__device__ __forceinline__ uint32_t pack16(const uint4 &v, int sh) {
  return pack4(v.x, sh) | (pack4(v.y, sh) << 8) | (pack4(v.z, sh) << 16) | (pack4(v.w, sh) << 24);
}

__device__ __noinline__ uint4 load16_edge(const uint8_t *text, int64_t off, int64_t n) {
  uint32_t w[4] = {0, 0, 0, 0};
  for (int b = 0; b < 16; ++b) {
    const int64_t p = off + b;
    if (p >= 0 && p < n) w[b >> 2] |= (uint32_t)text[p] << (8 * (b & 3));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

#ifndef PM_TESTS_GROUP
#define PM_TESTS_GROUP 8
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte stream load.  NT (single-combo plans: the stream is read exactly once) marks it
// non-temporal so that it does not push the bucket table out of L2; multi-combo plans want the
// superchunk to stay in MALL for the following combos and use the default policy.
template <bool NT>
__device__ __forceinline__ uint4 load16(const uint8_t *text, int64_t off, int64_t n) {
  if (off >= 0 && off + 16 <= n) {
    if (NT) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(text + off));
      return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4 *>(text + off);
  }
  return load16_edge(text, off, n);
}

// One dword of the 2-bit packed stream = the 16 bases from `pos` (a multiple of 16) on; zero
// outside the stream, like the bytes load16_edge hands out.
template <bool NT>
__device__ __forceinline__ uint32_t load_packed(const uint32_t *packed, int64_t npacked, int64_t pos) {
  const int64_t i = pos >> 4;
  if (pos < 0 || i >= npacked) return 0u;
  return NT ? __builtin_nontemporal_load(packed + i) : packed[i];
}

// Hash of the combo's part of a window.  MODE 0: any piece layout (mask + fold + multiply);
// MODE 1: 4-base pieces = bytes, three of them gathered by one v_perm, 24-bit multiply;
// MODE 2: four byte pieces, 32-bit multiply.  The Bloom word index is the top 15 bits; *selsrc is
// what bloom_selectors derives the three bit selectors from (seed_build mirrors both on the host).
template <int MODE>
__device__ __host__ __forceinline__ uint32_t window_hash(uint32_t wlo, uint32_t whi, uint32_t mlo, uint32_t mhi, uint32_t sel, uint32_t *selsrc = nullptr) {
  if (MODE == 0) {
    uint32_t x = (wlo & mlo) + (whi & mhi) * HASH_HI;
    x ^= x >> 16;
    x *= HASH_LO;
    if (selsrc) *selsrc = x >> 8;
    return x;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t key = __builtin_amdgcn_perm(whi, wlo, sel);
  if (MODE == 1) { if (selsrc) *selsrc = key; return __umul24(key, HASH_HI); }
  const uint32_t h = key * HASH_LO;
  if (selsrc) *selsrc = h >> 8;
  return h;
#else
  uint32_t key = 0;
  const uint64_t W = ((uint64_t)whi << 32) | wlo;
  for (int q = 0; q < 4; ++q) {
    const uint32_t sb = (sel >> (8 * q)) & 0xffu;
    if (sb < 8) key |= (uint32_t)((W >> (8 * sb)) & 0xffu) << (8 * q);
  }
  if (MODE == 1) { if (selsrc) *selsrc = key; return (uint32_t)((uint64_t)(key & 0xffffffu) * HASH_HI); }
  const uint32_t h = key * HASH_LO;
  if (selsrc) *selsrc = h >> 8;
  return h;
#endif
}

// The three bit selectors of a key are the low five bits of bytes 1, 2, 3 of a second 24-bit
// product: each test is then one SDWA shift (the byte is picked by the operand selector, the
// shifter only looks at five bits).  seed_build mirrors bloom_selectors on the host.
__device__ __host__ __forceinline__ uint32_t bloom_selectors(uint32_t selsrc) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(selsrc, HASH_SEL);
#else
  return (uint32_t)((uint64_t)(selsrc & 0xffffffu) * HASH_SEL);
#endif
}

// bit 0 = 1 iff all three selected bits of the Bloom word are set
// The filter block of a key: the dword at LDS byte address (h >> 15) & ~3 (the 15 best-mixed hash
// bits).  The address is used as it is -- the filter sits at LDS address 0 (checked at kernel
// start) -- which saves the add of a block base the compiler cannot fold for dynamic LDS.
// Byte-granular blocks (no mask) would save one more instruction and gfx950 does serve unaligned
// ds_read_b32 (scripts/probe/lds_unaligned.hip), but at a price: the scan kernels ran 2x slower.
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
constexpr uint32_t BLOOM_ADDR_MASK = 0x1fffcu;
__device__ __host__ __forceinline__ uint32_t bloom_addr(uint32_t h) { return (h >> 15) & BLOOM_ADDR_MASK; }
__device__ __forceinline__ uint32_t bloom_block(uint32_t h) {
  return *reinterpret_cast<lds_u32 *>((uintptr_t)bloom_addr(h));
}

__device__ __forceinline__ uint32_t bloom_test(uint32_t word, uint32_t hsel) {
  uint32_t a, b, c;
  asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(hsel), "v"(word));
  asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(b) : "v"(hsel), "v"(word));
  asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(c) : "v"(hsel), "v"(word));
  return a & b & c;
}

// exact_halves with edits (-k): the "patterns" are halves and a seed is an exact occurrence of one
// (reference exact_halves.cc:199-224).  The reference then runs a banded DP of the partner half
// next to every seed (primer_alignment.cc:568-728) -- 0.4 seeds per base at 10^5 primers.  A DP
// with <= k edits leaves one of k+1 pieces of the partner untouched, displaced by at most k, so
// this cheap necessary test on 2-bit packed bases removes >99 % of the seeds on the GPU; only the
// rest are emitted as seed records for the DP (host stage for now).
__device__ __forceinline__ int64_t partner_window(int64_t p, int total, int side, int k) {
  // first base of the 24-base stream window the partner test looks at (total = half + partner length)
  return (side ? (p + 1 - total) : (p + 1)) - k;
}

// T = the stream window from partner_window on, 2 bits per base (base i at bits 2i)
__device__ __forceinline__ bool partner_possible_packed(int k, int plen, uint32_t part, uint64_t T) {
  // piece t of the partner = bases [t*plen/(k+1), (t+1)*plen/(k+1)); it sits unedited at displacement s
  // iff the XOR of the stream (shifted by k+s bases) with the partner is zero over the piece's bits
  auto low = [](int bases) -> uint32_t { return bases >= 16 ? 0xffffffffu : ((1u << (2 * bases)) - 1u); };
  if (k == 1) {
    const uint32_t m0 = low(plen >> 1), m1 = low(plen) & ~m0;
    bool ok = false;
#pragma unroll
    for (int s = 0; s <= 2; ++s) {
      const uint32_t x = (uint32_t)(T >> (2 * s)) ^ part;
      ok = ok || (x & m0) == 0 || (x & m1) == 0;
    }
    return ok;
  }
  if (k == 2) {
    const int e0 = plen / 3, e1 = 2 * plen / 3;
    const uint32_t m0 = low(e0), m1 = low(e1) & ~m0, m2 = low(plen) & ~low(e1);
    bool ok = false;
#pragma unroll
    for (int s = 0; s <= 4; ++s) {
      const uint32_t x = (uint32_t)(T >> (2 * s)) ^ part;
      ok = ok || (x & m0) == 0 || (x & m1) == 0 || (x & m2) == 0;
    }
    return ok;
  }
  const uint64_t PP = part;
  for (int t = 0; t <= k; ++t) {
    const int o = t * plen / (k + 1), len = (t + 1) * plen / (k + 1) - o;
    const uint64_t mask = (1ull << (2 * len)) - 1ull;
    const uint64_t piece = (PP >> (2 * o)) & mask;
    for (int s = -k; s <= k; ++s)
      if (((T >> (2 * (k + o + s))) & mask) == piece) return true;
  }
  return false;
}

__device__ __forceinline__ bool partner_possible(const SeedArgs &a, int plen, uint32_t part, uint64_t raw0, uint64_t raw1, uint64_t raw2) {
  uint64_t T = 0;                                                   // 2 bits per base, base i of the window at bits 2i
  const int sh = a.ascii ? 1 : 0;
  const uint64_t raws[3] = {raw0, raw1, raw2};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const uint32_t lo = pack4((uint32_t)raws[q], sh), hi = pack4((uint32_t)(raws[q] >> 32), sh);
    T |= (uint64_t)(lo | (hi << 8)) << (16 * q);
  }
  return partner_possible_packed(a.hk, plen, part, T);
}

// Third stage, exact part: (window ending at p, pattern pi) already passed the packed-distance
// test; count mismatches on the raw stream codes (N = mismatch, EOS = reject) and report.
__device__ __noinline__ void verify_exact(const SeedArgs *ap, uint32_t mlo, uint32_t mhi, int64_t p, uint32_t pi) {
  const SeedArgs &a = *ap;
  const int L = a.pat_len[pi];
  const int64_t start = p + 1 - L;
  if (start < 0) return;
  // four stream bytes against four pattern codes per step, per-byte verdicts by SWAR.  Rolled on
  // purpose: this function must stay small in registers, or the callers' allocation suffers.
  const uint32_t *pc = reinterpret_cast<const uint32_t *>(a.pat_codes + (size_t)pi * 32);
  const uint32_t eb = (uint32_t)(a.eos_code & 0xff) * 0x01010101u;
  uint32_t mism = 0, eos = 0;                         // bit i: stream byte i differs from the pattern / is EOS
#pragma unroll 1
  for (int d = 0; 4 * d < L; ++d) {
    const int64_t off = start + 4 * d;
    uint32_t tw = 0;
    if (off + 4 <= a.n) __builtin_memcpy(&tw, a.text + off, 4);
    else
      for (int b = 0; off + b < a.n; ++b) tw |= (uint32_t)a.text[off + b] << (8 * b);   // last bytes of the stream
    const uint32_t x = tw ^ pc[d], z = tw ^ eb;
    const uint32_t y = (x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t e = ~(z | ((z & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    mism |= (((y >> 7) & 1u) | ((y >> 14) & 2u) | ((y >> 21) & 4u) | ((y >> 28) & 8u)) << (4 * d);
    eos |= (((e >> 7) & 1u) | ((e >> 14) & 2u) | ((e >> 21) & 4u) | ((e >> 28) & 8u)) << (4 * d);
  }
  const uint32_t lenmask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
  mism &= lenmask;
  if (a.eos_code >= 0 && (eos & lenmask)) return;     // EOS inside the window: never a candidate
  const int ham = __popc(mism);                       // N (or any other code) = mismatch
  if (ham > a.k) return;
  const int half = L / 2, m = a.k + a.r;
  const bool left_clean = (mism & ((1u << half) - 1u)) == 0, right_clean = (mism >> half) == 0;
  uint32_t dirty = 0;                                 // pieces (of the last Lw bases) with a mismatch
  const uint32_t sm = mism >> (L - a.Lw), pmask = (1u << a.pb) - 1u;
  for (int j = 0; j < m; ++j)
    if ((sm >> (j * a.pb)) & pmask) dirty |= 1u << j;
  // report once: only through the combo made of the first r clean pieces
  uint64_t first = 0;
  const uint64_t field = (1ull << (2 * a.pb)) - 1ull;
  for (int j = 0, t = 0; j < m && t < a.r; ++j)
    if (!(dirty >> j & 1)) { first |= field << (2 * a.pb * j); ++t; }
  if (first != (((uint64_t)mhi << 32) | mlo)) return;
  const unsigned long long o = atomicAdd(a.counter, 1ull);
  if (o < a.cap) {
    pm_hit hh;
    hh.end = p + 1; hh.pid = a.pat_id[pi]; hh.k = (uint8_t)ham;
    hh.aux[0] = (uint8_t)((left_clean ? 1 : 0) | (right_clean ? 2 : 0)); hh.aux[1] = hh.aux[2] = 0;
    a.out[o] = hh;
  }
}

// exact_halves -k, one seed (half pi ends at p).  Record of a half (seed_build), 32 bytes:
// codes right-aligned in bytes [16-L, 16) | partner 2-bit | L, partner length, side | inner id.
// th0/th1 = stream bytes [p-15, p-8], [p-7, p]: the half matches iff its codes equal the last L of
// them (codes are all A,C,G,T, so N and EOS inside the window reject by themselves).
__device__ __forceinline__ bool half_codes_equal(const uint4 &c, int L, uint64_t th0, uint64_t th1) {
  const uint64_t c0 = ((uint64_t)c.y << 32) | c.x, c1 = ((uint64_t)c.w << 32) | c.z;
  const uint64_t m1 = L >= 8 ? ~0ull : (~0ull << (8 * (8 - L)));
  const uint64_t m0 = L <= 8 ? 0ull : (L >= 16 ? ~0ull : (~0ull << (8 * (16 - L))));
  return (((th0 ^ c0) & m0) | ((th1 ^ c1) & m1)) == 0;
}

__device__ __forceinline__ void load_tail16(const SeedArgs &a, int64_t p, uint64_t &th0, uint64_t &th1) {
  th0 = th1 = 0;
  if (p >= 15) {                                       // two unaligned 8-byte loads
    __builtin_memcpy(&th0, a.text + p - 15, 8);
    __builtin_memcpy(&th1, a.text + p - 7, 8);
  } else {
    for (int i = 0; i <= (int)p; ++i) {                // the very start of the stream
      const uint64_t b = a.text[p - i];
      if (i < 8) th1 |= b << (8 * (7 - i)); else th0 |= b << (8 * (15 - i));
    }
  }
}

// all loads dependent (rare paths, and pattern sets of mixed length)
__device__ __forceinline__ bool half_seed_ok(const SeedArgs &a, int64_t p, uint32_t pi, uint32_t *pid) {
  const uint4 *rec = reinterpret_cast<const uint4 *>(a.pat_codes + (size_t)pi * 32);
  const uint4 c = rec[0], r = rec[1];
  const int L = r.y & 0xffu, plen = (r.y >> 8) & 0xffu, side = (r.y >> 16) & 0xffu;
  if (p + 1 - L < 0) return false;
  uint64_t th0, th1;
  load_tail16(a, p, th0, th1);
  if (!half_codes_equal(c, L, th0, th1)) return false;
  *pid = r.z;
  const int64_t r0 = partner_window(p, L + plen, side, a.hk);
  if (r0 < 0 || r0 + 24 > a.n) return true;            // near the stream ends: let the DP decide
  uint64_t w0, w1, w2;
  __builtin_memcpy(&w0, a.text + r0, 8);
  __builtin_memcpy(&w1, a.text + r0 + 8, 8);
  __builtin_memcpy(&w2, a.text + r0 + 16, 8);
  return partner_possible(a, plen, r.x, w0, w1, w2);
}

__device__ __forceinline__ pm_hit half_seed_record(int64_t p, uint32_t pid) {
  pm_hit hh;
  hh.end = p + 1; hh.pid = pid; hh.k = 0;
  hh.aux[0] = 3; hh.aux[1] = hh.aux[2] = 0;
  return hh;
}

// out-of-line copy for the rare paths (second match in a bucket, probe continuation).  The main
// path of the HALVES kernel inlines the test -- every seed gets there -- and writes its records into
// slots it reserves 64 at a time: one atomic on the shared counter per record serialises the chip
// (10^7 same-address atomics cost 25 ms).  Unused slots of a block are marked PM_SEED_HOLE.
__device__ __noinline__ void verify_half(const SeedArgs *ap, int64_t p, uint32_t pi) {
  const SeedArgs &a = *ap;
  uint32_t pid = 0;
  if (!half_seed_ok(a, p, pi, &pid)) return;
  const unsigned long long o = atomicAdd(a.counter, 1ull);
  if (o < a.cap) a.out[o] = half_seed_record(p, pid);
}

template <bool HALVES>
__device__ __forceinline__ void verify_hit(const SeedArgs *ap, uint32_t mlo, uint32_t mhi, int64_t p, uint32_t pi) {
  if (HALVES) verify_half(ap, p, pi); else verify_exact(ap, mlo, mhi, p, pi);
}

// ---- edits (-k with filter_bitvec / shift_and_inexact semantics) --------------------------------
// The candidates of these engines are the end positions at which the Wu-Manber k-error automaton
// (shift_and_inexact.cc:249-352) has the pattern's last bit set.  A seed (3 clean pieces under one
// of the displacement patterns) says "pattern pi ends within +-k of p+1"; the automaton is then
// run for that one pattern over the L+k+4 stream characters in front of p+3, from the empty state
// (its last bit depends on the last L+k characters only), and the ends p-1..p+3 are read off.
// Record (seed_build): masks of A,C,G,T over the pattern positions (bit i = character i) | L | id.
// Returns one nibble per end p-1+d, d = 0..4: level+1, or 0.
__device__ __forceinline__ uint32_t edits_verify(const SeedArgs &a, int64_t p, uint32_t pi, uint32_t *pid) {
  const uint4 *rec = reinterpret_cast<const uint4 *>(a.pat_codes + (size_t)pi * 32);
  const uint4 M = rec[0], r = rec[1];
  const int L = (int)(r.x & 0xffu), k = a.edits;
  *pid = r.y;
  // characters [base + skip, p+3) are consumed: maxlen+k+4 of them, so that every end p-1..p+3 has
  // its L+k characters; an earlier start only adds true history
  const int nch = (a.maxlen + k + 4 + 15) >> 4;                   // wave-uniform number of 16-byte pieces
  const int skip = 16 * nch - (a.maxlen + k + 4);                 // wave-uniform: leading characters of the first piece not needed
  const int64_t base = p + 3 - 16 * (int64_t)nch;
  uint32_t R0 = 0, R1 = 0, R2 = 0;
  if (base + skip <= 0) { R1 = 1u; R2 = 3u; }                     // true start of the stream: l prefix bits in row l (:162-164)
  const uint32_t last = 1u << (L - 1);
  uint32_t res = 0;
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const int64_t off = base + 16 * c;
    uint4 v;
    const bool inside = off >= 0 && off + 16 <= a.n;
    if (inside) __builtin_memcpy(&v, a.text + off, 16);
    else {
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int64_t q = off + b;
        const uint32_t x = (q >= 0 && q < a.n) ? (uint32_t)a.text[q] << (8 * (b & 3)) : 0u;
        if (b < 4) w[0] |= x; else if (b < 8) w[1] |= x; else if (b < 12) w[2] |= x; else w[3] |= x;
      }
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      if (c == 0 && b < skip) continue;                            // wave-uniform
      const uint32_t ch = (vw[b >> 2] >> (8 * (b & 3))) & 0xffu;
      uint32_t U;
      if (a.ascii) U = ch == 'A' ? M.x : ch == 'C' ? M.y : ch == 'G' ? M.z : ch == 'T' ? M.w : 0u;
      else {
        const uint32_t u01 = (ch & 1u) ? M.y : M.x, u23 = (ch & 1u) ? M.w : M.z;
        U = ch > 3u ? 0u : ((ch & 2u) ? u23 : u01);
      }
      // one character (pm_bitpar.hip step<K, true>): substitution, insertion and deletion terms
      const uint32_t x0 = (R0 << 1) | 1u, m1 = x0 | R0;
      uint32_t n0 = x0 & U;
      const uint32_t x1 = (R1 << 1) | 1u;
      uint32_t n1 = (x1 & U) | m1 | (n0 << 1) | 1u | n0;
      uint32_t n2 = 0;
      if (k >= 2) {                                                // wave-uniform
        const uint32_t m2 = x1 | R1, x2 = (R2 << 1) | 1u;
        n2 = (x2 & U) | m2 | (n1 << 1) | 1u | n1;
      }
      bool live = true;
      if (!inside) { const int64_t t = off + b; live = t >= 0 && t < a.n; }   // only at the stream's ends
      if ((int)ch == a.eos_code) { n0 = 0; n1 = 0; n2 = 0; }      // EOS clears every row
      if (live) { R0 = n0; R1 = n1; R2 = n2; }
      if (c == nch - 1 && b >= 11) {                               // ends p-1 .. p+3 = after the characters p-2 .. p+2
        const uint32_t lvl = (R0 & last) ? 1u : (R1 & last) ? 2u : (R2 & last) ? 3u : 0u;
        if (live) res |= lvl << (4 * (b - 11));
      }
    }
  }
  return res;
}

// q-gram lemma with positions, q = 4: k edits touch at most 4k of the 17 four-base words of the
// pattern's last 20 bases; every untouched word sits in the stream within k positions of where the
// seed (whose right-most piece is in place) expects it.  XOR of the packed pattern with the packed
// stream at the 2k+1 displacements, zero test over 8-bit windows at 2-bit steps, OR, popcount:
// random key matches pass with probability ~1e-6, real ones always (N and EOS pack to arbitrary
// bases, which can only add words).
// The q-gram count on 32-bit registers.  P = the pattern's last 20 bases (2 bits each), tl : th = the 24 stream
// bases p-21 .. p+2.  Words 0..12 live in the low 32 bits of P ^ (T >> c); words 13..16 in bits 24..39, which
// are handled two displacements per register (16 bits each; the zero test never looks across the halves for
// the word offsets used).
__device__ __forceinline__ bool qgram_close(uint32_t plo, uint32_t phi, uint32_t tl, uint32_t th, int k) {
  const uint32_t tm = __builtin_amdgcn_alignbit(th, tl, 24);       // stream bits 24 .. 55
  auto words = [](uint32_t x) __attribute__((always_inline)) -> uint32_t {   // bit 2j: the four bases from j on are equal
    const uint32_t z = x | (x >> 2) | (x >> 4) | (x >> 6);
    return ~(z | (z >> 1));
  };
  uint32_t mlo = 0, mhi = 0;
  uint32_t pend = 0;
  int npend = 0;
#pragma unroll
  for (int d = -2; d <= 2; ++d) {
    if (d < -k || d > k) continue;                                 // wave-uniform
    const int c = 2 * (2 + d);
    mlo |= words(plo ^ (c ? __builtin_amdgcn_alignbit(th, tl, c) : tl));
    const uint32_t y = (phi ^ (tm >> c)) & 0xffffu;
    if (npend == 0) { pend = y; npend = 1; }
    else { mhi |= words(pend | (y << 16)); npend = 0; }
  }
  if (npend) mhi |= words(pend | 0xffff0000u);
  mhi |= mhi >> 16;
  return (int)(__popc(mlo & 0x1555555u) + __popc(mhi & 0x154u)) >= 17 - 4 * k;
}

// The same count with three-base words (18 of them, at most 3k missing).  Twelve clean bases in a row -- the key of
// the combos made of adjacent pieces -- hold 9 four-base words, all that qgram_close asks for at k = 2, but only 10
// of the 12 three-base words needed: for those combos this is the test that rejects chance key matches.
__device__ __forceinline__ bool qgram3_close(uint32_t plo, uint32_t phi, uint32_t tl, uint32_t th, int k) {
  const uint32_t tm = __builtin_amdgcn_alignbit(th, tl, 24);
  auto words = [](uint32_t x) __attribute__((always_inline)) -> uint32_t {   // bit 2j: the three bases from j on are equal
    const uint32_t z = x | (x >> 2) | (x >> 4);
    return ~(z | (z >> 1));
  };
  uint32_t mlo = 0, mhi = 0, pend = 0;
  int npend = 0;
#pragma unroll
  for (int d = -2; d <= 2; ++d) {
    if (d < -k || d > k) continue;                                 // wave-uniform
    const int c = 2 * (2 + d);
    mlo |= words(plo ^ (c ? __builtin_amdgcn_alignbit(th, tl, c) : tl));
    const uint32_t y = (phi ^ (tm >> c)) & 0xffffu;
    if (npend == 0) { pend = y; npend = 1; }
    else { mhi |= words(pend | (y << 16)); npend = 0; }
  }
  if (npend) mhi |= words(pend | 0xffff0000u);
  mhi |= mhi >> 16;
  return (int)(__popc(mlo & 0x5555555u) + __popc(mhi & 0x550u)) >= 18 - 3 * k;   // words 0..13 in the low dword, 14..17 at bits 28..39
}

// qgram_close && qgram3_close in one pass: the pattern XOR the stream at the 2k+1 displacements, the shifted ORs
// behind both word lengths and the funnel of the high part are shared (the two calls one after the other computed
// them twice: ~200 vector instructions per key match, 130 here); accumulated as AND of "some base differs" and
// inverted once.
__device__ __forceinline__ bool qgram_both(uint32_t plo, uint32_t phi, uint32_t tl, uint32_t th, int k) {
  const uint32_t tm = __builtin_amdgcn_alignbit(th, tl, 24);       // stream bits 24 .. 55
  uint32_t n4lo = ~0u, n3lo = ~0u, n4hi = ~0u, n3hi = ~0u;         // bit 2j clear: the four / three bases from j on are equal at some displacement
  uint32_t pend = 0;
  int npend = 0;
  auto hi_pair = [&](uint32_t x) __attribute__((always_inline)) {
    const uint32_t z3 = x | (x >> 2) | (x >> 4), z4 = z3 | (x >> 6);
    n3hi &= z3 | (z3 >> 1); n4hi &= z4 | (z4 >> 1);
  };
#pragma unroll
  for (int d = -2; d <= 2; ++d) {
    if (d < -k || d > k) continue;                                 // wave-uniform
    const int c = 2 * (2 + d);
    const uint32_t x = plo ^ (c ? __builtin_amdgcn_alignbit(th, tl, c) : tl);
    const uint32_t z3 = x | (x >> 2) | (x >> 4), z4 = z3 | (x >> 6);
    n3lo &= z3 | (z3 >> 1); n4lo &= z4 | (z4 >> 1);
    const uint32_t y = (phi ^ (tm >> c)) & 0xffffu;
    if (npend == 0) { pend = y; npend = 1; }
    else { hi_pair(pend | (y << 16)); npend = 0; }
  }
  if (npend) hi_pair(pend | 0xffff0000u);
  const uint32_t m4hi = ~n4hi, m3hi = ~n3hi;
  const int c4 = __popc(~n4lo & 0x1555555u) + __popc((m4hi | (m4hi >> 16)) & 0x154u);
  const int c3 = __popc(~n3lo & 0x5555555u) + __popc((m3hi | (m3hi >> 16)) & 0x550u);
  return c4 >= 17 - 4 * k && c3 >= 18 - 3 * k;
}

__device__ __forceinline__ bool edits_plausible(const SeedArgs &a, int64_t p, uint32_t pi) {
  if (p - 21 < 0 || p + 3 > a.n) return true;                     // stream ends: let the automaton decide
  const uint2 pp = a.pat40[pi];                                   // base j of the last 20 at bits 2j
  uint64_t raw0, raw1, raw2;
  __builtin_memcpy(&raw0, a.text + p - 21, 8);
  __builtin_memcpy(&raw1, a.text + p - 13, 8);
  __builtin_memcpy(&raw2, a.text + p - 5, 8);
  const int sh = a.ascii ? 1 : 0;
  const uint32_t tl = pack4((uint32_t)raw0, sh) | (pack4((uint32_t)(raw0 >> 32), sh) << 8) |
                      (pack4((uint32_t)raw1, sh) << 16) | (pack4((uint32_t)(raw1 >> 32), sh) << 24);
  const uint32_t th = pack4((uint32_t)raw2, sh) | (pack4((uint32_t)(raw2 >> 32), sh) << 8);   // stream p-21 .. p+2
  return qgram_close(pp.x, (((pp.y & 0xffu) << 8) | (pp.x >> 24)), tl, th, a.edits);
}

__device__ __forceinline__ pm_hit edit_record(int64_t end, uint32_t pid, uint32_t lvl1) {
  pm_hit hh;
  hh.end = end; hh.pid = pid; hh.k = (uint8_t)(lvl1 - 1u);
  hh.aux[0] = hh.aux[1] = hh.aux[2] = 0;
  return hh;
}

__device__ __forceinline__ uint64_t edit_seed_record(int64_t p, uint32_t pi) {
  return ((uint64_t)pi << 40) | ((uint64_t)p & 0xffffffffffull);
}

// rare paths (second match in a bucket, probe continuation): one atomic per seed record
__device__ __noinline__ void verify_edits(const SeedArgs *ap, int64_t p, uint32_t pi) {
  const SeedArgs &a = *ap;
  if (!edits_plausible(a, p, pi)) return;
  const unsigned long long o = atomicAdd(a.counter, 1ull);
  if (o < a.cap) a.seed_out[o] = edit_seed_record(p, pi);
}

// packed distance (2 bits per base) never exceeds the true one: a cheap necessary condition
__device__ __forceinline__ bool packed_close(const uint2 &pp, uint64_t W, int k) {
  const uint64_t x = W ^ (((uint64_t)pp.y << 32) | pp.x);
  return (int)__popcll((x | (x >> 1)) & 0x5555555555555555ull) <= k;
}

template <bool HALVES>
__device__ __forceinline__ void verify_pattern(const SeedArgs &a, uint32_t mlo, uint32_t mhi, uint64_t W, int64_t p, uint32_t pi) {
  if (HALVES || packed_close(a.pat40[pi], W, a.k)) verify_hit<HALVES>(a.self, mlo, mhi, p, pi);
}

// slots of a loaded bucket whose fingerprint matches, as a bit mask; bit 8 = the bucket is full
// (the probe sequence continues in the next bucket)
__device__ __forceinline__ uint32_t match_mask(const uint4 &q0, const uint4 &q1, uint32_t fp, uint32_t imask) {
  const uint32_t sl[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
  uint32_t mm = 0;
  bool open = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    open = open || sl[i] == EMPTY;
    if (!open && (sl[i] & ~imask) == fp) mm |= 1u << i;
  }
  return mm | (open ? 0u : 256u);
}

// Second stage.  A 32-byte bucket (two 16-byte loads issued together) holds 8 fingerprinted
// slots; the probe sequence ends at the first bucket with a free slot, which is almost always the
// first one.  check_bucket tests the slots of a bucket that is already in registers.
template <bool HALVES, bool EDITS = false>
__device__ __forceinline__ bool check_bucket(const SeedArgs &a, const uint4 &q0, const uint4 &q1, uint32_t fp, uint32_t imask,
                                             uint32_t mlo, uint32_t mhi, uint64_t W, int64_t p) {
  const uint32_t sl[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (sl[i] == EMPTY) return true;
    if ((sl[i] & ~imask) == fp) {
      if (EDITS) verify_edits(a.self, p, sl[i] & imask);
      else verify_pattern<HALVES>(a, mlo, mhi, W, p, sl[i] & imask);
    }
  }
  return false;
}

// continue a probe sequence from bucket b (rare: only after a full bucket)
template <bool HALVES, bool EDITS = false>
__device__ __noinline__ void probe_from(const SeedArgs *ap, const uint4 *buckets, uint32_t b, uint32_t fp, uint32_t imask,
                                        uint32_t mlo, uint32_t mhi, uint64_t W, int64_t p) {
  const SeedArgs &a = *ap;
  const uint32_t bmask = (1u << (32 - a.bucket_shift)) - 1u;
  for (;;) {
    b &= bmask;
    const uint4 q0 = buckets[2 * (size_t)b], q1 = buckets[2 * (size_t)b + 1];
    if (check_bucket<HALVES, EDITS>(a, q0, q1, fp, imask, mlo, mhi, W, p)) break;
    ++b;
  }
}


// LW > 0: window length known at compile time (all shifts immediate); LW == 0: taken from a.Lw.
// MODE: see window_hash.
template <int LW, int MODE, bool HALVES, bool EDITS = false>
__global__ __launch_bounds__(SEED_THREADS) void pm_seed_scan(SeedArgs a) {
  extern __shared__ uint32_t lds[];
  uint32_t *bloom = lds;                                          // SEED_BLOOM_STRIDE dwords, at LDS address 0 (bloom_block)
  uint2 *queue_all = reinterpret_cast<uint2 *>(lds + SEED_BLOOM_STRIDE);
  // the filter must sit at LDS address 0 (bloom_block): these kernels have no static LDS, which
  // seed_upload checks on the host (hipFuncGetAttributes) before anything is launched

  // blockIdx -> (combo, chunk): runs of `group` chunks share a combo, all combos of a superchunk
  // follow each other, so the stream bytes of a superchunk are re-read from MALL and the combo's
  // bucket table stays in every XCD's L2 while a run is in flight.
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
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.bloom + (size_t)combo * SEED_BLOOM_STRIDE);
    u32x4 *dst = reinterpret_cast<u32x4 *>(bloom);
    for (int i = threadIdx.x; i < SEED_BLOOM_STRIDE / 4; i += SEED_THREADS) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint2 *queue = queue_all + wave * QCAP;
  const int64_t sub = a.chunk_len / WAVES;
  const int64_t ws = (a.chunk0 + cj) * a.chunk_len + (int64_t)wave * sub;   // first base of this wave
  // p = index of the window's last base.  EDITS: a candidate end e comes from seeds at p = e-1+-k, so the
  // positions k beyond either side of (begin, end] are scanned too; records are filtered by e.
  const int64_t p_lo = EDITS ? a.begin - a.edits : a.begin, p_hi = EDITS ? a.end + a.edits : a.end;
  int64_t own_lo = ws > p_lo ? ws : p_lo;
  int64_t own_hi = ws + sub;
  if (own_hi > p_hi) own_hi = p_hi;
  if (own_hi > a.n) own_hi = a.n;
  const int Lw = LW > 0 ? LW : a.Lw;
  if (own_lo < Lw - 1) own_lo = Lw - 1;                           // the window must fit in the stream
  if (own_lo >= own_hi) return;

  const uint32_t mlo = a.mask_lo[combo], mhi = a.mask_hi[combo], sel = a.perm_sel[combo];
  const uint4 *buckets = a.buckets + (size_t)combo * 2 * ((size_t)1 << (32 - a.bucket_shift));
  const uint32_t *bitmap2 = a.bitmap2 + (size_t)combo * ((size_t)1 << (a.lb2 - 5));
  // the 32 bases in front of the wave's range
  uint32_t carry1, carry2;
  {
    const uint32_t pk = load_packed<MODE == 2>(a.packed, a.npacked, ws - 32 + 16 * (lane & 1));
    carry2 = __builtin_amdgcn_readlane(pk, 0);
    carry1 = __builtin_amdgcn_readlane(pk, 1);
  }
  const int wbits = 2 * Lw;
  const uint32_t lo_mask = wbits >= 32 ? 0xffffffffu : ((1u << wbits) - 1u);
  const uint32_t hi_mask = wbits > 32 ? ((1u << (wbits - 32)) - 1u) : 0u;
  int qn = 0;                                                     // wave-uniform queue fill

  // window whose last base is base i of this lane's 16: bits [s, s+2Lw) of prev2:prev1:cur
  auto window = [&](int i, uint32_t prev2, uint32_t prev1, uint32_t cur, uint32_t &wlo, uint32_t &whi) {
    const int s = 2 * (i - Lw + 33);
    if (s < 32) { wlo = __builtin_amdgcn_alignbit(prev1, prev2, s); whi = __builtin_amdgcn_alignbit(cur, prev1, s); }
    else if (s < 64) { wlo = __builtin_amdgcn_alignbit(cur, prev1, s - 32); whi = cur >> (s - 32); }
    else { wlo = cur >> (s - 64); whi = 0; }
    if (MODE == 0 || LW == 0) { wlo &= lo_mask; whi &= hi_mask; }
  };
  const uint32_t imask = (1u << a.idx_bits) - 1u;
  // Second stage, two queues per wave.
  //   Q1 <- Bloom survivors (~11 % of the tests).  When it fills up, every lane takes up to two
  //   of its windows into registers and issues their loads from the small second-level bitmap
  //   (stage 2a); the wave goes back to the first stage and only looks at the answers when Q1 is
  //   full again, so this round trip is never waited for.
  //   Q2 <- the ~5 % of Q1 that the bitmap lets through.  Only when 64 of them have gathered does
  //   the wave load their buckets and packed patterns (stages 2b/3): one dense pass instead of a
  //   mostly idle one per Q1 batch.
  uint2 *queue2 = reinterpret_cast<uint2 *>(lds + SEED_BLOOM_STRIDE) + WAVES * QCAP + wave * SEED_Q2CAP;
  int q2n = 0;
  // HALVES: records go to slots reserved SEED_OUT_BLOCK at a time (wave-uniform state)
  unsigned long long ob_next = 0;
  int ob_left = 0;
  auto emit_half = [&](bool pass, int64_t p, uint32_t pid) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(pass);
    if (bal == 0) return;
    const int c = __popcll(bal);
    if (c > ob_left) {
      if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
      ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)base);
      ob_left = SEED_OUT_BLOCK;
    }
    if (pass) {
      const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (slot < a.cap) a.out[slot] = half_seed_record(p, pid);
    }
    ob_next += c; ob_left -= c;
  };
  auto emit_edit = [&](bool pass, int64_t p, uint32_t pi) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(pass);
    if (bal == 0) return;
    const int c = __popcll(bal);
    if (c > ob_left) {
      if (lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
      ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)base);
      ob_left = SEED_OUT_BLOCK;
    }
    if (pass) {
      const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (slot < a.cap) a.seed_out[slot] = edit_seed_record(p, pi);
    }
    ob_next += c; ob_left -= c;
  };
  auto process_q2 = [&]() __attribute__((always_inline)) {
    if (a.debug & 2) { q2n = 0; return; }
    for (int base = 0; base < q2n; base += 64) {
      const bool on = base + lane < q2n;
      uint32_t wlo = 0, whi = 0, h2 = 0, mm = 0, pidx = 0;
      int64_t p = 0;
      uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
      uint2 pp = make_uint2(0, 0);
      uint64_t th0 = 0, th1 = 0;
      if (on) {
        const uint2 e = queue2[base + lane];
        wlo = e.x & lo_mask; whi = e.y & 0xffu & hi_mask; p = ws + (e.y >> 8);
        h2 = window_hash<MODE>(e.x, e.y & 0xffu, mlo, mhi, sel) * HASH_SLOT;
        const size_t b = h2 >> a.bucket_shift;
        b0 = buckets[2 * b]; b1 = buckets[2 * b + 1];
        if (HALVES) load_tail16(a, p, th0, th1);                   // independent of the bucket: same round trip
        mm = match_mask(b0, b1, h2 << a.idx_bits, imask);
      }
      const uint32_t sl[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      if (mm & 255u) {
        const int sidx = __ffs(mm) - 1;
        uint32_t slot = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) slot = sidx == t ? sl[t] : slot;
        pidx = slot & imask;
        if (!HALVES && !EDITS) pp = a.pat40[pidx];                   // halves / edits: their own record decides
      }
      if (EDITS) {                                                 // wave-uniform: three-base-word test, seeds out in reserved blocks
        bool pass = false;
        if ((mm & 255u) && !(a.debug & 4)) pass = edits_plausible(a, p, pidx);
        emit_edit(pass, p, pidx);
      }
      if (HALVES) {                                                // wave-uniform: block-reserved output
        uint32_t pid = 0;
        bool pass = false;
        if ((mm & 255u) && !(a.debug & 4)) {
          if (a.hfast) {
            // record and partner window in one round trip (side = parity of the half's index)
            const uint4 *rec = reinterpret_cast<const uint4 *>(a.pat_codes + (size_t)pidx * 32);
            const int64_t r0 = partner_window(p, a.hfast, pidx & 1, a.hk);
            const bool inside = r0 >= 0 && r0 + 24 <= a.n;
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            const uint4 c = rec[0], r = rec[1];
            if (inside) {
              __builtin_memcpy(&w0, a.text + r0, 8);
              __builtin_memcpy(&w1, a.text + r0 + 8, 8);
              __builtin_memcpy(&w2, a.text + r0 + 16, 8);
            }
            const int L = r.y & 0xffu, plen = (r.y >> 8) & 0xffu;
            pid = r.z;
            pass = p + 1 - L >= 0 && half_codes_equal(c, L, th0, th1) &&
                   (!inside || partner_possible(a, plen, r.x, w0, w1, w2));
          } else {
            pass = half_seed_ok(a, p, pidx, &pid);
          }
        }
        emit_half(pass, p, pid);
      }
      if ((mm & 511u) && !(a.debug & 4)) {
        const uint64_t W = ((uint64_t)whi << 32) | wlo;
        if (mm & 255u) {
          if (!HALVES && !EDITS && packed_close(pp, W, a.k)) verify_hit<HALVES>(a.self, mlo, mhi, p, pidx);
          uint32_t rest = (mm & 255u) & ((mm & 255u) - 1u);         // matches beyond the first (rare)
          while (rest) {
            const int sidx = __ffs(rest) - 1;
            rest &= rest - 1;
            uint32_t slot = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) slot = sidx == t ? sl[t] : slot;
            if (EDITS) verify_edits(a.self, p, slot & imask);
            else verify_pattern<HALVES>(a, mlo, mhi, W, p, slot & imask);
          }
        }
        if (mm & 256u) probe_from<HALVES, EDITS>(a.self, buckets, (h2 >> a.bucket_shift) + 1, h2 << a.idx_bits, imask, mlo, mhi, W, p);
      }
    }
    q2n = 0;
  };
  // (scalars, not arrays: the state must stay in VGPRs across the block loop)
  uint32_t pw0 = 0, pw1 = 0, px0 = 0, px1 = 0, pb0 = 0, pb1 = 0, pt0 = 0, pt1 = 0;
  int pend = 0;                                                   // wave-uniform: windows in flight
  auto finish_one = [&](int j, uint32_t pw, uint32_t px, uint32_t pb, uint32_t pt) __attribute__((always_inline)) {
    const bool pass = (64 * j + lane < pend) && ((pb >> (pt & 31)) & 1u);
    const unsigned long long bal = __ballot(pass);
    if (bal == 0) return;
    if (q2n + 64 > SEED_Q2CAP) process_q2();
    if (pass) queue2[q2n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0))] = make_uint2(pw, px);
    q2n += __popcll(bal);
  };
  auto finish = [&]() __attribute__((always_inline)) {
    if (pend == 0) return;
    finish_one(0, pw0, px0, pb0, pt0);
    if (pend > 64) finish_one(1, pw1, px1, pb1, pt1);
    pend = 0;
  };
  auto issue_one = [&](int j, uint32_t &pw, uint32_t &px, uint32_t &pb, uint32_t &pt) __attribute__((always_inline)) {
    const int q = 64 * j + lane;
    if (q < qn) {
      const uint2 e = queue[q];
      pw = e.x; px = e.y;
      if (HALVES && a.exact_filter) {                              // nothing false came through the key bitmap
        pt = 0; pb = 1u;
      } else {
        const uint32_t h2 = window_hash<MODE>(e.x, e.y & 0xffu, mlo, mhi, sel) * HASH_SLOT;
        pt = h2 >> (32 - a.lb2);                                     // bit index inside the bitmap
        pb = bitmap2[h2 >> (37 - a.lb2)];
      }
    }
  };
  auto drain = [&]() __attribute__((always_inline)) {             // qn <= QCAP = 128 = 2 per lane
    if (a.debug & 1) { qn = 0; return; }
    finish();
    issue_one(0, pw0, px0, pb0, pt0);
    issue_one(1, pw1, px1, pb1, pt1);
    pend = qn;
    qn = 0;
  };

  // four blocks of 1024 bases (one packed dword per lane each) in flight per wave: rolling prefetch ring
  uint32_t q0 = load_packed<MODE == 2>(a.packed, a.npacked, ws + 16 * lane);
  uint32_t q1 = ws + 1024 < own_hi ? load_packed<MODE == 2>(a.packed, a.npacked, ws + 1024 + 16 * lane) : 0u;
  uint32_t q2 = ws + 2048 < own_hi ? load_packed<MODE == 2>(a.packed, a.npacked, ws + 2048 + 16 * lane) : 0u;
  uint32_t q3 = ws + 3072 < own_hi ? load_packed<MODE == 2>(a.packed, a.npacked, ws + 3072 + 16 * lane) : 0u;
  for (int64_t bb = ws; bb < own_hi; bb += 1024) {
    const uint32_t cur = q0;
    q0 = q1; q1 = q2; q2 = q3;
    if (bb + 4096 < own_hi) q3 = load_packed<MODE == 2>(a.packed, a.npacked, bb + 4096 + 16 * lane);
    uint32_t prev1 = __shfl_up(cur, 1), prev2 = __shfl_up(cur, 2);
    if (lane == 0) { prev1 = carry1; prev2 = carry2; }
    if (lane == 1) prev2 = carry1;
    carry2 = __builtin_amdgcn_readlane(cur, 62);
    carry1 = __builtin_amdgcn_readlane(cur, 63);
    const int64_t pbase = bb + 16 * lane;
    // positions of this lane that this wave owns (all 16 except in the first/last block)
    uint32_t own = 0xffffu;
    if (bb < own_lo || bb + 1024 > own_hi) {                       // wave-uniform: edge blocks only
      const int64_t lo = own_lo - pbase, hi = own_hi - pbase;
      const uint32_t l = lo <= 0 ? 0u : (lo >= 16 ? 16u : (uint32_t)lo), hh = hi <= 0 ? 0u : (hi >= 16 ? 16u : (uint32_t)hi);
      own = ((1u << hh) - 1u) & ~((1u << l) - 1u);
    }
    // 32 window bits at base offset d from window i's first base (EDITS: displaced pieces; they only
    // ever come from the low word, the last piece is never displaced)
    auto wlo_at = [&](int i, int d) __attribute__((always_inline)) -> uint32_t {
      const int s = 2 * (i - Lw + 33) + 2 * d;                     // 22 .. 60
      return s < 32 ? __builtin_amdgcn_alignbit(prev1, prev2, s) : __builtin_amdgcn_alignbit(cur, prev1, s - 32);
    };
    const bool exact = HALVES && a.exact_filter != 0;
    const uint32_t ema = EDITS ? a.emask_a[combo] : 0u, emb = EDITS ? a.emask_b[combo] : 0u, evar = EDITS ? a.evar[combo] : 0u;
    // first stage, parts 1 and 2 for one displacement pattern (sa, sb = displacement of the combo's
    // first and second piece; compile-time constants at every call): 16 hashes, 16 LDS reads in
    // flight, three bit tests per window, verdicts funnelled into one register
    constexpr int TG = PM_TESTS_GROUP;
    auto tests = [&](int sa, int sb) __attribute__((always_inline)) -> uint32_t {
      uint32_t acc = 0;
#pragma unroll
      for (int half = 0; half < 16 / TG; ++half) {                   // TG reads in flight, then their TG verdicts
        uint32_t hs[TG], wd[TG];
#pragma unroll
        for (int j = 0; j < TG; ++j) {
          const int i = TG * half + j;
          uint32_t wlo, whi;
          window(i, prev2, prev1, cur, wlo, whi);
          if (EDITS && (sa != 0 || sb != 0)) {
            const uint32_t wb = sb ? wlo_at(i, sb) : wlo, wa = sa ? wlo_at(i, sa) : wlo;
            wlo = (wa & ema) | (~ema & ((wb & emb) | (~emb & wlo)));
          }
          if (HALVES && MODE == 0 && exact) {                       // wave-uniform: bit `key` of the key bitmap
            const uint32_t key = wlo & mlo;
            hs[j] = key;
            wd[j] = *reinterpret_cast<lds_u32 *>((uintptr_t)((key >> 3) & ~3u));
          } else {
            uint32_t ss;
            const uint32_t h = window_hash<MODE>(wlo, whi, mlo, mhi, sel, &ss);
            hs[j] = bloom_selectors(ss);
            wd[j] = bloom_block(h);
          }
        }
        __builtin_amdgcn_sched_barrier(0);                          // without it the scheduler waits for every read by itself
        if (HALVES && MODE == 0 && exact) {
#pragma unroll
          for (int j = 0; j < TG; ++j) acc = __builtin_amdgcn_alignbit(wd[j] >> (hs[j] & 31u), acc, 1);
        } else {
#pragma unroll
          for (int j = 0; j < TG; ++j) acc = __builtin_amdgcn_alignbit(bloom_test(wd[j], hs[j]), acc, 1);
        }
      }
      return (acc >> 16) & own;
    };
    // EDITS: the displacement patterns (d1 between third and second piece, d2 between second and
    // first), |d1|+|d2| <= k; survivors of all of them first, then one compaction loop
    constexpr int NV = EDITS ? 13 : 1;
    constexpr int VD1[13] = {0, 0, 0, 1, -1, 0, 0, 2, -2, 1, 1, -1, -1};
    constexpr int VD2[13] = {0, 1, -1, 0, 0, 2, -2, 0, 0, 1, -1, 1, -1};
    uint32_t rems2[(NV + 1) / 2];                                  // two 16-bit survivor masks per register
#pragma unroll
    for (int v = 0; v < (NV + 1) / 2; ++v) rems2[v] = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int cost = (VD1[v] < 0 ? -VD1[v] : VD1[v]) + (VD2[v] < 0 ? -VD2[v] : VD2[v]);
      // which (combo, displacement) pairs are needed is decided on the host (edit_cover): a set
      // cover over all placements of <= k edits, 34 of the 130 pairs for k = 2
      if (!EDITS || (cost <= a.edits && ((evar >> v) & 1u))) rems2[v >> 1] |= tests(VD1[v] + VD2[v], VD1[v]) << (16 * (v & 1));
    }
    // part 3: compaction, one survivor per lane and round (ballot + mbcnt give the queue slots)
#pragma unroll 1
    for (int v = 0; v < NV; ++v) {
      uint32_t rem = rems2[0] & 0xffffu;
      int sa = 0, sb = 0;
      if (EDITS) {
#pragma unroll
        for (int t = 1; t < NV; ++t) if (v == t) { rem = (rems2[t >> 1] >> (16 * (t & 1))) & 0xffffu; sa = VD1[t] + VD2[t]; sb = VD1[t]; }
      }
      for (;;) {
        const unsigned long long bal = __ballot(rem != 0);
        if (bal == 0) break;
        if (qn + 64 > QCAP) drain();
        if (rem != 0) {
          const int i = __ffs(rem) - 1;
          __builtin_assume(i >= 0 && i < 16);
          rem &= rem - 1;
          const int sft = 2 * (i - Lw + 33);                       // per-lane bit offset of the window
          const uint32_t x0 = __builtin_amdgcn_alignbit(prev1, prev2, sft), x1 = __builtin_amdgcn_alignbit(cur, prev1, sft),
                         x2 = cur >> (sft & 31);
          uint32_t wlo = sft < 32 ? x0 : (sft < 64 ? x1 : x2);
          const uint32_t whi = sft < 32 ? x1 : (sft < 64 ? x2 : 0u);
          if (EDITS && (sa | sb)) {                                // wave-uniform
            const int fa = sft + 2 * sa, fb = sft + 2 * sb;
            const uint32_t wa = fa < 32 ? __builtin_amdgcn_alignbit(prev1, prev2, fa) : __builtin_amdgcn_alignbit(cur, prev1, fa);
            const uint32_t wb = fb < 32 ? __builtin_amdgcn_alignbit(prev1, prev2, fb) : __builtin_amdgcn_alignbit(cur, prev1, fb);
            wlo = (wa & ema) | (~ema & ((wb & emb) | (~emb & wlo)));
          }
          const int slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
          queue[slot] = make_uint2(wlo, (whi & 0xffu) | ((uint32_t)(pbase + i - ws) << 8));
        }
        qn += __popcll(bal);
      }
    }
  }
  drain();
  finish();
  process_q2();
  if (HALVES && lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
  if (EDITS && lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
}

// ---- edit-distance plan, first stage: pm_edit_scan ------------------------------------------------
//
// Same seeds as the EDITS instance of pm_seed_scan -- three clean 4-base pieces of the seeded window
// under one of the displacement patterns edit_cover chose -- with a first stage built around two
// facts: every piece is one BYTE of the 2-bit packed stream at some bit offset, and the (combo,
// displacement) pairs of one combo only differ in WHICH bytes they combine.
//   * The key hash is tabulated per piece: H = F0(first piece) ^ F1(second) ^ F2(third), F = 24-bit
//     multiply + fold of one byte (edit_piece_hash).  A lane computes the F values of the bytes its
//     eight windows can use once per half block (two SDWA instructions per value: the byte select rides
//     on the multiply) and every test is then one three-way XOR (v_bitop3) of registers picked at
//     compile time -- no funnel shifts, no piece merging, no multiply per test.
//   * The combo's filter in LDS is a blocked Bloom filter addressed by H itself: dword at byte
//     H & 0x1fffc, bits (H >> 16) & 31, (H >> 24) & 31 and five bits of H >> et_shift (three SDWA
//     shifts; et_shift = 12 for the default map size).
//   * Survivors (12 % at 200k patterns) look their key up in a 2^23-bit map in L2 (1 MiB per combo:
//     dword H >> 14, bit (H >> 8) & 31 -- 2.4 % of the false survivors get through at 200k keys): a
//     branch-free dword load per test -- non-survivors read dword 0, one cached line -- consumed one
//     unit later with one SDWA shift.  That replaces the compaction of every Bloom survivor into an
//     LDS queue and the second-level bitmap: only the ~1.3 % of the tests that are real 12-base key
//     matches are compacted (ballot + mbcnt), probed in the bucket table and put to the word counts.
// One instance per combo (pieces and displacement list are template parameters: all register indices
// and stream offsets are immediates); the lists are edit_cover's, checked on the host at upload.
constexpr uint32_t EDIT_MUL0 = 0x9E3779u, EDIT_MUL1 = 0xC2B2AFu, EDIT_MUL2 = 0x85EBCBu;
#ifndef PM_EDIT_BUCKET
#define PM_EDIT_BUCKET 16
#endif
constexpr int EDIT_BUCKET = PM_EDIT_BUCKET;                       // slots per bucket of the second stage (16: 64 bytes, half a cache line)
constexpr int EDIT_QCAP = (SEED_QCAP + SEED_Q2CAP) * 2 / 3;       // suspicious windows per wave (12-byte entries)
__device__ __host__ __forceinline__ uint32_t edit_piece_hash(uint32_t g, uint32_t mul) {
  const uint32_t y = g * mul;                                      // g < 256, mul < 2^24: no overflow
  return y ^ (y >> 16);
}
constexpr int EVD1[13] = {0, 0, 0, 1, -1, 0, 0, 2, -2, 1, 1, -1, -1};   // displacement second - third piece
constexpr int EVD2[13] = {0, 1, -1, 0, 0, 2, -2, 0, 0, 1, -1, 1, -1};   // first - second piece
struct EditVariants {
  int n;
  int sa[13], sb[13];                                              // displacement of the first and second piece, variants sorted by sb
  int lo_a, hi_a, lo_b, hi_b;
};
constexpr EditVariants edit_variants(uint32_t evar) {
  EditVariants L{};
  L.n = 0; L.lo_a = 0; L.hi_a = 0; L.lo_b = 0; L.hi_b = 0;
  for (int i = 0; i < 13; ++i) { L.sa[i] = 0; L.sb[i] = 0; }
  for (int s = -2; s <= 2; ++s)
    for (int v = 0; v < 13; ++v)
      if (((evar >> v) & 1u) && EVD1[v] == s) {
        const int sa = EVD1[v] + EVD2[v];
        L.sa[L.n] = sa; L.sb[L.n] = s; ++L.n;
        if (sa < L.lo_a) L.lo_a = sa;
        if (sa > L.hi_a) L.hi_a = sa;
        if (s < L.lo_b) L.lo_b = s;
        if (s > L.hi_b) L.hi_b = s;
      }
  return L;
}
// the displacement lists edit_cover finds for the combos of the k = 1 and k = 2 plans (bit v = pattern v)
constexpr uint32_t edit_cover_of(int qa, int qb, int qc) {
  const int c = qa * 100 + qb * 10 + qc;
  return c == 12 ? 0x1u : c == 13 ? 0x19u : c == 14 ? 0x199u : c == 23 ? 0x7u : c == 24 ? 0x1e1fu : c == 34 ? 0x67u :
         c == 123 ? 0x1u : c == 124 ? 0x19u : c == 134 ? 0x7u : c == 234 ? 0x1u : 0u;
}

template <int N, class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// 32 bits from bit O (compile time) of the 96-bit string p2 : p1 : cur (bit 0 = bit 0 of p2)
template <int O>
__device__ __forceinline__ uint32_t bits_at(uint32_t p2, uint32_t p1, uint32_t cur) {
  static_assert(O >= 0 && O < 96, "offset");
  if constexpr (O == 0) return p2;
  else if constexpr (O < 32) return __builtin_amdgcn_alignbit(p1, p2, O);
  else if constexpr (O == 32) return p1;
  else if constexpr (O < 64) return __builtin_amdgcn_alignbit(cur, p1, O - 32);
  else if constexpr (O == 64) return cur;
  else return cur >> (O - 64);
}
// F of the byte at bit B of p2 : p1 : cur: the byte is picked by the multiply's operand selector
template <int B>
__device__ __forceinline__ uint32_t piece_hash_at(uint32_t p2, uint32_t p1, uint32_t cur, uint32_t mul) {
  constexpr int PH = B & 7, R = (B - PH) / 8, O = PH + 32 * (R / 4), T = R % 4;
  const uint32_t w = bits_at<O>(p2, p1, cur);
  uint32_t y, f;
  if constexpr (T == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(y) : "v"(w), "v"(mul));
  else if constexpr (T == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(y) : "v"(w), "v"(mul));
  else if constexpr (T == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(y) : "v"(w), "v"(mul));
  else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(y) : "v"(w), "v"(mul));
  asm("v_xor_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(f) : "v"(y));
  return f;
}
typedef __attribute__((address_space(3))) uint32_t lds_w32;

// Second stage of pm_edit_scan.  A suspicious window is a 12-base key match with one of the patterns
// (1.2 % of the tests at 200k patterns), so this path is kept to ONE cache line per window: the queue
// entry carries the 24 stream bases around the window (no stream re-read), the bucket slot carries the
// pattern's other two pieces next to a 16-bit fingerprint (the key pieces are the window's own: no
// pattern load), and the q-gram test runs on registers.  The pattern index sits in a parallel table
// that only the windows that pass read.  Bucket = 16 slots of (other pieces << 16 | fingerprint), 64 bytes.
__device__ __forceinline__ uint32_t edit_fp16(uint32_t h2) { const uint32_t f = h2 & 0xffffu; return f == 0xffffu ? 0xfffeu : f; }
// pattern bytes (pieces 0..3 in plo, piece 4 in bits 0..7 of p4) from the merged window and a slot's other pieces
__device__ __forceinline__ void edit_pattern_of(uint32_t wlo, uint32_t whi, uint32_t other, uint32_t sel, uint32_t *plo, uint32_t *p4) {
  const int qa = sel & 0xff, qb = (sel >> 8) & 0xff, qc = (sel >> 16) & 0xff;
  uint32_t pl = 0, ph = 0;
  int t = 0;
  for (int q = 0; q < 5; ++q) {
    const bool key = q == qa || q == qb || q == qc;
    const uint32_t by = key ? (q < 4 ? (wlo >> (8 * q)) & 0xffu : whi & 0xffu) : (other >> (8 * t)) & 0xffu;
    if (!key) ++t;
    if (q < 4) pl |= by << (8 * q); else ph = by;
  }
  *plo = pl; *p4 = ph;
}
// rare (1e-3 of the suspicious windows): a full bucket -- the probe sequence goes on in the next one
__device__ __noinline__ void edit_rare(const SeedArgs *ap, int combo, uint32_t h2, uint32_t wlo, uint32_t whi, uint32_t tl, uint32_t th, int64_t p) {
  const SeedArgs &a = *ap;
  const size_t nb = (size_t)1 << (32 - a.bucket_shift);
  const uint32_t *slots = reinterpret_cast<const uint32_t *>(a.buckets) + (size_t)combo * nb * EDIT_BUCKET;
  const uint32_t fp = edit_fp16(h2);
  uint32_t b = (h2 >> a.bucket_shift) + 1;
  for (;;) {
    b &= (uint32_t)(nb - 1);
    for (int q = 0; q < EDIT_BUCKET; ++q) {
      const uint32_t sv = slots[(size_t)b * EDIT_BUCKET + q];
      if (sv == EMPTY) return;
      if ((sv & 0xffffu) != fp) continue;
      uint32_t plo, p4;
      edit_pattern_of(wlo, whi, sv >> 16, a.perm_sel[combo], &plo, &p4);
      const bool ends = p - 21 < 0 || p + 3 > a.n;
      if (ends || (qgram_close(plo, (p4 << 8) | (plo >> 24), tl, th, a.edits) && qgram3_close(plo, (p4 << 8) | (plo >> 24), tl, th, a.edits))) {
        const unsigned long long o = atomicAdd(a.counter, 1ull);
        if (o < a.cap) a.seed_out[o] = edit_seed_record(p, ((uint32_t)combo << 20) | (uint32_t)((size_t)b * EDIT_BUCKET + q));
      }
    }
    ++b;
  }
}

template <int QA, int QB, int QC, uint32_t EVAR>
__device__ __forceinline__ void edit_scan_body(const SeedArgs &a, const int combo, const int cj, uint32_t *lds) {
  constexpr EditVariants VL = edit_variants(EVAR);
  constexpr int NV = VL.n, LOA = VL.lo_a, HIA = VL.hi_a, LOB = VL.lo_b, HIB = VL.hi_b;
  constexpr int NA = 8 + HIA - LOA, NB = 8 + HIB - LOB;
  static_assert(NV >= 1 && NV <= 10 && QA < QB && QB < QC && QC <= 4 && QB <= 3, "combo");
  // the two pieces that are not in the key, and the byte selector that puts a slot's copy of them between the key pieces
  constexpr int O1 = QA != 0 ? 0 : (QB != 1 ? 1 : (QC != 2 ? 2 : 3));
  constexpr int O2 = QC != 4 ? 4 : (QB != 3 ? 3 : (QA != 2 ? 2 : 1));
  static_assert(O1 < O2 && O1 != QA && O1 != QB && O1 != QC && O2 != QA && O2 != QB && O2 != QC, "other pieces");
  constexpr uint32_t PSEL = (uint32_t)(0 == O1 ? 4 : (0 == O2 ? 5 : 0)) | ((uint32_t)(1 == O1 ? 4 : (1 == O2 ? 5 : 1)) << 8) |
                            ((uint32_t)(2 == O1 ? 4 : (2 == O2 ? 5 : 2)) << 16) | ((uint32_t)(3 == O1 ? 4 : (3 == O2 ? 5 : 3)) << 24);
  // per displacement pattern: stream bit offsets of the first and second piece's source window, 2 (s + 2), three bits each
  constexpr uint64_t VCODE = [] { uint64_t c = 0; for (int v = 0; v < VL.n; ++v) c |= (uint64_t)((VL.sa[v] + 2) | ((VL.sb[v] + 2) << 3)) << (6 * v); return c; }();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the wave's queue of suspicious windows: {stream bases p-21 .. p-6, bases p-5 .. p+2, position | pattern << 20}
  uint32_t *q_tl = lds + SEED_BLOOM_STRIDE + wave * (3 * EDIT_QCAP), *q_th = q_tl + EDIT_QCAP, *q_pv = q_th + EDIT_QCAP;
  const int64_t sub = a.chunk_len / WAVES;
  const int64_t ws = (a.chunk0 + cj) * a.chunk_len + (int64_t)wave * sub;
  // p = last base of the 20-base window; a candidate end e comes from seeds at p = e-1+-k
  const int64_t p_lo = a.begin - a.edits, p_hi = a.end + a.edits;
  int64_t own_lo = ws > p_lo ? ws : p_lo;
  int64_t own_hi = ws + sub;
  if (own_hi > p_hi) own_hi = p_hi;
  if (own_hi > a.n) own_hi = a.n;
  if (own_lo < 19) own_lo = 19;
  if (own_lo >= own_hi) return;

  const uint32_t sel = a.perm_sel[combo];
  const uint32_t ema = a.emask_a[combo], emb = a.emask_b[combo];
  const size_t nbs = (size_t)EDIT_BUCKET << (32 - a.bucket_shift);
  const uint4 *buckets = a.buckets + (size_t)combo * (nbs / 4);
  const uint8_t *etable = a.etable + ((size_t)combo << a.et_bytes_log);
  const int jshift = a.et_shift;
  uint32_t carry1, carry2;
  {
    const uint32_t pk = load_packed<false>(a.packed, a.npacked, ws - 32 + 16 * (lane & 1));
    carry2 = __builtin_amdgcn_readlane(pk, 0);
    carry1 = __builtin_amdgcn_readlane(pk, 1);
  }
  int qn = 0;                                                     // wave-uniform queue fill
  unsigned long long ob_next = 0;                                 // seed records go to slots reserved SEED_OUT_BLOCK at a time
  int ob_left = 0;
  auto emit = [&](bool pass, int64_t p, uint32_t slot_at) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(pass);
    if (bal == 0) return;
    const int c = __popcll(bal);
    if (c > ob_left) {
      if (lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
      ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)base);
      ob_left = SEED_OUT_BLOCK;
    }
    if (pass) {
      const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (slot < a.cap) a.seed_out[slot] = edit_seed_record(p, ((uint32_t)combo << 20) | slot_at);   // (combo, slot): pm_edits_verify looks the pattern up
    }
    ob_next += c; ob_left -= c;
  };
  // bucket probe + q-gram test of whole batches of 64 queued windows (all: of what is left, too)
  auto process = [&](bool all) __attribute__((always_inline)) {
    while (qn >= 64 || (all && qn > 0)) {
      const int cnt = qn >= 64 ? 64 : qn;
      qn -= cnt;
      if (a.debug & 2) continue;
      const bool on = lane < cnt;
      uint32_t h2 = 0, tl = 0, th = 0, wlo = 0, whi = 0, slot_at = 0, c6 = 0, wraw = 0;
      int64_t p = 0;
      uint4 bq[EDIT_BUCKET / 4];
#pragma unroll
      for (int t = 0; t < EDIT_BUCKET / 4; ++t) bq[t] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
      if (on) {
        tl = q_tl[qn + lane]; th = q_th[qn + lane];
        const uint32_t pv = q_pv[qn + lane];
        p = ws + (pv & 0xfffffu);
        // the window (stream bases p-19 .. p) with the key's first and second piece taken from where the pattern displaces them
        c6 = (uint32_t)(VCODE >> (6 * (pv >> 20))) & 63u;
        wlo = __builtin_amdgcn_alignbit(th, tl, 4); whi = (th >> 4) & 0xffu;
        wraw = wlo;                                                  // the window as it stands in the stream
        const uint32_t wa = __builtin_amdgcn_alignbit(th, tl, 2 * (c6 & 7u)), wb = __builtin_amdgcn_alignbit(th, tl, 2 * (c6 >> 3));
        wlo = (wa & ema) | (~ema & ((wb & emb) | (~emb & wlo)));
        h2 = window_hash<1>(wlo, whi, 0, 0, sel) * HASH_SLOT;
        const size_t b = h2 >> a.bucket_shift;
#pragma unroll
        for (int t = 0; t < EDIT_BUCKET / 4; ++t) bq[t] = buckets[(EDIT_BUCKET / 4) * b + t];
        slot_at = (uint32_t)b * EDIT_BUCKET;
      }
      uint32_t sl[EDIT_BUCKET];
#pragma unroll
      for (int t = 0; t < EDIT_BUCKET / 4; ++t) { sl[4 * t] = bq[t].x; sl[4 * t + 1] = bq[t].y; sl[4 * t + 2] = bq[t].z; sl[4 * t + 3] = bq[t].w; }
      // slots fill up in order and an empty slot's low half (0xffff) is no fingerprint: no "first empty slot" bookkeeping
      const uint32_t fp = edit_fp16(h2);
      uint32_t mm = 0;                                               // slots whose fingerprint matches
#pragma unroll
      for (int t = 0; t < EDIT_BUCKET; ++t) mm |= (sl[t] & 0xffffu) == fp ? 1u << t : 0u;
      const bool full = on && sl[EDIT_BUCKET - 1] != EMPTY && !(a.debug & 4);
      if (!on || (a.debug & 4)) mm = 0;
      const bool ends = p - 21 < 0 || p + 3 > a.n;                   // stream ends: the automaton decides
      // first match: seed records in reserved blocks; further ones (two patterns with the same twelve bases) one by one
      uint32_t rest = mm;
      bool first = true;
      while (__ballot(rest != 0)) {
        bool pass = false;
        uint32_t at = slot_at;
        if (rest) {
          const int sidx = __ffs(rest) - 1;
          rest &= rest - 1;
          uint32_t sv = 0;
#pragma unroll
          for (int t = 0; t < EDIT_BUCKET; ++t) sv = sidx == t ? sl[t] : sv;
          at += (uint32_t)sidx;
          const uint32_t plo = __builtin_amdgcn_perm(sv >> 16, wlo, PSEL);
          const uint32_t p4 = QC == 4 ? whi : sv >> 24;
          pass = ends || qgram_both(plo, (p4 << 8) | (plo >> 24), tl, th, a.edits);
          // One report per (window, pattern) among the undisplaced tests: a pattern that agrees with the window on
          // more than three pieces is found by every triple of them; the lexicographically first triple reports
          // (every combo tests the zero displacement, and its own three pieces are among the equal ones).
          if (pass && c6 == 18u) {
            const uint32_t x = plo ^ wraw;
            uint32_t eq = ((x & 0xffu) == 0 ? 1u : 0u) | ((x & 0xff00u) == 0 ? 2u : 0u) | ((x & 0xff0000u) == 0 ? 4u : 0u) | ((x >> 24) == 0 ? 8u : 0u) |
                          (((p4 ^ whi) & 0xffu) == 0 ? 16u : 0u);
            const uint32_t m1 = eq & (0u - eq); eq ^= m1;
            const uint32_t m2 = eq & (0u - eq); eq ^= m2;
            const uint32_t m3 = eq & (0u - eq);
            pass = (m1 | m2 | m3) == ((1u << QA) | (1u << QB) | (1u << QC));
          }
        }
        if (first) { emit(pass, p, at); first = false; }
        else if (pass) {
          const unsigned long long o = atomicAdd(a.counter, 1ull);
          if (o < a.cap) a.seed_out[o] = edit_seed_record(p, ((uint32_t)combo << 20) | at);
        }
      }
      if (full) edit_rare(a.self, combo, h2, wlo, whi, tl, th, p);
    }
  };

  // one block beyond the wave's range is loaded too: the last windows' q-gram tests look two bases ahead
  uint32_t q0 = load_packed<false>(a.packed, a.npacked, ws + 16 * lane);
  uint32_t q1 = load_packed<false>(a.packed, a.npacked, ws + 1024 + 16 * lane);
  uint32_t q2 = ws + 1024 < own_hi ? load_packed<false>(a.packed, a.npacked, ws + 2048 + 16 * lane) : 0u;
  uint32_t q3 = ws + 2048 < own_hi ? load_packed<false>(a.packed, a.npacked, ws + 3072 + 16 * lane) : 0u;
  const uint32_t mul0 = EDIT_MUL0, mul1 = EDIT_MUL1, mul2 = EDIT_MUL2, tmask = a.et_mask;
  for (int64_t bb = ws; bb < own_hi; bb += 1024) {
    const uint32_t cur = q0;
    q0 = q1; q1 = q2; q2 = q3;
    if (bb + 3072 < own_hi) q3 = load_packed<false>(a.packed, a.npacked, bb + 4096 + 16 * lane);
    const uint32_t prev1 = __builtin_amdgcn_update_dpp(carry1, cur, 0x138, 0xf, 0xf, false);       // wave_shr:1
    const uint32_t prev2 = __builtin_amdgcn_update_dpp(carry2, prev1, 0x138, 0xf, 0xf, false);
    carry2 = __builtin_amdgcn_readlane(cur, 62);
    carry1 = __builtin_amdgcn_readlane(cur, 63);
    const int64_t pbase = bb + 16 * lane;
    uint32_t own = 0xffffu;
    if (bb < own_lo || bb + 1024 > own_hi) {                       // wave-uniform: edge blocks only
      const int64_t lo = own_lo - pbase, hi = own_hi - pbase;
      const uint32_t l = lo <= 0 ? 0u : (lo >= 16 ? 16u : (uint32_t)lo), hh = hi <= 0 ? 0u : (hi >= 16 ? 16u : (uint32_t)hi);
      own = ((1u << hh) - 1u) & ~((1u << l) - 1u);
    }
    uint32_t sus[NV];                                              // per displacement pattern: the lane's suspicious windows
    static_for<NV>([&](auto V) __attribute__((always_inline)) { sus[decltype(V)::value] = 0; });

    static_for<2>([&](auto HH) __attribute__((always_inline)) {
      constexpr int HALF = decltype(HH)::value;
      // window i = 8 HALF + j of the lane starts at bit 26 + 2 i of prev2 : prev1 : cur, its piece q at + 8 q,
      // displaced by s bases at + 2 s
      uint32_t FA[NA], FB[NB], FC[8];
      static_for<NA>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        FA[t] = piece_hash_at<26 + 2 * (8 * HALF + LOA + t) + 8 * QA>(prev2, prev1, cur, mul0);
      });
      static_for<NB>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        FB[t] = piece_hash_at<26 + 2 * (8 * HALF + LOB + t) + 8 * QB>(prev2, prev1, cur, mul1);
      });
      static_for<8>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        FC[t] = piece_hash_at<26 + 2 * (8 * HALF + t) + 8 * QC>(prev2, prev1, cur, mul2);
      });

      uint32_t Hb[2][8], Eb[2][8], surv[2] = {0, 0};
      // consume stage of a unit: its table dwords have landed.  Suspicious = the key's bit is set
      auto consume = [&](auto PP) __attribute__((always_inline)) -> uint32_t {
        constexpr int P = decltype(PP)::value;
        uint32_t sacc = 0;
        static_for<8>([&](auto J) __attribute__((always_inline)) {
          constexpr int j = decltype(J)::value;
          uint32_t t;
          asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(t) : "v"(Hb[P][j]), "v"(Eb[P][j]));
          uint32_t &sr = sacc;
          asm volatile("v_alignbit_b32 %0, %1, %0, 1" : "+v"(sr) : "v"(t));   // (volatile: stays here, see the test stage)
        });
        return (sacc >> 24) & surv[P] & ((own >> (8 * HALF)) & 0xffu);
      };
      static_for<NV>([&](auto VV) __attribute__((always_inline)) {
        constexpr int V = decltype(VV)::value, P = V & 1;
        constexpr int SA = VL.sa[V], SB = VL.sb[V];
        __builtin_amdgcn_sched_barrier(0);                          // one unit at a time: hashes of later units computed early only cost registers
        uint32_t wd[8];
        static_for<8>([&](auto J) __attribute__((always_inline)) {
          constexpr int j = decltype(J)::value;
          Hb[P][j] = __builtin_amdgcn_bitop3_b32(FA[j + SA - LOA], FB[j + SB - LOB], FC[j], 0x96);   // three-way XOR
          wd[j] = *reinterpret_cast<lds_w32 *>((uintptr_t)(Hb[P][j] & 0x1fffcu));
        });
        __builtin_amdgcn_sched_barrier(0);                          // the eight reads go out before the first verdict waits
        uint32_t acc = 0;
        static_for<8>([&](auto J) __attribute__((always_inline)) {
          constexpr int j = decltype(J)::value;
          const uint32_t H = Hb[P][j], Jx = H >> jshift;
          uint32_t s1, s2, s3;
          asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(s1) : "v"(H), "v"(wd[j]));
          asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(s2) : "v"(H), "v"(wd[j]));
          asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(s3) : "v"(Jx), "v"(wd[j]));
          const uint32_t v = s1 & s2 & s3;
          // (volatile: the funnel stays here -- left to the scheduler it sinks to the consume stage and the eight verdicts are spilled)
          asm volatile("v_alignbit_b32 %0, %1, %0, 1" : "+v"(acc) : "v"(v));
          // the key's dword for a survivor, dword 0 otherwise (three-way AND)
          const uint32_t off = __builtin_amdgcn_bitop3_b32(Jx, (uint32_t)__builtin_amdgcn_sbfe((int)v, 0, 1), tmask, 0x80);
          Eb[P][j] = *reinterpret_cast<const uint32_t *>(etable + off);
        });
        surv[P] = acc >> 24;
        if constexpr (V > 0) sus[V - 1] |= consume(std::integral_constant<int, P ^ 1>{}) << (8 * HALF);
      });
      sus[NV - 1] |= consume(std::integral_constant<int, (NV - 1) & 1>{}) << (8 * HALF);
    });

    // compaction of the suspicious windows, one per lane and round (ballot + mbcnt give the queue slots)
    if (!(a.debug & 1)) {
      // the dword behind the lane's own (lane 63: the next block's first) -- a window's 24 bases reach two beyond it
      const uint32_t next = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readfirstlane(q0), cur, 0x130, 0xf, 0xf, false);   // wave_shl:1
#pragma unroll 1
      for (int v = 0; v < NV; ++v) {
        uint32_t rem = 0;
        static_for<NV>([&](auto T) __attribute__((always_inline)) {
          constexpr int t = decltype(T)::value;
          if (v == t) rem = sus[t];
        });
        for (;;) {
          const unsigned long long bal = __ballot(rem != 0);
          if (bal == 0) break;
          if (qn + 64 > EDIT_QCAP) process(false);
          if (rem != 0) {
            const int i = __ffs(rem) - 1;
            __builtin_assume(i >= 0 && i < 16);
            rem &= rem - 1;
            const int sft = 2 * i + 22;                              // base p-21 of window i in prev2 : prev1 : cur : next
            const uint32_t x0 = __builtin_amdgcn_alignbit(prev1, prev2, sft), x1 = __builtin_amdgcn_alignbit(cur, prev1, sft),
                           x2 = __builtin_amdgcn_alignbit(next, cur, sft);
            const int slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
            q_tl[slot] = sft < 32 ? x0 : x1;
            q_th[slot] = (sft < 32 ? x1 : x2) & 0xffffu;
            q_pv[slot] = (uint32_t)(pbase + i - ws) | ((uint32_t)v << 20);
          }
          qn += __popcll(bal);
        }
      }
    }
    if (qn >= 128) process(false);
  }
  process(true);
  if (lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
}

__global__ __launch_bounds__(SEED_THREADS) void pm_edit_scan(SeedArgs a) {
  extern __shared__ uint32_t lds[];
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
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.bloom + (size_t)combo * SEED_BLOOM_STRIDE);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < SEED_BLOOM_STRIDE / 4; i += SEED_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  switch (a.perm_sel[combo] & 0xffffffu) {                          // wave-uniform: the combo's pieces
    case 0x020100u: edit_scan_body<0, 1, 2, edit_cover_of(0, 1, 2)>(a, combo, cj, lds); break;
    case 0x030100u: edit_scan_body<0, 1, 3, edit_cover_of(0, 1, 3)>(a, combo, cj, lds); break;
    case 0x040100u: edit_scan_body<0, 1, 4, edit_cover_of(0, 1, 4)>(a, combo, cj, lds); break;
    case 0x030200u: edit_scan_body<0, 2, 3, edit_cover_of(0, 2, 3)>(a, combo, cj, lds); break;
    case 0x040200u: edit_scan_body<0, 2, 4, edit_cover_of(0, 2, 4)>(a, combo, cj, lds); break;
    case 0x040300u: edit_scan_body<0, 3, 4, edit_cover_of(0, 3, 4)>(a, combo, cj, lds); break;
    case 0x030201u: edit_scan_body<1, 2, 3, edit_cover_of(1, 2, 3)>(a, combo, cj, lds); break;
    case 0x040201u: edit_scan_body<1, 2, 4, edit_cover_of(1, 2, 4)>(a, combo, cj, lds); break;
    case 0x040301u: edit_scan_body<1, 3, 4, edit_cover_of(1, 3, 4)>(a, combo, cj, lds); break;
    case 0x040302u: edit_scan_body<2, 3, 4, edit_cover_of(2, 3, 4)>(a, combo, cj, lds); break;
    default: break;
  }
}

// ---- exact_halves -k, ranked form: pm_half_scan + pm_half_verify ------------------------------------
//
// A seed is an exact occurrence of a half (exact_halves.cc:199-224); at 10^5 primers 0.31 positions per
// base end one (4^10 keys, 4e5 halves).  The HALVES instance of pm_seed_scan resolves every one of
// them through a bucket, a 32-byte record and two raw-stream reads -- 20 MB of tables that do not fit
// an XCD's L2, and 10^9 random 64-byte reads of the raw stream per 3 Gbp.  Here, for halves of >= 10
// bases (patterns of >= 20):
//   * the LDS image is the exact bitmap of the halves' last ten bases plus a rank directory (the
//     layout of pm_pair.hip): a key hit's RANK among the set bits indexes a dense table with one 8-byte
//     slot per distinct key -- the partner half at 2 bits per base, its length, the side it lies on;
//     further halves with the same key (a third of the key hits) follow in a second dense table;
//   * the queue entry carries the 48 stream bases p-31 .. p+16 (three dwords the lane has in registers),
//     so the partner test (one of the partner's k+1 pieces unedited within k positions) runs on
//     registers: one cache line per key hit, no stream re-read;
//   * what passes (a few per thousand key hits) leaves as 8-byte records (rank, position); pm_half_verify
//     walks the key's halves with the exact test on the raw stream (half_seed_ok: N and EOS reject, the
//     half's bases in front of the key) and writes the seed records pm_seed_extend takes.
// Halves whose partner window does not fit the 48 carried bases (patterns of 31, 32 characters) pass
// the first kernel unfiltered.
constexpr int HR_BITMAP_WORDS = 32768, HR_SUPER = 512, HR_REL_WORDS = 2048;
constexpr int HR_IMAGE_WORDS = HR_BITMAP_WORDS + HR_SUPER + HR_REL_WORDS;
constexpr int HR_QCAP = 80;                                         // key hits per wave queue (16-byte entries)
static_assert(HR_IMAGE_WORDS * 4 + WAVES * HR_QCAP * 16 <= SEED_LDS_BYTES, "LDS");
// slot.y: partner length (5 bits) | side << 5 | (half length - 10) << 6 | further halves with this key (3 bits, 7 = seven or more) << 9 | their first index << 12
__device__ __host__ __forceinline__ uint32_t half_slot_info(int plen, int side, int L, uint32_t nmore, uint32_t more_at) {
  return (uint32_t)plen | ((uint32_t)side << 5) | ((uint32_t)(L - 10) << 6) | ((nmore > 7 ? 7u : nmore) << 9) | (more_at << 12);
}

// the partner test on the carried stream bases D (base 0 = stream position p - 31); true = cannot be ruled out
__device__ __forceinline__ bool half_partner_fast(const SeedArgs &a, uint32_t part, uint32_t info, uint32_t d0, uint32_t d1, uint32_t d2, int64_t p) {
  const int k = a.hk;
  const int plen = info & 31u, side = (info >> 5) & 1u, L = 10 + (int)((info >> 6) & 7u);
  const int o = side ? 32 - L - plen - k : 32 - k;                  // first base of the partner's window (partner_window) in D
  const int64_t r0 = p - 31 + o;
  if (o < 0 || o + plen + 2 * k > 48 || r0 < 0 || r0 + 24 > a.n) return true;
  const uint32_t w = (uint32_t)(2 * o) >> 5, sh = (uint32_t)(2 * o) & 31u;
  const uint32_t a0 = w == 0 ? d0 : (w == 1 ? d1 : d2), a1 = w == 0 ? d1 : (w == 1 ? d2 : 0u), a2 = w == 0 ? d2 : 0u;
  const uint64_t T = ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sh) << 32) | __builtin_amdgcn_alignbit(a1, a0, sh);
  return partner_possible_packed(k, plen, part, T);
}

__global__ __launch_bounds__(SEED_THREADS) void pm_half_scan(SeedArgs a) {
  extern __shared__ uint32_t lds[];
  const int cj = blockIdx.x;
  if (cj >= a.nchunks) return;
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(a.hr_image);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int i = threadIdx.x; i < HR_IMAGE_WORDS / 4; i += SEED_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t *q_d0 = lds + HR_IMAGE_WORDS + wave * (4 * HR_QCAP), *q_d1 = q_d0 + HR_QCAP, *q_d2 = q_d1 + HR_QCAP, *q_ps = q_d2 + HR_QCAP;
  const int64_t sub = a.chunk_len / WAVES;
  const int64_t ws = (a.chunk0 + cj) * a.chunk_len + (int64_t)wave * sub;   // p = the half's last base
  int64_t own_lo = ws > a.begin ? ws : a.begin;
  int64_t own_hi = ws + sub;
  if (own_hi > a.end) own_hi = a.end;
  if (own_hi > a.n) own_hi = a.n;
  if (own_lo < 9) own_lo = 9;                                      // the ten key bases must fit in the stream
  if (own_lo >= own_hi) return;
  uint32_t carry1, carry2;
  {
    const uint32_t pk = load_packed<false>(a.packed, a.npacked, ws - 32 + 16 * (lane & 1));
    carry2 = __builtin_amdgcn_readlane(pk, 0);
    carry1 = __builtin_amdgcn_readlane(pk, 1);
  }
  int qn = 0;
  unsigned long long ob_next = 0;
  int ob_left = 0;
  auto emit = [&](bool pass, int64_t p, uint32_t rank) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(pass);
    if (bal == 0) return;
    const int c = __popcll(bal);
    if (c > ob_left) {
      if (lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
      ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)base);
      ob_left = SEED_OUT_BLOCK;
    }
    if (pass) {
      const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (slot < a.cap) a.seed_out[slot] = edit_seed_record(p, rank);
    }
    ob_next += c; ob_left -= c;
  };
  // whole batches of 64 key hits (all: what is left, too): rank -> slot -> partner test on the carried bases
  auto process = [&](bool all) __attribute__((always_inline)) {
    while (qn >= 64 || (all && qn > 0)) {
      const int cnt = qn >= 64 ? 64 : qn;
      qn -= cnt;
      if (a.debug & 2) continue;
      const bool on = lane < cnt;
      uint32_t d0 = 0, d1 = 0, d2 = 0, rank = 0;
      int64_t p = 0;
      uint2 sv = make_uint2(0, 0);
      if (on) {
        d0 = q_d0[qn + lane]; d1 = q_d1[qn + lane]; d2 = q_d2[qn + lane];
        p = ws + q_ps[qn + lane];
        const uint32_t key = d1 >> 12;                               // bases p-9 .. p
        // rank of the key among the set bits: superblock (2048 bits) + block (256 bits) + words + bit
        const uint32_t word = key & 0x7fffu, bit = key >> 15;
        const uint32_t sup = lds[HR_BITMAP_WORDS + (word >> 6)];
        const uint32_t rel = reinterpret_cast<const uint16_t *>(lds + HR_BITMAP_WORDS + HR_SUPER)[word >> 3];
        const u32x4 b0 = *reinterpret_cast<const u32x4 *>(lds + ((word >> 3) << 3)), b1 = *reinterpret_cast<const u32x4 *>(lds + ((word >> 3) << 3) + 4);
        const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const uint32_t wq = word & 7u;
        uint32_t c = sup + rel, part = 0;
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
          c += t < wq ? (uint32_t)__popc(bw[t]) : 0u;
          part = t == wq ? bw[t] : part;
        }
        rank = c + __popc(__builtin_amdgcn_ubfe(part, 0, bit));
        sv = a.hr_slots[rank];
      }
      bool pass = false;
      uint32_t more = 0, at = 0;
      if (on && !(a.debug & 4)) {
        pass = half_partner_fast(a, sv.x, sv.y, d0, d1, d2, p);
        more = (sv.y >> 9) & 7u; at = sv.y >> 12;
        if (more == 7u) { pass = true; more = 0; }                   // many halves share the key: the verify kernel walks them
        if (pass) more = 0;
      }
      while (__ballot(more != 0)) {                                  // further halves with the same key
        if (more) {
          const uint2 mv = a.hr_more[at];
          ++at; --more;
          if (half_partner_fast(a, mv.x, mv.y, d0, d1, d2, p)) { pass = true; more = 0; }
        }
      }
      emit(pass, p, rank);
    }
  };

  uint32_t q0 = load_packed<false>(a.packed, a.npacked, ws + 16 * lane);
  uint32_t q1 = load_packed<false>(a.packed, a.npacked, ws + 1024 + 16 * lane);      // (one block beyond the range: the carried bases reach 16 ahead)
  uint32_t q2 = ws + 1024 < own_hi ? load_packed<false>(a.packed, a.npacked, ws + 2048 + 16 * lane) : 0u;
  uint32_t q3 = ws + 2048 < own_hi ? load_packed<false>(a.packed, a.npacked, ws + 3072 + 16 * lane) : 0u;
  for (int64_t bb = ws; bb < own_hi; bb += 1024) {
    const uint32_t cur = q0;
    q0 = q1; q1 = q2; q2 = q3;
    if (bb + 3072 < own_hi) q3 = load_packed<false>(a.packed, a.npacked, bb + 4096 + 16 * lane);
    const uint32_t prev1 = __builtin_amdgcn_update_dpp(carry1, cur, 0x138, 0xf, 0xf, false);       // wave_shr:1
    const uint32_t prev2 = __builtin_amdgcn_update_dpp(carry2, prev1, 0x138, 0xf, 0xf, false);
    carry2 = __builtin_amdgcn_readlane(cur, 62);
    carry1 = __builtin_amdgcn_readlane(cur, 63);
    const int64_t pbase = bb + 16 * lane;
    uint32_t own = 0xffffu;
    if (bb < own_lo || bb + 1024 > own_hi) {                       // wave-uniform: edge blocks only
      const int64_t lo = own_lo - pbase, hi = own_hi - pbase;
      const uint32_t l = lo <= 0 ? 0u : (lo >= 16 ? 16u : (uint32_t)lo), hh = hi <= 0 ? 0u : (hi >= 16 ? 16u : (uint32_t)hi);
      own = ((1u << hh) - 1u) & ~((1u << l) - 1u);
    }
    // the key of window i (its last ten bases) sits at bits 2i + 46 .. 2i + 65 of prev2 : prev1 : cur
    uint32_t acc = 0;
    static_for<2>([&](auto HH) __attribute__((always_inline)) {
      constexpr int HALF = decltype(HH)::value;
      uint32_t ks[8], wd[8];
      static_for<8>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, i = 8 * HALF + j;
        ks[j] = bits_at<2 * i + 44>(prev2, prev1, cur);              // the key at bits 2 .. 21
        wd[j] = *reinterpret_cast<lds_w32 *>((uintptr_t)(ks[j] & 0x1fffcu));
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<8>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        acc = __builtin_amdgcn_alignbit(wd[j] >> ((ks[j] >> 17) & 31u), acc, 1);
      });
    });
    uint32_t rem = (acc >> 16) & own;
    if (a.debug & 1) rem = 0;
    const uint32_t next = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readfirstlane(q0), cur, 0x130, 0xf, 0xf, false);   // wave_shl:1
    for (;;) {
      const unsigned long long bal = __ballot(rem != 0);
      if (bal == 0) break;
      const int slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
      if (rem != 0 && slot < HR_QCAP) {                              // lanes the queue has no room for keep their window for the next round
        const int i = __ffs(rem) - 1;
        __builtin_assume(i >= 0 && i < 16);
        rem &= rem - 1;
        const int sft = 2 * i + 2;                                   // stream base p - 31 in prev2 : prev1 : cur : next
        const bool whole = sft >= 32;                                // i = 15: the three dwords as they are
        const uint32_t x0 = whole ? prev1 : __builtin_amdgcn_alignbit(prev1, prev2, sft), x1 = whole ? cur : __builtin_amdgcn_alignbit(cur, prev1, sft),
                       x2 = whole ? next : __builtin_amdgcn_alignbit(next, cur, sft);
        q_d0[slot] = x0; q_d1[slot] = x1; q_d2[slot] = x2;
        q_ps[slot] = (uint32_t)(pbase + i - ws);
      }
      qn += __popcll(bal);
      if (qn > HR_QCAP) qn = HR_QCAP;
      if (qn >= 64) process(false);                                  // whole batches from the top of the queue; fewer than 64 stay
    }
  }
  process(true);
  if (lane < ob_left && ob_next + lane < a.cap) a.seed_out[ob_next + lane] = ~0ull;
}

// Second kernel of the edit-distance plan: the seed list is dense (every lane has work), one seed
// per lane, grid-stride; the seed count is read from device memory (no host round trip).
struct EditVerifyArgs {
  SeedArgs a;                           // text, records (pat_codes), eos, edits, maxlen, begin/end of the scan
  const uint64_t *seeds;
  const unsigned long long *nseeds;
  unsigned long long seed_cap;
};

// pattern of a seed record: pm_edit_scan writes (combo << 20 | bucket slot) -- the pattern index of a slot sits in a
// table of its own (a.eidx) that the scan kernel never reads, so it does not compete for L2 with the slots; the
// round-1 first stage writes the pattern index itself
__device__ __forceinline__ uint32_t seed_pattern(const SeedArgs &a, uint32_t code) {
  return a.eidx ? a.eidx[(size_t)(code >> 20) * ((size_t)EDIT_BUCKET << (32 - a.bucket_shift)) + (code & 0xfffffu)] : code;
}

__global__ __launch_bounds__(256) void pm_edits_verify(EditVerifyArgs v) {
  const SeedArgs &a = v.a;
  unsigned long long n = *v.nseeds;
  if (n > v.seed_cap) n = v.seed_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long rounds = (n + stride - 1) / stride;     // same trip count for every lane: ballots below stay whole-wave
  const int lane = threadIdx.x & 63;
  // output slots are reserved 64 at a time per wave (one counter serves every wave of the grid and
  // same-address atomics serialise); what a wave leaves unused is marked PM_SEED_HOLE, which the
  // dedup that follows drops
  unsigned long long ob_next = 0;
  int ob_left = 0;
  for (unsigned long long it = 0; it < rounds; ++it) {
    const unsigned long long i = it * stride + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t res = 0, pid = 0;
    int64_t p = 0;
    if (i < n) {
      const uint64_t sd = v.seeds[i];
      if (sd != ~0ull) { p = (int64_t)(sd & 0xffffffffffull); res = edits_verify(a, p, seed_pattern(a, (uint32_t)(sd >> 40)), &pid); }
    }
    if (__ballot(res != 0) == 0) continue;
#pragma unroll 1
    for (int d = 0; d < 5; ++d) {
      const uint32_t lvl1 = (res >> (4 * d)) & 15u;
      const int64_t e = p - 1 + d;
      const bool pass = lvl1 != 0 && e > a.begin && e <= a.end;
      const unsigned long long bal = __ballot(pass);
      if (bal == 0) continue;
      const int c = __popcll(bal);
      if (c > ob_left) {
        if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
        ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                  __builtin_amdgcn_readfirstlane((uint32_t)base);
        ob_left = SEED_OUT_BLOCK;
      }
      if (pass) {
        const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
        if (slot < a.cap) a.out[slot] = edit_record(e, pid, lvl1);
      }
      ob_next += c; ob_left -= c;
    }
  }
  if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
}

// Second kernel of the ranked exact_halves -k plan: every record (rank of a key, position) is resolved
// into the halves that have the key; the exact test on the raw stream (half_seed_ok) decides, the seed
// records go out in reserved blocks.
__global__ __launch_bounds__(256) void pm_half_verify(EditVerifyArgs v) {
  const SeedArgs &a = v.a;
  unsigned long long n = *v.nseeds;
  if (n > v.seed_cap) n = v.seed_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long rounds = (n + stride - 1) / stride;     // same trip count for every lane: ballots stay whole-wave
  const int lane = threadIdx.x & 63;
  unsigned long long ob_next = 0;
  int ob_left = 0;
  for (unsigned long long it = 0; it < rounds; ++it) {
    const unsigned long long i = it * stride + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t p = 0;
    uint32_t t = 0, t1 = 0;
    if (i < n) {
      const uint64_t sd = v.seeds[i];
      if (sd != ~0ull) { p = (int64_t)(sd & 0xffffffffffull); const uint32_t rank = (uint32_t)(sd >> 40); t = a.hr_first[rank]; t1 = a.hr_first[rank + 1]; }
    }
    while (__ballot(t < t1)) {
      uint32_t pid = 0;
      bool pass = false;
      if (t < t1) { pass = half_seed_ok(a, p, a.hr_order[t], &pid); ++t; }
      const unsigned long long bal = __ballot(pass);
      if (bal == 0) continue;
      const int c = __popcll(bal);
      if (c > ob_left) {
        if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
        ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                  __builtin_amdgcn_readfirstlane((uint32_t)base);
        ob_left = SEED_OUT_BLOCK;
      }
      if (pass) {
        const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
        if (slot < a.cap) a.out[slot] = half_seed_record(p, pid);
      }
      ob_next += c; ob_left -= c;
    }
  }
  if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
}

// exact_bases with edits on the seed family.  exact_bases (exact_bases.cc:69-129) reports, for every
// exact occurrence of a pattern's mandated block (its first esb or last eeb characters), the banded
// extension of the remainder when it succeeds (primer_alignment.cc:568-728).  Such a hit is a window
// within k edits of the whole pattern, which the edit-distance plan's seeds (three clean pieces under a
// displacement pattern + the q-gram test) cannot miss -- while exact block seeds (4^8 keys for an
// 8-base block: every position is a seed of some pattern) would flood the second stage.  The k-error
// automaton itself is NOT the right test here: behind an end-of-sequence character it cannot delete a
// pattern's first characters (rows cleared), the extension DP can.  So: for a seed (pattern, p) the
// pattern ends within k of p+1; try the block at every start that allows, and emit the reference's seed
// record (end of the block occurrence) for the host's extension DPs; duplicates leave with the dedup.
struct BasesArgs {
  const uint8_t *codes, *len;
  const int32_t *esb, *eeb;
  int64_t own_lo, own_hi;
};

__global__ __launch_bounds__(256) void pm_bases_verify(EditVerifyArgs v, BasesArgs b) {
  const SeedArgs &a = v.a;
  unsigned long long n = *v.nseeds;
  if (n > v.seed_cap) n = v.seed_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long rounds = (n + stride - 1) / stride;     // same trip count for every lane: ballots stay whole-wave
  const int lane = threadIdx.x & 63;
  const int k = a.edits;
  unsigned long long ob_next = 0;
  int ob_left = 0;
  for (unsigned long long it = 0; it < rounds; ++it) {
    const unsigned long long i = it * stride + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    int64_t p = 0;
    uint32_t pid = 0;
    int L = 0, blk = 0;
    bool prefix = true;
    const uint8_t *pc = nullptr;
    if (i < n) {
      const uint64_t sd = v.seeds[i];
      if (sd != ~0ull) {
        live = true;
        p = (int64_t)(sd & 0xffffffffffull);
        pid = a.pat_id[seed_pattern(a, (uint32_t)(sd >> 40))];        // 1-based index into the whole pattern list
        L = b.len[pid - 1];
        const int es = b.esb[pid - 1], ee = b.eeb[pid - 1];
        prefix = es >= ee;                                           // exact_bases.cc:139-150: the larger block decides
        blk = prefix ? es : ee;
        pc = b.codes + (size_t)(pid - 1) * 32 + (prefix ? 0 : L - blk);
      }
    }
    // pattern end e in p+1-k .. p+1+k; prefix block: the pattern starts at e - L - d, |d| <= k
    for (int t = -2 * k; t <= 2 * k; ++t) {
      bool ok = live && blk > 0 && (prefix || (t >= -k && t <= k));
      const int64_t b0 = prefix ? p + 1 - L + t : p + 1 + t - blk;
      ok = ok && b0 >= 0 && b0 + blk <= a.n;
      const int64_t send = b0 + blk;
      ok = ok && send > b.own_lo && send <= b.own_hi;
      for (int q = 0; q < blk && ok; ++q) ok = a.text[b0 + q] == pc[q];
      const unsigned long long bal = __ballot(ok);
      if (bal == 0) continue;
      const int c = __popcll(bal);
      if (c > ob_left) {
        if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.counter, (unsigned long long)SEED_OUT_BLOCK);
        ob_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                  __builtin_amdgcn_readfirstlane((uint32_t)base);
        ob_left = SEED_OUT_BLOCK;
      }
      if (ok) {
        const unsigned long long slot = ob_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
        if (slot < a.cap) { pm_hit hh; hh.end = send; hh.pid = pid; hh.k = 0; hh.aux[0] = hh.aux[1] = hh.aux[2] = 0; a.out[slot] = hh; }
      }
      ob_next += c; ob_left -= c;
    }
  }
  if (lane < ob_left && ob_next + lane < a.cap) a.out[ob_next + lane].pid = PM_SEED_HOLE;
}

// The stream at 2 bits per base, made once per pm_init (a re-encoding of the database like the
// reference's compress_seq output formats, not part of a scan): dword i = bases 16i .. 16i+15,
// base j in bits 2j, codes as pack4 derives them (A,C,G,T exact; any other byte aliases to one of
// them and is weeded out where the full characters are compared).
__global__ void pm_pack_stream(const uint8_t *text, int64_t n, int sh, uint32_t *packed, int64_t npacked) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= npacked) return;
  packed[i] = pack16(load16<true>(text, 16 * i, n), sh);
}

}  // namespace

hipError_t pack_stream(const uint8_t *d_text, int64_t n, bool ascii, uint32_t *d_packed, int64_t npacked, hipStream_t st) {
  if (npacked <= 0) return hipSuccess;
  const int threads = 256;
  hipLaunchKernelGGL(pm_pack_stream, dim3((unsigned)((npacked + threads - 1) / threads)), dim3(threads), 0, st,
                     d_text, n, ascii ? 1 : 0, d_packed, npacked);
  return hipGetLastError();
}

// ---- host side: plan, tables, launch ------------------------------------------------------------

static uint64_t binom(int n, int r) {
  uint64_t v = 1;
  for (int i = 1; i <= r; ++i) v = v * (n - r + i) / i;
  return v;
}

// Which (combo, displacement pattern) pairs the edit-distance first stage has to test.
// Every way of placing <= k edits on the m = k+3 pieces of the seeded window (a substitution, an
// inserted or a deleted character inside a piece -- the piece is then unusable -- or characters
// inserted between two pieces) leaves >= 3 clean pieces, each displaced by the indels to its
// right.  A placement is found if one of its clean triples is tested under its displacement
// pattern (d1 = second minus third piece, d2 = first minus second).  Greedy set cover over all
// placements: 34 pairs for k = 2 (of 85 that can occur, 130 in all) and 8 for k = 1 -- both equal the
// optimum of the exact set-cover integer program, so the greedy choice loses nothing here.
static void edit_cover(int k, int m, const std::vector<std::array<int, 4>> &combos, uint32_t *evar) {
  static const int VD1[13] = {0, 0, 0, 1, -1, 0, 0, 2, -2, 1, 1, -1, -1};
  static const int VD2[13] = {0, 1, -1, 0, 0, 2, -2, 0, 0, 1, -1, 1, -1};
  const int C = (int)combos.size();
  struct Loc { int kind, x; };                         // 0 substitution, 1 insertion, 2 deletion inside piece x; 3 insertion after piece x
  std::vector<Loc> locs;
  for (int kind = 0; kind < 3; ++kind) for (int q = 0; q < m; ++q) locs.push_back({kind, q});
  for (int b = 0; b + 1 < m; ++b) locs.push_back({3, b});
  std::vector<std::vector<int>> scen(1);               // placements as lists of location indices (with repetition)
  for (int i = 0; i < (int)locs.size(); ++i) {
    scen.push_back({i});
    if (k >= 2) for (int j = i; j < (int)locs.size(); ++j) scen.push_back({i, j});
  }
  std::vector<std::vector<uint8_t>> covers(scen.size(), std::vector<uint8_t>((size_t)C * 13, 0));
  for (size_t si = 0; si < scen.size(); ++si) {
    std::vector<int> shift(m, 0);
    std::vector<bool> dirty(m, false);
    for (int li : scen[si]) {
      const Loc &l = locs[li];
      if (l.kind < 3) dirty[l.x] = true;
      if (l.kind == 1) for (int q = 0; q < l.x; ++q) --shift[q];          // extra stream character: pieces to the left sit further left
      if (l.kind == 2) for (int q = 0; q < l.x; ++q) ++shift[q];
      if (l.kind == 3) for (int q = 0; q <= l.x; ++q) --shift[q];
    }
    for (int c = 0; c < C; ++c) {
      const int pa = combos[c][0], pb = combos[c][1], pc = combos[c][2];
      if (dirty[pa] || dirty[pb] || dirty[pc]) continue;
      const int d1 = shift[pb] - shift[pc], d2 = shift[pa] - shift[pb];
      for (int v = 0; v < 13; ++v) if (VD1[v] == d1 && VD2[v] == d2) covers[si][(size_t)c * 13 + v] = 1;
    }
  }
  for (int c = 0; c < C; ++c) evar[c] = 0;
  std::vector<bool> done(scen.size(), false);
  for (;;) {
    int best = -1, bestn = 0;
    for (int u = 0; u < C * 13; ++u) {
      int cnt = 0;
      for (size_t si = 0; si < scen.size(); ++si) if (!done[si] && covers[si][u]) ++cnt;
      if (cnt > bestn) { bestn = cnt; best = u; }
    }
    if (best < 0) break;
    evar[best / 13] |= 1u << (best % 13);
    for (size_t si = 0; si < scen.size(); ++si) if (covers[si][best]) done[si] = true;
  }
}

std::string seed_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids,
                       const Alphabet &alpha, int k, int eos_code, SeedTables *out, int force_lmin,
                       const std::vector<std::string> *partners, const std::vector<uint8_t> *sides, int halves_k, bool edits, const Knobs &knobs) {
  SeedTables &t = *out;
  t = SeedTables();
  t.k = k;
  if (k < 0 || k > 3) return "seed kernels are built for k <= 3";
  // stream packing: codes 0..3 = A,C,G,T (compress_seq -n true) or raw ASCII ((c >> 1) & 3)
  const bool norm = alpha.nch['A'] == 0 && alpha.nch['C'] == 1 && alpha.nch['G'] == 2 && alpha.nch['T'] == 3;
  const bool ascii = alpha.size == 256 && alpha.nch['A'] == 'A' && alpha.nch['C'] == 'C' && alpha.nch['G'] == 'G' && alpha.nch['T'] == 'T';
  if (!norm && !ascii) return "stream alphabet is neither A,C,G,T-normalized nor raw ASCII";
  t.ascii = ascii && !norm;
  auto base2 = [&](unsigned char ch) -> int {
    switch (ch) { case 'A': return t.ascii ? 0 : 0; case 'C': return 1; case 'G': return t.ascii ? 3 : 2; case 'T': return t.ascii ? 2 : 3; }
    return -1;
  };
  int lmin = 1 << 30, lmax = 0;
  for (const Pattern &p : pats) {
    if (p.s.empty()) return "empty pattern";
    if (p.s.size() > 32) return "pattern longer than 32 characters";
    for (unsigned char ch : p.s) if (base2(ch) < 0) return "pattern with characters other than A,C,G,T";
    lmin = std::min(lmin, (int)p.s.size()); lmax = std::max(lmax, (int)p.s.size());
  }
  if (pats.empty()) { lmin = lmax = 1; }
  t.maxlen = lmax;
  if (force_lmin > 0) lmin = std::min(lmin, force_lmin);
  t.Lw = std::min(lmin, 20);
  // choose m = k + r pieces of pb bases: fewest filter lookups + verifies
  t.mode = 0;
  if (t.Lw == 20 && k <= 2) {
    // 4-base pieces are bytes of the packed window: one v_perm builds the key
    t.pb = 4;
    t.r = k == 0 ? 4 : 3;
    t.mode = k == 0 ? 2 : 1;
  } else {
    double best = 1e300;
    const double P = std::max<size_t>(pats.size(), 1);
    for (int r = 1; r <= 4; ++r) {
      const int m = k + r, pb = std::min(t.Lw / m, 16 / r);
      if (pb < 1 || m > 8) continue;
      const double C = (double)binom(m, r);
      if (C > SEED_MAX_COMBOS) continue;
      double keys = 1; for (int i = 0; i < r * pb; ++i) keys *= 4;
      const double cost = C * 1.1 + 20.0 * C * P / keys;
      if (cost < best) { best = cost; t.r = r; t.pb = pb; }
    }
    if (best > 1e299) return "patterns too short for a k-mismatch seed plan";
  }
  t.edits = edits ? k : 0;
  if (edits && (t.mode != 1 || k < 1 || k > 2)) return "the edit-distance seed plan needs 20..32 character patterns and k = 1 or 2";
  const int m = k + t.r;
  // enumerate combos (r-subsets of m pieces) in lexicographic order
  std::vector<int> c(t.r);
  for (int i = 0; i < t.r; ++i) c[i] = i;
  for (;;) {
    std::array<int, 4> cc = {0, 0, 0, 0};
    for (int i = 0; i < t.r; ++i) cc[i] = c[i];
    t.combos.push_back(cc);
    int i = t.r - 1;
    while (i >= 0 && c[i] == m - t.r + i) --i;
    if (i < 0) break;
    ++c[i];
    for (int j = i + 1; j < t.r; ++j) c[j] = c[j - 1] + 1;
  }
  const int C = (int)t.combos.size();
  t.perm_sel.assign(C, 0);
  for (int ci = 0; ci < C; ++ci) {
    uint32_t sel = 0;
    for (int q = 0; q < 4; ++q) sel |= (q < t.r ? (uint32_t)t.combos[ci][q] : 0x0cu) << (8 * q);
    t.perm_sel[ci] = sel;
  }
  const size_t np = pats.size();
  t.pat40.resize(np); t.pat_len.resize(np); t.pat_id.resize(np); t.pat_codes.assign(np * 32, 0);
  t.halves = partners != nullptr; t.hk = halves_k;
  t.part32.assign(np, 0); t.part_len.assign(np, 0); t.part_side.assign(np, 0);
  if (partners)
    for (size_t j = 0; j < np; ++j) {
      const std::string &ps = (*partners)[j];
      if (ps.size() > 16) return "exact_halves partner longer than 16 characters";
      uint32_t w2 = 0;
      for (size_t i = 0; i < ps.size(); ++i) { const int b2 = base2((unsigned char)ps[i]); if (b2 < 0) return "pattern with characters other than A,C,G,T"; w2 |= (uint32_t)b2 << (2 * i); }
      t.part32[j] = w2; t.part_len[j] = (uint8_t)ps.size(); t.part_side[j] = (*sides)[j];
    }
  // halves whose key is one piece of <= 10 bases at the low end of the window: 4^10 keys = the bits of the LDS filter
  t.exact_filter = partners != nullptr && t.mode == 0 && C == 1 && t.r == 1 && t.combos[0][0] == 0 && 2 * t.pb <= 20;
  t.hfast = 0;
  if (partners && np > 0) {                        // same total length everywhere and side = parity of the index?
    const int tot = (int)pats[0].s.size() + t.part_len[0];
    bool ok = true;
    for (size_t j = 0; j < np && ok; ++j) ok = (int)pats[j].s.size() + t.part_len[j] == tot && t.part_side[j] == (j & 1);
    if (ok) t.hfast = tot;
  }
  // buckets of 8 slots, average fill <= 3; slot = fingerprint (high bits) | pattern index (low idx_bits)
  int idx_bits = 1;
  while (((size_t)1 << idx_bits) <= np) ++idx_bits;
  const bool edit_bloom_v1 = edits && knobs.edit_bloom;
  const bool tabulated = edits && !edit_bloom_v1;  // first stage pm_edit_scan: 64-byte buckets of 16 slots (a full bucket is then a 1e-3 event)
  const size_t bslots = tabulated ? EDIT_BUCKET : 8;
  size_t nbuckets = 256;
  // (tabulated: 200k patterns -> 32768 buckets = 2 MiB per combo; 65536 with the 1 MiB key map overflowed an XCD's 4 MiB of L2)
  while (tabulated ? nbuckets * 7 * EDIT_BUCKET < np * 16 : nbuckets * 3 < np) nbuckets <<= 1;
  int lb = 0;
  while (((size_t)1 << lb) < nbuckets) ++lb;
  t.idx_bits = idx_bits; t.bucket_shift = 32 - lb;
  const size_t nslots = nbuckets * bslots;
  t.nslots = nslots;
  t.bloom.assign((size_t)C * SEED_BLOOM_STRIDE, 0);
  int lb2 = 16;
  while (((size_t)1 << lb2) < 20 * np && lb2 < 26) ++lb2;
  if (edits) lb2 = 5;                              // pm_edit_scan has its key map instead (the older first stage builds the bitmap on request)
  if (edit_bloom_v1) { lb2 = 16; while (((size_t)1 << lb2) < 20 * np && lb2 < 26) ++lb2; }
  t.lb2 = lb2;
  t.bitmap2.assign((size_t)C << (lb2 - 5), 0);
  t.etable_log = 0;
  if (tabulated) {
    t.etable_log = 23;                               // 2^23 key-hash bits = 1 MiB per combo: with the 2 MiB of buckets it stays in one XCD's L2
    if (knobs.edit_table_log >= 16 && knobs.edit_table_log <= 26) t.etable_log = knobs.edit_table_log;
    if (nslots > ((size_t)1 << 20)) return "pattern tile too large for the edit-distance plan's seed records (20-bit slot index)";
    t.etable.assign((size_t)C << (t.etable_log - 3), 0); t.eidx.assign((size_t)C * nslots, 0);
  }
  t.slots.assign((size_t)C * nslots, EMPTY);
  for (int i = 0; i < 256; ++i) t.cmap[i] = 0;
  t.eos_code = -1;
  if (eos_code >= 0 && eos_code < 256) { t.cmap[eos_code] = 1; t.eos_code = eos_code; }
  const uint64_t pmask = (1ull << (2 * t.pb)) - 1ull;
  for (size_t j = 0; j < np; ++j) {
    const std::string &s = pats[j].s;
    const int L = (int)s.size();
    uint64_t w = 0;
    for (int i = 0; i < t.Lw; ++i) w |= (uint64_t)base2((unsigned char)s[L - t.Lw + i]) << (2 * i);
    t.pat40[j] = {(uint32_t)w, (uint32_t)(w >> 32)};
    t.pat_len[j] = (uint8_t)L;
    t.pat_id[j] = ids[j];
    for (int i = 0; i < L; ++i) t.pat_codes[j * 32 + i] = (uint8_t)alpha.nch[(unsigned char)s[i]];
    if (edits) {                                   // the row becomes the automaton record edits_verify reads
      uint32_t M[4] = {0, 0, 0, 0};
      for (int i = 0; i < L; ++i) M[base2((unsigned char)s[i]) ^ (t.ascii && base2((unsigned char)s[i]) >= 2 ? 1 : 0)] |= 1u << i;   // A,C,G,T order
      uint8_t *rec = &t.pat_codes[j * 32];
      const uint32_t len = (uint32_t)L, idv = ids[j];
      memcpy(rec, M, 16); memcpy(rec + 16, &len, 4); memcpy(rec + 20, &idv, 4); memset(rec + 24, 0, 8);
    }
    if (partners) {                                // halves: the row doubles as the 32-byte record verify_half reads
      if (L > 16) return "exact_halves half longer than 16 characters";
      uint8_t *rec = &t.pat_codes[j * 32];
      const uint32_t w32 = t.part32[j], idv = ids[j];
      memmove(rec + 16 - L, rec, (size_t)L);         // codes right-aligned in bytes [16-L, 16)
      memset(rec, 0, (size_t)(16 - L));
      memcpy(rec + 16, &w32, 4);
      rec[20] = (uint8_t)L; rec[21] = t.part_len[j]; rec[22] = t.part_side[j]; rec[23] = 0;
      memcpy(rec + 24, &idv, 4);
    }
  }
  // exact_halves -k, ranked form: halves of >= 10 bases, key = their last ten.  Its partner test needs the partner's
  // window inside the 48 stream bases a queue entry carries (half_partner_fast); a half whose window does not fit
  // (patterns of 31, 32 characters) passes unfiltered, which is fine for a few of them -- a set made of such
  // patterns stays on the round-1 form, whose filter reads the stream itself.
  size_t half_fits = 0;
  if (partners)
    for (size_t j = 0; j < np; ++j) {
      const int L = (int)pats[j].s.size(), plen = t.part_len[j];
      if (t.part_side[j] ? L + plen + halves_k <= 32 : plen + halves_k <= 16) ++half_fits;
    }
  if (partners && np > 0 && lmin >= 10 && half_fits * 10 >= np * 9 && !knobs.half_bloom) {
    std::vector<uint64_t> srt(np);
    for (size_t j = 0; j < np; ++j) {
      const std::string &s = pats[j].s;
      const int L = (int)s.size();
      uint32_t key = 0;
      for (int i = 0; i < 10; ++i) key |= (uint32_t)base2((unsigned char)s[L - 10 + i]) << (2 * i);
      const uint32_t bitpos = ((key & 0x7fffu) << 5) | (key >> 15);   // word-major position of the key's bit (pm_pair.hip layout)
      srt[j] = ((uint64_t)bitpos << 32) | (uint64_t)j;
    }
    std::sort(srt.begin(), srt.end());
    t.hr_image.assign(HR_IMAGE_WORDS, 0);
    t.hr_order.resize(np);
    for (size_t j = 0; j < np;) {
      const uint32_t bitpos = (uint32_t)(srt[j] >> 32);
      size_t j2 = j;
      while (j2 < np && (uint32_t)(srt[j2] >> 32) == bitpos) ++j2;
      t.hr_image[bitpos >> 5] |= 1u << (bitpos & 31u);
      t.hr_first.push_back((uint32_t)j);
      const uint32_t more_at = (uint32_t)t.hr_more.size();
      for (size_t q = j; q < j2; ++q) {
        const uint32_t pi = (uint32_t)srt[q];
        t.hr_order[q] = pi;
        const uint64_t rec = (uint64_t)t.part32[pi] | ((uint64_t)half_slot_info(t.part_len[pi], t.part_side[pi], (int)pats[pi].s.size(), q == j ? (uint32_t)(j2 - j - 1) : 0u, q == j ? more_at : 0u) << 32);
        if (q == j) t.hr_slots.push_back(rec); else t.hr_more.push_back(rec);
      }
      j = j2;
    }
    t.hr_first.push_back((uint32_t)np);
    if (t.hr_more.size() >= ((size_t)1 << 20)) { t.hr_image.clear(); t.hr_slots.clear(); t.hr_more.clear(); t.hr_first.clear(); t.hr_order.clear(); }   // (index field of the slot: 20 bits)
    else {
      uint32_t *img = t.hr_image.data();
      uint32_t *sup = img + HR_BITMAP_WORDS;
      uint16_t *rel = reinterpret_cast<uint16_t *>(img + HR_BITMAP_WORDS + HR_SUPER);
      uint32_t run = 0;
      for (int sb = 0; sb < HR_SUPER; ++sb) {          // set bits before every 2048-bit superblock (u32) and, inside it, before every 256-bit block (u16)
        sup[sb] = run;
        uint32_t in = 0;
        for (int b = 0; b < 8; ++b) {
          rel[sb * 8 + b] = (uint16_t)in;
          for (int w = 0; w < 8; ++w) in += (uint32_t)__builtin_popcount(img[sb * 64 + b * 8 + w]);
        }
        run += in;
      }
    }
  }
  // filter, second-level bitmap and bucket table per combo: the combos' tables are disjoint, and
  // the inserts are cache misses into megabytes of table, so one thread per combo
  auto build_combo = [&](int ci) {
    uint64_t cm = 0;
    for (int q = 0; q < t.r; ++q) cm |= pmask << (2 * t.pb * t.combos[ci][q]);
    const uint32_t mlo = (uint32_t)cm, mhi = (uint32_t)(cm >> 32);
    for (size_t j = 0; j < np; ++j) {
      const uint32_t wlo = t.pat40[j].lo, whi = t.pat40[j].hi;
      uint32_t ss = 0;
      const uint32_t h = t.mode == 0 ? window_hash<0>(wlo, whi, mlo, mhi, 0, &ss)
                       : t.mode == 1 ? window_hash<1>(wlo, whi, mlo, mhi, t.perm_sel[ci], &ss)
                                     : window_hash<2>(wlo, whi, mlo, mhi, t.perm_sel[ci], &ss);
      const uint32_t hsel = bloom_selectors(ss);
      static_assert(SEED_BLOOM_WORDS == 1 << 15, "block address = h >> 15");
      if (!t.etable.empty()) {
        // pm_edit_scan: tabulated key hash, filter block addressed by the hash itself, key map by its top bits
        const uint64_t W = ((uint64_t)whi << 32) | wlo;
        const uint32_t H = edit_piece_hash((uint32_t)(W >> (8 * t.combos[ci][0])) & 0xffu, EDIT_MUL0) ^
                           edit_piece_hash((uint32_t)(W >> (8 * t.combos[ci][1])) & 0xffu, EDIT_MUL1) ^
                           edit_piece_hash((uint32_t)(W >> (8 * t.combos[ci][2])) & 0xffu, EDIT_MUL2);
        const int jshift = 32 - t.etable_log + 3;         // H >> jshift = byte offset of the key's dword in the map
        t.bloom[(size_t)ci * SEED_BLOOM_STRIDE + ((H & 0x1fffcu) >> 2)] |= (1u << ((H >> 16) & 31)) | (1u << ((H >> 24) & 31)) | (1u << ((H >> (jshift + 8)) & 31));
        const uint32_t dw = H >> (jshift + 2), bit = (H >> 8) & 31u;
        t.etable[((size_t)ci << (t.etable_log - 3)) + (size_t)dw * 4 + (bit >> 3)] |= (uint8_t)(1u << (bit & 7u));
      } else if (t.exact_filter) {
        const uint32_t key = wlo & mlo;                           // < 2^20 = the filter's bit count
        t.bloom[(size_t)ci * SEED_BLOOM_STRIDE + (key >> 5)] |= 1u << (key & 31u);
      } else {
        // the key's block: the dword at byte bloom_addr(h) (bloom_block on the device)
        const uint32_t bits = (1u << ((hsel >> 8) & 31)) | (1u << ((hsel >> 16) & 31)) | (1u << ((hsel >> 24) & 31));
        uint8_t *blk = reinterpret_cast<uint8_t *>(&t.bloom[(size_t)ci * SEED_BLOOM_STRIDE]) + bloom_addr(h);
        for (int q = 0; q < 4; ++q) blk[q] |= (uint8_t)(bits >> (8 * q));
      }
      const uint32_t h2 = h * HASH_SLOT;
      if (t.etable.empty()) t.bitmap2[((size_t)ci << (lb2 - 5)) + (h2 >> (37 - lb2))] |= 1u << ((h2 >> (32 - lb2)) & 31);
      const uint32_t imask = (1u << idx_bits) - 1u;
      uint32_t b = h2 >> t.bucket_shift;
      uint32_t *tb = &t.slots[(size_t)ci * nslots];
      for (;;) {                                   // first bucket with a free slot; probes stop at such a bucket
        int q = 0;
        while (q < (int)bslots && tb[(size_t)b * bslots + q] != EMPTY) ++q;
        if (q < (int)bslots) {
          if (!t.etable.empty()) {                 // pm_edit_scan: the pattern's two pieces outside the key | 16-bit fingerprint; index in a table of its own
            const uint64_t W = ((uint64_t)whi << 32) | wlo;
            uint32_t other = 0;
            int nt = 0;
            for (int piece = 0; piece < 5; ++piece)
              if (piece != t.combos[ci][0] && piece != t.combos[ci][1] && piece != t.combos[ci][2]) other |= ((uint32_t)(W >> (8 * piece)) & 0xffu) << (8 * nt++);
            const uint32_t f = h2 & 0xffffu;
            tb[(size_t)b * bslots + q] = (other << 16) | (f == 0xffffu ? 0xfffeu : f);
            t.eidx[(size_t)ci * nslots + (size_t)b * bslots + q] = (uint32_t)j;
          } else tb[(size_t)b * bslots + q] = ((h2 << idx_bits) & ~imask) | (uint32_t)j;
          break;
        }
        b = (b + 1) & (uint32_t)(nbuckets - 1);
      }
    }
  };
  if (np * (size_t)C < 20000) {
    for (int ci = 0; ci < C; ++ci) build_combo(ci);
  } else {
    const int T = (int)std::min<unsigned>((unsigned)C, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    for (int w = 0; w < T; ++w) th.emplace_back([&, w]() { for (int ci = w; ci < C; ci += T) build_combo(ci); });
    for (std::thread &x : th) x.join();
  }
  return "";
}

hipError_t seed_upload(const SeedTables &t, SeedDevice *d, hipStream_t st) {
  seed_free(d);
  d->k = t.k; d->Lw = t.Lw; d->pb = t.pb; d->r = t.r; d->ascii = t.ascii; d->maxlen = t.maxlen;
  d->ncombos = (int)t.combos.size(); d->nslots = t.nslots; d->idx_bits = t.idx_bits; d->bucket_shift = t.bucket_shift;
  for (int c = 0; c < d->ncombos; ++c) {
    uint64_t cm = 0;
    for (int q = 0; q < t.r; ++q) cm |= ((1ull << (2 * t.pb)) - 1ull) << (2 * t.pb * t.combos[c][q]);
    d->mask_lo[c] = (uint32_t)cm; d->mask_hi[c] = (uint32_t)(cm >> 32);
    d->perm_sel[c] = t.perm_sel[c];
  }
  d->mode = t.mode;
  auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
  };
  hipError_t e;
  if ((e = up(t.bloom.data(), t.bloom.size() * 4, (void **)&d->bloom)) != hipSuccess) return e;
  if ((e = up(t.slots.data(), t.slots.size() * 4, (void **)&d->slots)) != hipSuccess) return e;
  if ((e = up(t.bitmap2.data(), t.bitmap2.size() * 4, (void **)&d->bitmap2)) != hipSuccess) return e;
  d->lb2 = t.lb2;
  if ((e = up(t.pat40.data(), t.pat40.size() * 8, (void **)&d->pat40)) != hipSuccess) return e;
  if ((e = up(t.pat_len.data(), t.pat_len.size(), (void **)&d->pat_len)) != hipSuccess) return e;
  if ((e = up(t.pat_id.data(), t.pat_id.size() * 4, (void **)&d->pat_id)) != hipSuccess) return e;
  if ((e = up(t.pat_codes.data(), t.pat_codes.size(), (void **)&d->pat_codes)) != hipSuccess) return e;
  if ((e = up(t.cmap, 256, (void **)&d->cmap)) != hipSuccess) return e;
  if ((e = up(t.part32.data(), t.part32.size() * 4, (void **)&d->part32)) != hipSuccess) return e;
  if ((e = up(t.part_len.data(), t.part_len.size(), (void **)&d->part_len)) != hipSuccess) return e;
  if ((e = up(t.part_side.data(), t.part_side.size(), (void **)&d->part_side)) != hipSuccess) return e;
  d->halves = t.halves; d->hk = t.hk; d->hfast = t.hfast; d->eos_code = t.eos_code; d->exact_filter = t.exact_filter;
  d->half_ranked = !t.hr_image.empty();
  if (d->half_ranked) {
    if ((e = up(t.hr_image.data(), t.hr_image.size() * 4, (void **)&d->hr_image)) != hipSuccess) return e;
    if ((e = up(t.hr_slots.data(), t.hr_slots.size() * 8, (void **)&d->hr_slots)) != hipSuccess) return e;
    if ((e = up(t.hr_more.data(), t.hr_more.size() * 8, (void **)&d->hr_more)) != hipSuccess) return e;
    if ((e = up(t.hr_first.data(), t.hr_first.size() * 4, (void **)&d->hr_first)) != hipSuccess) return e;
    if ((e = up(t.hr_order.data(), t.hr_order.size() * 4, (void **)&d->hr_order)) != hipSuccess) return e;
  }
  d->edits = t.edits;
  d->edit_tabulated = !t.etable.empty(); d->etable_log = t.etable_log;
  if (d->edit_tabulated && (e = up(t.etable.data(), t.etable.size(), (void **)&d->etable)) != hipSuccess) return e;
  if (d->edit_tabulated && (e = up(t.eidx.data(), t.eidx.size() * 4, (void **)&d->eidx)) != hipSuccess) return e;
  for (int c = 0; c < d->ncombos; ++c) {              // byte masks of the combo's first and second piece (edits: displaced pieces)
    d->emask_a[c] = t.r >= 3 && t.combos[c][0] < 4 ? 0xffu << (8 * t.combos[c][0]) : 0u;
    d->emask_b[c] = t.r >= 3 && t.combos[c][1] < 4 ? 0xffu << (8 * t.combos[c][1]) : 0u;
  }
  if (t.edits) edit_cover(t.edits, t.k + t.r, t.combos, d->evar);
  if (d->edit_tabulated)                              // pm_edit_scan's instances are compiled for these displacement lists
    for (int c = 0; c < d->ncombos; ++c)
      if (t.r != 3 || d->evar[c] == 0 || d->evar[c] != edit_cover_of(t.combos[c][0], t.combos[c][1], t.combos[c][2])) return hipErrorInvalidConfiguration;
  if ((e = hipMalloc(&d->d_args, 1024)) != hipSuccess) return e;
  static_assert(sizeof(SeedArgs) <= 1024, "argument block");
  const void *kernels[] = {reinterpret_cast<const void *>(pm_seed_scan<20, 1, false>), reinterpret_cast<const void *>(pm_seed_scan<20, 2, false>),
                           reinterpret_cast<const void *>(pm_seed_scan<20, 0, false>), reinterpret_cast<const void *>(pm_seed_scan<0, 0, false>),
                           reinterpret_cast<const void *>(pm_seed_scan<0, 0, true>),
                           reinterpret_cast<const void *>(pm_seed_scan<20, 1, false, true>), reinterpret_cast<const void *>(pm_edit_scan),
                           reinterpret_cast<const void *>(pm_half_scan)};
  for (const void *kf : kernels) {
    // bloom_block addresses the filter from LDS address 0: a kernel that acquired static LDS (which
    // the dynamic block would follow) must fail here, at init, not compute with a shifted filter
    hipFuncAttributes fa;
    if ((e = hipFuncGetAttributes(&fa, kf)) != hipSuccess) return e;
    if (fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
    if ((e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, SEED_LDS_BYTES)) != hipSuccess) return e;
  }
  return hipStreamSynchronize(st);
}

void seed_free(SeedDevice *d) {
  void *ptrs[] = {d->hr_image, d->hr_slots, d->hr_more, d->hr_first, d->hr_order, d->etable, d->eidx, d->part32, d->part_len, d->part_side, d->d_args, d->bloom, d->slots, d->bitmap2, d->pat40, d->pat_len, d->pat_id, d->pat_codes, d->cmap};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  *d = SeedDevice();
}

ScanGeometry seed_geometry(const SeedDevice &d, int64_t begin, int64_t end) {
  ScanGeometry g;
  int64_t chunk = 1 << 19;                                         // 512 Ki positions per workgroup
  // ten-combo substitution plans: 2 Mi positions per workgroup amortise the 128 KiB filter staging and
  // the pipeline fill/drain over four times the work (-K 2, 3 Gbp: 26.6 -> 25.2 ms) while the grid
  // still has dozens of workgroups per CU; the one- and four-combo plans and the edit-distance plan
  // measured no gain or a loss (sweep in DESIGN.md section 7)
  if (!d.edits && !d.halves && d.ncombos >= 8 && (end - begin) / ((int64_t)1 << 21) * d.ncombos >= 256 * 16) chunk = (int64_t)1 << 21;
  if (d.knobs.seed_chunk >= 1024 * WAVES) chunk = d.knobs.seed_chunk / (1024 * WAVES) * (1024 * WAVES);   // test knob
  g.seg_len = chunk;
  if (d.edits) { begin = begin > d.edits ? begin - d.edits : 0; end += d.edits; }   // seeds up to k positions outside (begin, end]
  const int64_t c_lo = begin / chunk, c_hi = end > begin ? (end - 1) / chunk : c_lo - 1;
  g.nseg = (int)(c_hi - c_lo + 1);
  g.threads = SEED_THREADS;
  g.blocks = g.nseg * d.ncombos;
  return g;
}

hipError_t seed_launch(const SeedDevice &d, const uint8_t *d_text, const uint32_t *d_packed, int64_t n, int64_t begin, int64_t end,
                       pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, hipStream_t st,
                       ScanGeometry *geo_out, const EditStage *es) {
  if (!d_packed) return hipErrorInvalidValue;
  if (end > n) end = n;
  ScanGeometry g = seed_geometry(d, begin, end);
  if (geo_out) *geo_out = g;
  if (g.blocks <= 0 || d.nslots == 0) return hipSuccess;
  SeedArgs a;
  a.text = d_text; a.n = n; a.begin = begin; a.end = end;
  a.packed = d_packed; a.npacked = (n + 15) / 16;
  a.chunk_len = g.seg_len; a.chunk0 = (d.edits && begin > d.edits ? begin - d.edits : (d.edits ? 0 : begin)) / g.seg_len; a.nchunks = g.nseg; a.ncombos = d.ncombos;
  a.group = 256;                                                   // one run ~ one chunk per CU
  if (d.edits && d.edit_tabulated) a.group = 512;                   // pm_edit_scan (3 Gbp, 100k primers): 128: 86.8 ms, 256: 82.2, 512: 81.3, 1024: 81.1, 2048: 83.9
  if (d.knobs.seed_group > 0) a.group = d.knobs.seed_group;
  a.k = d.k; a.Lw = d.Lw; a.pb = d.pb; a.r = d.r; a.ascii = d.ascii ? 1 : 0;
  a.debug = d.knobs.seed_debug;
  memcpy(a.mask_lo, d.mask_lo, sizeof(a.mask_lo));
  memcpy(a.mask_hi, d.mask_hi, sizeof(a.mask_hi));
  memcpy(a.perm_sel, d.perm_sel, sizeof(a.perm_sel));
  a.bloom = d.bloom; a.buckets = reinterpret_cast<const uint4 *>(d.slots); a.bucket_shift = (uint32_t)d.bucket_shift; a.idx_bits = (uint32_t)d.idx_bits;
  a.bitmap2 = d.bitmap2; a.lb2 = (uint32_t)d.lb2;
  a.halves = d.halves ? 1 : 0; a.hk = d.hk; a.hfast = d.hfast; a.eos_code = d.eos_code; a.exact_filter = d.exact_filter ? 1 : 0;
  a.edits = d.edits; a.maxlen = d.maxlen;
  memcpy(a.emask_a, d.emask_a, sizeof(a.emask_a)); memcpy(a.emask_b, d.emask_b, sizeof(a.emask_b)); memcpy(a.evar, d.evar, sizeof(a.evar)); a.part32 = d.part32; a.part_len = d.part_len; a.part_side = d.part_side;
  a.pat40 = reinterpret_cast<const uint2 *>(d.pat40); a.pat_len = d.pat_len; a.pat_id = d.pat_id;
  a.pat_codes = d.pat_codes; a.cmap = d.cmap; a.out = d_out; a.counter = d_counter; a.cap = cap;
  a.etable = nullptr; a.eidx = nullptr; a.seed_out = nullptr; a.et_shift = 0; a.et_bytes_log = 0; a.et_mask = 0;
  a.hr_image = d.hr_image; a.hr_slots = reinterpret_cast<const uint2 *>(d.hr_slots); a.hr_more = reinterpret_cast<const uint2 *>(d.hr_more); a.hr_first = d.hr_first; a.hr_order = d.hr_order;
  // the rare out-of-line paths read their parameters from a device copy of the argument block
  if (!d.d_args) return hipErrorInvalidValue;
  a.self = reinterpret_cast<const SeedArgs *>(d.d_args);
  hipError_t ce = hipMemcpyAsync(d.d_args, &a, sizeof(a), hipMemcpyHostToDevice, st);
  if (ce != hipSuccess) return ce;
  const dim3 grid(g.blocks), block(SEED_THREADS);
  if (d.edits) {
    // two kernels: the scan writes seed records (pattern, position) that passed the three-base-word
    // test into es->d_seeds; pm_edits_verify runs the automaton over that dense list and writes
    // the candidates to d_out.  *es->d_seed_count is zeroed by the caller (stream order).
    if (!es || !es->d_seeds || !es->d_seed_count) return hipErrorInvalidValue;
    SeedArgs sa = a;
    sa.seed_out = es->d_seeds; sa.counter = es->d_seed_count; sa.cap = es->seed_cap;
    sa.etable = d.etable; sa.eidx = d.eidx; sa.et_shift = 32 - d.etable_log + 3; sa.et_mask = ((1u << (d.etable_log - 3)) - 1u) & ~3u; sa.et_bytes_log = d.etable_log - 3;
    // the rare out-of-line paths read the scan's view of the argument block
    ce = hipMemcpyAsync(d.d_args, &sa, sizeof(sa), hipMemcpyHostToDevice, st);
    if (ce != hipSuccess) return ce;
    if (es->skip_scan) {}
    else if (d.edit_tabulated) hipLaunchKernelGGL(pm_edit_scan, grid, block, SEED_LDS_BYTES, st, sa);
    else hipLaunchKernelGGL((pm_seed_scan<20, 1, false, true>), grid, block, SEED_LDS_BYTES, st, sa);
    if ((ce = hipGetLastError()) != hipSuccess) return ce;
    EditVerifyArgs v;
    v.a = a;
    if (d.edit_tabulated && !es->skip_scan) v.a.eidx = d.eidx;
    v.seeds = es->d_seeds; v.nseeds = es->d_seed_count; v.seed_cap = es->seed_cap;
    if (es->bases) {
      BasesArgs b;
      b.codes = es->b_codes; b.len = es->b_len; b.esb = es->b_esb; b.eeb = es->b_eeb; b.own_lo = es->own_lo; b.own_hi = es->own_hi;
      hipLaunchKernelGGL(pm_bases_verify, dim3(256 * 16), dim3(256), 0, st, v, b);
    } else
    hipLaunchKernelGGL(pm_edits_verify, dim3(256 * 16), dim3(256), 0, st, v);
  }
  else if (d.halves && d.half_ranked) {
    // two kernels: pm_half_scan writes (rank of the key, position) records of the key hits whose partner test passes
    // into es->d_seeds, pm_half_verify resolves them on the raw stream into seed records.  *es->d_seed_count zeroed by the caller.
    if (!es || !es->d_seeds || !es->d_seed_count) return hipErrorInvalidValue;
    SeedArgs sa = a;
    sa.seed_out = es->d_seeds; sa.counter = es->d_seed_count; sa.cap = es->seed_cap;
    hipLaunchKernelGGL(pm_half_scan, dim3(g.nseg), block, SEED_LDS_BYTES, st, sa);
    if ((ce = hipGetLastError()) != hipSuccess) return ce;
    EditVerifyArgs v;
    v.a = a;
    v.seeds = es->d_seeds; v.nseeds = es->d_seed_count; v.seed_cap = es->seed_cap;
    hipLaunchKernelGGL(pm_half_verify, dim3(256 * 16), dim3(256), 0, st, v);
  }
  else if (d.halves) hipLaunchKernelGGL((pm_seed_scan<0, 0, true>), grid, block, SEED_LDS_BYTES, st, a);
  else if (d.Lw == 20 && d.mode == 1) hipLaunchKernelGGL((pm_seed_scan<20, 1, false>), grid, block, SEED_LDS_BYTES, st, a);
  else if (d.Lw == 20 && d.mode == 2) hipLaunchKernelGGL((pm_seed_scan<20, 2, false>), grid, block, SEED_LDS_BYTES, st, a);
  else if (d.Lw == 20) hipLaunchKernelGGL((pm_seed_scan<20, 0, false>), grid, block, SEED_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((pm_seed_scan<0, 0, false>), grid, block, SEED_LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace pm

// pm_bitpar.hip -- bit-parallel Shift-And / k-error Shift-And scan kernels for gfx950.
//
// What it computes (reference shift_and.cc:208-255, shift_and_inexact.cc:249-352): for every
// stream position, advance (k+1) bit rows per pattern and report (position, pattern, level)
// whenever a pattern's last bit is set in row k.  In the reference all patterns sit in one long
// bit string that is updated word by word for every character; the only coupling between
// neighbouring patterns is a carry that the first-bit mask `s` overrides, so patterns are
// independent.  That is what this kernel exploits:
//
//   * a LANE owns a 256-bit string (BP_WPL = 8 dwords) into which whole patterns are packed, all
//     rows and all masks of that string live in VGPRs (no LDS, no memory traffic in the loop);
//   * a WAVE (64 lanes = one "tile" of patterns) walks one segment of the stream; the character
//     is wave-uniform, so the per-character mask choice is a scalar branch, not a gather;
//   * the stream is read once per 256 characters with one coalesced dword load per lane; the
//     class codes are then handed out lane by lane with v_readlane (no LDS round trip);
//   * the grid is (segments x tiles) waves, tile-major, so waves that run together read the same
//     stream bytes (L2/MALL hits); a segment starts `halo` = maxlen+k characters early from an
//     empty state, which reproduces the serial automaton exactly for every position it owns;
//   * hits are rare: one v_cmp per character detects them, the decode loop ranks the hit bit
//     among the lane's last-bits with popcounts to find the pattern and appends a 16-byte record
//     through one global atomic counter.
//
// Cost model: 4 VALU ops per (dword, character) exact, 10 for k=2 substitutions, 19 for k=2
// edits -> integer-ALU bound (DESIGN.md "bitpar roofline"); HBM traffic is negligible.
#include "pm_internal.h"
#include "pm_iupac.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace pm {

namespace {

struct BitparArgs {
  const uint8_t *text;
  int64_t n, begin, end;        // owned hit ends: begin < end_pos <= end
  int64_t seg_len, seg0;
  int nseg, ntiles, halo;
  int nstr;                     // text-parallel form: pattern strings (lanes of the tables) in use
  int64_t sub_len;              // text-parallel form: stream characters per lane (seg_len = 64 * sub_len)
  const uint32_t *U, *S, *LAST, *INIT, *lane_first, *pid_of;
  const uint8_t *cmap;
  pm_hit *out;
  unsigned long long *counter;
  unsigned long long cap;
};

constexpr int W = BP_WPL;

template <int K>
struct Rows { uint32_t r[K + 1][W]; };

__device__ __forceinline__ uint32_t shl1(const uint32_t (&x)[W], const uint32_t (&s)[W], int w) {
  // (X << 1 | carry from the word below) | first-bit mask   (shift_and.cc:219-222)
  return w == 0 ? ((x[0] << 1) | s[0]) : (__builtin_amdgcn_alignbit(x[w], x[w - 1], 31) | s[w]);
}

// One character.  MASKED: the character's class has a mask row `u`; otherwise u == 0.
// NOTEOS=false is the end-of-sequence code: every row collapses to sh(R)&u (no error terms,
// shift_and_inexact.cc:293), which is 0 because no pattern contains the EOS code.
template <int K, bool INDELS, bool MASKED>
__device__ __forceinline__ void step(Rows<K> &R, const uint32_t (&u)[W], const uint32_t (&s)[W]) {
  uint32_t m1[W];   // what row l-1 hands up to row l ("m1" in the reference)
#pragma unroll
  for (int w = W - 1; w >= 0; --w) {
    const uint32_t o = R.r[0][w];
    const uint32_t x = shl1(R.r[0], s, w);
    m1[w] = INDELS ? (x | o) : x;
    R.r[0][w] = MASKED ? (x & u[w]) : 0u;
  }
#pragma unroll
  for (int l = 1; l <= K; ++l) {
#pragma unroll
    for (int w = W - 1; w >= 0; --w) {
      const uint32_t o = R.r[l][w];
      const uint32_t x = shl1(R.r[l], s, w);
      uint32_t nv = MASKED ? ((x & u[w]) | m1[w]) : m1[w];
      if (INDELS) nv |= shl1(R.r[l - 1], s, w) | R.r[l - 1][w];   // row l-1 already holds its new value
      m1[w] = INDELS ? (x | o) : x;
      R.r[l][w] = nv;
    }
  }
}

template <int K>
__device__ __forceinline__ void clear_rows(Rows<K> &R) {
#pragma unroll
  for (int l = 0; l <= K; ++l)
#pragma unroll
    for (int w = 0; w < W; ++w) R.r[l][w] = 0u;
}

template <int K, bool INDELS>
__global__ __launch_bounds__(256) void pm_bitpar_scan(BitparArgs a) {
  __shared__ uint8_t s_cmap[256];
  s_cmap[threadIdx.x] = a.cmap[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= (long)a.nseg * a.ntiles) return;          // wave-uniform
  const int tile = (int)(wv % a.ntiles);
  const int64_t seg = a.seg0 + wv / a.ntiles;

  const int64_t pos0 = seg * a.seg_len;
  const int64_t own_lo = pos0 > a.begin ? pos0 : a.begin;               // index of last matched char
  int64_t own_hi = pos0 + a.seg_len;
  if (own_hi > a.end) own_hi = a.end;
  if (own_hi > a.n) own_hi = a.n;
  if (own_lo >= own_hi) return;
  int64_t start = pos0 - a.halo;

  uint32_t u[BP_NC][W], s[W], last[W];
#pragma unroll
  for (int w = 0; w < W; ++w) {
#pragma unroll
    for (int c = 0; c < BP_NC; ++c) u[c][w] = a.U[(((size_t)tile * BP_NC + c) * W + w) * 64 + lane];
    s[w] = a.S[((size_t)tile * W + w) * 64 + lane];
    last[w] = a.LAST[((size_t)tile * W + w) * 64 + lane];
  }
  Rows<K> R;
  clear_rows<K>(R);
  if (start <= 0) {              // true start of the stream: rows l>=1 begin with l prefix bits
    start = 0;                   // (shift_and_inexact.cc:162-164)
#pragma unroll
    for (int l = 1; l <= K; ++l)
#pragma unroll
      for (int w = 0; w < W; ++w) R.r[l][w] = a.INIT[(((size_t)tile * (K > 0 ? K : 1) + (l - 1)) * W + w) * 64 + lane];
  }
  const uint32_t lane_base = a.lane_first[(size_t)tile * 64 + lane];
  const uint32_t zero[W] = {0, 0, 0, 0, 0, 0, 0, 0};

  for (int64_t bb = start; bb < own_hi; bb += BP_BLOCK) {
    // one coalesced dword per lane = 256 stream bytes per wave, mapped to class codes
    const int64_t off = bb + 4 * lane;
    uint32_t raw;
    if (off + 3 < a.n) raw = *reinterpret_cast<const uint32_t *>(a.text + off);
    else {
      raw = 0;
      for (int b = 0; b < 4; ++b) raw |= (uint32_t)(off + b < a.n ? a.text[off + b] : 0) << (8 * b);
    }
    const uint32_t vc = (uint32_t)s_cmap[raw & 0xff] | ((uint32_t)s_cmap[(raw >> 8) & 0xff] << 8) |
                        ((uint32_t)s_cmap[(raw >> 16) & 0xff] << 16) | ((uint32_t)s_cmap[raw >> 24] << 24);
    const int nb = (int)((own_hi - bb) < BP_BLOCK ? (own_hi - bb) : BP_BLOCK);
    for (int j = 0; 4 * j < nb; ++j) {
      const uint32_t c4 = __builtin_amdgcn_readlane(vc, j);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (4 * j + b >= nb) break;
        const uint32_t cls = (c4 >> (8 * b)) & 0xffu;   // wave-uniform
        switch (cls) {
          case 0: step<K, INDELS, true>(R, u[0], s); break;
          case 1: step<K, INDELS, true>(R, u[1], s); break;
          case 2: step<K, INDELS, true>(R, u[2], s); break;
          case 3: step<K, INDELS, true>(R, u[3], s); break;
          case 4: step<K, INDELS, true>(R, u[4], s); break;
          case 5: step<K, INDELS, true>(R, u[5], s); break;
          case BP_NC: step<K, INDELS, false>(R, zero, s); break;
          default: clear_rows<K>(R); break;               // EOS code
        }
        uint32_t hit = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) hit |= R.r[K][w] & last[w];
        if (hit != 0) {
          const int64_t t = bb + 4 * j + b;
          if (t >= own_lo) {
            uint32_t running = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              uint32_t hb = R.r[K][w] & last[w];
              while (hb) {
                const int bit = __ffs(hb) - 1;
                hb &= hb - 1;
                const uint32_t rank = running + __popc(last[w] & ((1u << bit) - 1u));
                int lvl = K;                               // shift_and_inexact.cc:323-328
#pragma unroll
                for (int l = K - 1; l >= 0; --l) {
                  if (lvl == l + 1 && ((R.r[l][w] >> bit) & 1u)) lvl = l;
                }
                const unsigned long long idx = atomicAdd(a.counter, 1ull);
                if (idx < a.cap) {
                  pm_hit h;
                  h.end = t + 1; h.pid = a.pid_of[lane_base + rank]; h.k = (uint8_t)lvl;
                  h.aux[0] = h.aux[1] = h.aux[2] = 0;
                  a.out[idx] = h;
                }
              }
              running += __popc(last[w]);
            }
          }
        }
      }
    }
  }
}

// Text-parallel form for small pattern sets.  pm_bitpar_scan gives every lane of a wave its own
// 256-bit string of patterns and all 64 the same stream character; a dozen patterns then keep one
// lane of 64 busy and a pass over the stream has a floor of ~150 ms per Gbp.  Here the roles are
// swapped: every lane of a wave runs the SAME pattern string (masks loaded as broadcasts) over its
// OWN run of sub_len stream characters (+ halo in front to warm the state up), the character
// class picks the mask row per lane (select chain instead of a scalar branch).  One wave per
// (64 runs, pattern string); used while the set has few strings (bitpar_launch).
template <int K, bool INDELS>
__global__ __launch_bounds__(256) void pm_bitpar_scan_tp(BitparArgs a) {
  __shared__ uint8_t s_cmap[256];
  s_cmap[threadIdx.x] = a.cmap[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wv >= (long)a.nseg * a.nstr) return;            // wave-uniform
  const int str = (int)(wv % a.nstr);
  const int tile = str >> 6, sl = str & 63;
  const int64_t seg = a.seg0 + wv / a.nstr;

  const int64_t pos0 = seg * a.seg_len + (int64_t)lane * a.sub_len;     // multiple of 16
  const int64_t own_lo = pos0 > a.begin ? pos0 : a.begin;
  int64_t own_hi = pos0 + a.sub_len;
  if (own_hi > a.end) own_hi = a.end;
  if (own_hi > a.n) own_hi = a.n;
  int64_t start = pos0 - a.halo;
  const bool at0 = start <= 0;                         // this lane's run begins at the true start of the stream
  if (at0) start = 0;
  const int64_t nchar = own_lo < own_hi ? own_hi - start : 0;

  uint32_t u[BP_NC][W], s[W], last[W];
#pragma unroll
  for (int w = 0; w < W; ++w) {
#pragma unroll
    for (int c = 0; c < BP_NC; ++c) u[c][w] = a.U[(((size_t)tile * BP_NC + c) * W + w) * 64 + sl];
    s[w] = a.S[((size_t)tile * W + w) * 64 + sl];
    last[w] = a.LAST[((size_t)tile * W + w) * 64 + sl];
  }
  Rows<K> R;
  clear_rows<K>(R);
#pragma unroll
  for (int l = 1; l <= K; ++l)                          // rows l>=1 begin with l prefix bits (shift_and_inexact.cc:162-164)
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const uint32_t iv = a.INIT[(((size_t)tile * (K > 0 ? K : 1) + (l - 1)) * W + w) * 64 + sl];
      R.r[l][w] = at0 ? iv : 0u;
    }
  const uint32_t str_base = a.lane_first[(size_t)tile * 64 + sl];

  const int64_t upper = a.sub_len + a.halo;
  for (int64_t o = 0; o < upper; o += 16) {
    if (__ballot(o < nchar) == 0) break;               // every lane of the wave is through
    const int64_t off = start + o;
    uint32_t v[4] = {0, 0, 0, 0};
    if (o < nchar) {
      if (off + 16 <= a.n) __builtin_memcpy(v, a.text + off, 16);
      else for (int b = 0; b < 16; ++b) if (off + b < a.n) v[b >> 2] |= (uint32_t)a.text[off + b] << (8 * (b & 3));
    }
#pragma unroll 1
    for (int b = 0; b < 16; ++b) {
      const uint32_t word = b < 4 ? v[0] : (b < 8 ? v[1] : (b < 12 ? v[2] : v[3]));
      const uint32_t cls = s_cmap[(word >> (8 * (b & 3))) & 0xffu];
      uint32_t uc[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        uint32_t x = 0u;
#pragma unroll
        for (int c = 0; c < BP_NC; ++c) x = cls == (uint32_t)c ? u[c][w] : x;
        uc[w] = x;                                       // class BP_NC (no pattern has it) and EOS: no mask row
      }
      step<K, INDELS, true>(R, uc, s);
      if (cls > (uint32_t)BP_NC) clear_rows<K>(R);       // EOS code
      uint32_t hit = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) hit |= R.r[K][w] & last[w];
      const int64_t t = off + b;
      if (hit != 0 && o + b < nchar && t >= own_lo) {
        uint32_t running = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          uint32_t hb = R.r[K][w] & last[w];
          while (hb) {
            const int bit = __ffs(hb) - 1;
            hb &= hb - 1;
            const uint32_t rank = running + __popc(last[w] & ((1u << bit) - 1u));
            int lvl = K;                               // shift_and_inexact.cc:323-328
#pragma unroll
            for (int l = K - 1; l >= 0; --l) {
              if (lvl == l + 1 && ((R.r[l][w] >> bit) & 1u)) lvl = l;
            }
            const unsigned long long idx = atomicAdd(a.counter, 1ull);
            if (idx < a.cap) {
              pm_hit h;
              h.end = t + 1; h.pid = a.pid_of[str_base + rank]; h.k = (uint8_t)lvl;
              h.aux[0] = h.aux[1] = h.aux[2] = 0;
              a.out[idx] = h;
            }
          }
          running += __popc(last[w]);
        }
      }
    }
  }
}

template <int K, bool INDELS>
hipError_t launch_tp(const BitparArgs &a, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((pm_bitpar_scan_tp<K, INDELS>), dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int K, bool INDELS>
hipError_t launch_t(const BitparArgs &a, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((pm_bitpar_scan<K, INDELS>), dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace

const char *bitpar_kernel_name(int k, bool indels) {
  static const char *names[2][4] = {
      {"pm_bitpar_scan<0,false>", "pm_bitpar_scan<1,false>", "pm_bitpar_scan<2,false>", "pm_bitpar_scan<3,false>"},
      {"pm_bitpar_scan<0,true>", "pm_bitpar_scan<1,true>", "pm_bitpar_scan<2,true>", "pm_bitpar_scan<3,true>"}};
  return names[indels ? 1 : 0][k < 0 ? 0 : (k > 3 ? 3 : k)];
}

std::string bitpar_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids,
                         const Alphabet &alpha, int k, int eos_code, BitparTables *out, bool wildcards, bool text_n) {
  // stream codes a pattern character accepts: itself, or with -w/-W every letter of its IUPAC
  // compatibility set that exists in the stream alphabet, text N only with -W (shift_and.cc:108-147)
  auto accepted = [&](unsigned char ch, int *codes) -> int {
    int nc = 0;
    const char *set = wildcards ? iupac_compatible_set(ch) : nullptr;
    if (set) {
      for (const char *q = set; *q; ++q) {
        const int code = alpha.nch[(unsigned char)*q];
        if (code >= 0 && code < alpha.size && alpha.present[code] && (*q != 'N' || text_n)) {
          bool dup = false;
          for (int i = 0; i < nc; ++i) dup = dup || codes[i] == code;
          if (!dup) codes[nc++] = code;
        }
      }
    } else {
      const int code = alpha.nch[ch];
      if (code >= 0 && code < alpha.size) codes[nc++] = code;
    }
    return nc;
  };
  BitparTables &t = *out;
  t = BitparTables();
  t.k = k;
  if (k < 0 || k > 3) return "bit-parallel kernels are built for k <= 3";
  // classes = distinct stream codes that some pattern position accepts (shift_and.cc:143-147)
  int cls_of_code[256];
  for (int i = 0; i < 256; ++i) { cls_of_code[i] = -1; t.cmap[i] = BP_NC; }
  for (const Pattern &p : pats) {
    if (p.s.empty()) return "empty pattern";
    if ((int)p.s.size() > 32 * BP_WPL) return "pattern longer than 256 characters";
    t.maxlen = std::max(t.maxlen, (int)p.s.size());
    for (unsigned char ch : p.s) {
      int codes[32];
      const int nc = accepted(ch, codes);                  // none: never matches (shift_and.cc:143: nch < 0)
      for (int q = 0; q < nc; ++q)
        if (cls_of_code[codes[q]] < 0) {
          if (t.nclasses == BP_NC) return "patterns accept more than 6 distinct stream codes";
          cls_of_code[codes[q]] = t.nclasses;
          t.cmap[codes[q]] = (uint8_t)t.nclasses++;
        }
    }
  }
  if (eos_code >= 0 && eos_code < 256) {
    if (cls_of_code[eos_code] >= 0) return "a pattern contains the end-of-sequence character";
    t.cmap[eos_code] = BP_NC + 1;
  }
  // pack whole patterns into 256-bit lane strings, in input order
  struct Slot { int lane, bit; };
  std::vector<Slot> slot(pats.size());
  int lane = 0, bit = 0;
  for (size_t j = 0; j < pats.size(); ++j) {
    const int L = (int)pats[j].s.size();
    if (bit + L > 32 * BP_WPL) { ++lane; bit = 0; }
    slot[j] = {lane, bit};
    bit += L;
  }
  const int nlanes = pats.empty() ? 0 : lane + 1;
  t.nlanes = nlanes;
  t.ntiles = (nlanes + 63) / 64;
  const size_t T = (size_t)t.ntiles;
  t.U.assign(T * BP_NC * W * 64, 0);
  t.S.assign(T * W * 64, 0);
  t.LAST.assign(T * W * 64, 0);
  t.INIT.assign(T * (k > 0 ? k : 1) * W * 64, 0);
  t.lane_first.assign(T * 64 + 1, 0);
  t.pid_of.resize(pats.size());
  size_t j = 0;
  for (int ln = 0; ln < (int)T * 64; ++ln) {
    t.lane_first[ln] = (uint32_t)j;
    const size_t tile = ln / 64, l64 = ln % 64;
    while (j < pats.size() && slot[j].lane == ln) {
      const Pattern &p = pats[j];
      const int L = (int)p.s.size();
      for (int i = 0; i < L; ++i) {
        const int b = slot[j].bit + i, w = b >> 5;
        const uint32_t m = 1u << (b & 31);
        int codes[32];
        const int nc = accepted((unsigned char)p.s[i], codes);
        for (int q = 0; q < nc; ++q) t.U[((tile * BP_NC + cls_of_code[codes[q]]) * W + w) * 64 + l64] |= m;
        if (i == 0) t.S[(tile * W + w) * 64 + l64] |= m;
        if (i == L - 1) t.LAST[(tile * W + w) * 64 + l64] |= m;
        for (int l = i + 1; l <= k; ++l) t.INIT[((tile * k + (l - 1)) * W + w) * 64 + l64] |= m;
      }
      t.pid_of[j] = ids[j];
      ++j;
    }
  }
  t.lane_first[T * 64] = (uint32_t)j;
  return "";
}

hipError_t bitpar_upload(const BitparTables &t, bool indels, BitparDevice *d, hipStream_t st) {
  bitpar_free(d);
  d->ntiles = t.ntiles; d->nlanes = t.nlanes; d->k = t.k; d->maxlen = t.maxlen; d->indels = indels;
  auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 4);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
  };
  hipError_t e;
  if ((e = up(t.U.data(), t.U.size() * 4, (void **)&d->U)) != hipSuccess) return e;
  if ((e = up(t.S.data(), t.S.size() * 4, (void **)&d->S)) != hipSuccess) return e;
  if ((e = up(t.LAST.data(), t.LAST.size() * 4, (void **)&d->LAST)) != hipSuccess) return e;
  if ((e = up(t.INIT.data(), t.INIT.size() * 4, (void **)&d->INIT)) != hipSuccess) return e;
  if ((e = up(t.lane_first.data(), t.lane_first.size() * 4, (void **)&d->lane_first)) != hipSuccess) return e;
  if ((e = up(t.pid_of.data(), t.pid_of.size() * 4, (void **)&d->pid_of)) != hipSuccess) return e;
  if ((e = up(t.cmap, 256, (void **)&d->cmap)) != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

void bitpar_free(BitparDevice *d) {
  void *ptrs[] = {d->U, d->S, d->LAST, d->INIT, d->lane_first, d->pid_of, d->cmap};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  *d = BitparDevice();
}

// pattern strings up to which the text-parallel form is the faster one (its cost grows with the
// strings, the tile form's floor is one pass with 64 strings per wave)
constexpr int BP_TP_MAX_STRINGS = 32;
constexpr int64_t BP_TP_SUB = 4096;

static bool bitpar_text_parallel(const BitparDevice &d) {
  if (d.knobs.bitpar_tp >= 0) return d.knobs.bitpar_tp != 0 && d.nlanes > 0;
  return d.nlanes > 0 && d.nlanes <= BP_TP_MAX_STRINGS;
}

ScanGeometry bitpar_geometry(const BitparDevice &d, int64_t begin, int64_t end) {
  ScanGeometry g;
  const int64_t range = std::max<int64_t>(end - begin, 1);
  if (bitpar_text_parallel(d)) {
    int64_t sub = BP_TP_SUB;
    while (sub > 256 && range / (64 * sub) < 1024) sub >>= 1;      // small ranges: still enough waves
    if (d.knobs.bitpar_seglen >= 16) sub = round_up((int64_t)d.knobs.bitpar_seglen, 16);   // test knob: force tiny runs
    g.seg_len = 64 * sub;
    const int64_t s_lo = begin / g.seg_len, s_hi = (end - 1) / g.seg_len;
    g.nseg = end > begin ? (int)(s_hi - s_lo + 1) : 0;
    g.threads = 256;
    g.blocks = (int)(((int64_t)g.nseg * d.nlanes + 3) / 4);
    return g;
  }
  // enough waves to fill 256 CUs several times over, but segments long enough that the
  // 256-byte halo each one re-reads stays small
  const int64_t target_waves = 32768;
  int64_t seg = round_up(std::max<int64_t>(range * std::max(d.ntiles, 1) / target_waves, 1), BP_BLOCK);
  seg = std::min<int64_t>(std::max<int64_t>(seg, 8192), 1 << 22);
  if (d.knobs.bitpar_seglen >= BP_BLOCK) seg = round_up((int64_t)d.knobs.bitpar_seglen, BP_BLOCK);   // test knob: force tiny segments
  g.seg_len = seg;
  const int64_t s_lo = begin / seg, s_hi = (end - 1) / seg;
  g.nseg = end > begin ? (int)(s_hi - s_lo + 1) : 0;
  g.threads = 256;
  g.blocks = (int)(((int64_t)g.nseg * d.ntiles + 3) / 4);
  return g;
}

hipError_t bitpar_launch(const BitparDevice &d, const uint8_t *d_text, int64_t n, int64_t begin, int64_t end,
                         pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, hipStream_t st,
                         ScanGeometry *geo_out) {
  if (end > n) end = n;
  ScanGeometry g = bitpar_geometry(d, begin, end);
  if (geo_out) *geo_out = g;
  if (g.blocks <= 0 || d.ntiles == 0) return hipSuccess;
  BitparArgs a;
  a.text = d_text; a.n = n; a.begin = begin; a.end = end;
  a.seg_len = g.seg_len; a.seg0 = begin / g.seg_len; a.nseg = g.nseg; a.ntiles = d.ntiles;
  a.halo = (int)round_up(d.maxlen + d.k, BP_BLOCK);
  a.U = d.U; a.S = d.S; a.LAST = d.LAST; a.INIT = d.INIT; a.lane_first = d.lane_first; a.pid_of = d.pid_of;
  a.cmap = d.cmap; a.out = d_out; a.counter = d_counter; a.cap = cap;
  const int key = d.k * 2 + (d.indels ? 1 : 0);
  a.nstr = d.nlanes; a.sub_len = g.seg_len / 64;
  if (bitpar_text_parallel(d)) {
    a.halo = (int)round_up(d.maxlen + d.k, 16);
    switch (key) {
      case 0: case 1: return launch_tp<0, false>(a, g.blocks, st);
      case 2: return launch_tp<1, false>(a, g.blocks, st);
      case 3: return launch_tp<1, true>(a, g.blocks, st);
      case 4: return launch_tp<2, false>(a, g.blocks, st);
      case 5: return launch_tp<2, true>(a, g.blocks, st);
      case 6: return launch_tp<3, false>(a, g.blocks, st);
      case 7: return launch_tp<3, true>(a, g.blocks, st);
    }
    return hipErrorInvalidValue;
  }
  switch (key) {
    case 0: case 1: return launch_t<0, false>(a, g.blocks, st);
    case 2: return launch_t<1, false>(a, g.blocks, st);
    case 3: return launch_t<1, true>(a, g.blocks, st);
    case 4: return launch_t<2, false>(a, g.blocks, st);
    case 5: return launch_t<2, true>(a, g.blocks, st);
    case 6: return launch_t<3, false>(a, g.blocks, st);
    case 7: return launch_t<3, true>(a, g.blocks, st);
  }
  return hipErrorInvalidValue;
}

// ---- verify-stage text windows ---------------------------------------------------------------
namespace {
__global__ void pm_gather_windows(const uint8_t *text, int64_t n, const int64_t *starts, const int32_t *lens,
                                  const int64_t *offs, int count, uint8_t *out) {
  const int wv = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (wv >= count) return;
  const int64_t s = starts[wv], o = offs[wv];
  const int len = lens[wv];
  for (int i = lane; i < len; i += 64) {
    const int64_t p = s + i;
    out[o + i] = (p >= 0 && p < n) ? text[p] : 0;
  }
}
}  // namespace

hipError_t gather_windows(const uint8_t *d_text, int64_t n, const int64_t *d_starts, const int32_t *d_lens,
                          const int64_t *d_offsets, int count, uint8_t *d_out, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  const int blocks = (count + 3) / 4;
  hipLaunchKernelGGL(pm_gather_windows, dim3(blocks), dim3(256), 0, st, d_text, n, d_starts, d_lens, d_offsets, count, d_out);
  return hipGetLastError();
}

}  // namespace pm

// pm_internal.h -- shared declarations of the product path (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pm_gpu.h"
#include "pm_align.h"

namespace pm {

struct Pattern {
  std::string s;
  uint64_t    id;
  int         esb, eeb;
};

// The CharacterProducer facts an engine needs (reference char_io.h:18-71): ch(), nch(), size().
struct Alphabet {
  int     size = 256;
  uint8_t ch[256];
  int     nch[256];
  bool    present[256];     // false: the code is known not to occur in the stream (only narrowed for raw streams with -w/-W)
  void set_raw();
  void set_table(const uint8_t *table, int len);
};

// Test and measurement knobs (PM_SEED_CHUNK, PM_SEED_GROUP, PM_SEED_DEBUG, PM_SEED_TILE, PM_PAIR, PM_HALF_SCAN,
// PM_EDIT_SCAN, PM_EDIT_TABLE_LOG, PM_BITPAR_TP, PM_BITPAR_SEGLEN, PM_DEBUG).  The environment is read ONCE, by
// pm_create (pm_api.cpp read_knobs), into the handle: nothing on the init or launch path calls getenv, and a
// handle's behaviour does not change when its caller's environment does.  Every field's 0 / -1 / false = unset.
struct Knobs {
  long long seed_chunk = 0;          // positions per workgroup of the seed-family scan kernels
  int seed_group = 0;                // chunks per run of one combo
  int seed_debug = 0;                // stage switches (measurement builds of the scan kernels)
  long seed_tile = 0;                // keys per pattern tile
  int pair = -1;                     // 0: keep -K 1 / -K 2 off the pair plan
  int pair_row = 0;                  // PM_PAIR_ROW: slots per row of the pair plan's slot table (measurement)
  bool half_bloom = false;           // exact_halves -k on the round-1 form (PM_HALF_SCAN=bloom)
  bool edit_bloom = false;           // edits: first stage = the round-1 pm_seed_scan instance (PM_EDIT_SCAN=bloom)
  bool edit_hash = false;            // edits: first stage = round 2's pm_edit_scan (PM_EDIT_SCAN=hash) where the pair geometry would run (-k 2, one tile)
  int edit_table_log = 0;            // edits: log2 of the key map's bits
  int bitpar_tp = -1;                // force the text-parallel (1) / tile (0) form of the bit-parallel kernel
  long long bitpar_seglen = 0;
  long long dense_bound = 0;         // PM_DENSE_BOUND: records per list beyond which pm_scan cuts a range in two (0: 2^29)
  bool debug = false;                // PM_DEBUG: stage timings on stderr
};

// ---- bit-parallel family (pm_bitpar.hip) -----------------------------------------------------
constexpr int BP_WPL = 8;          // 32-bit words per lane: a lane holds a 256-bit pattern string
constexpr int BP_NC = 6;           // distinct stream codes the patterns may accept (A,C,G,T,N + one)
constexpr int BP_BLOCK = 256;      // text bytes per block step (64 lanes x 4 bytes)

struct BitparTables {              // host-built, then uploaded
  int ntiles = 0;                  // 64 lanes per tile
  int nlanes = 0;                  // lanes (256-bit pattern strings) in use, packed from lane 0 of tile 0
  int k = 0;
  int maxlen = 0;
  int nclasses = 0;                // pattern codes in use (<= BP_NC)
  uint8_t cmap[256];               // text code -> class: 0..BP_NC-1, BP_NC = other, BP_NC+1 = EOS
  std::vector<uint32_t> U;         // [tile][cls][w][lane]
  std::vector<uint32_t> S;         // [tile][w][lane]   first-char bits
  std::vector<uint32_t> LAST;      // [tile][w][lane]   last-char bits
  std::vector<uint32_t> INIT;      // [tile][l-1][w][lane]  row l start state (l = 1..k)
  std::vector<uint32_t> lane_first;// [tile*64+lane] -> first index into pid_of for that lane
  std::vector<uint32_t> pid_of;    // pattern id per packed pattern, in (tile,lane,bit) order
};

struct BitparDevice {
  uint32_t *U = nullptr, *S = nullptr, *LAST = nullptr, *INIT = nullptr, *lane_first = nullptr, *pid_of = nullptr;
  uint8_t  *cmap = nullptr;
  int ntiles = 0, nlanes = 0, k = 0, maxlen = 0;
  bool indels = false;
  Knobs knobs;                     // set by the caller after bitpar_upload
};

// Build the packed tables.  Returns "" or an error message (PM_E_UNSUPPORTED).
std::string bitpar_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids,
                         const Alphabet &alpha, int k, int eos_code, BitparTables *out,
                         bool wildcards = false, bool text_n = false);
hipError_t bitpar_upload(const BitparTables &t, bool indels, BitparDevice *d, hipStream_t st);
void bitpar_free(BitparDevice *d);

struct ScanGeometry { int64_t seg_len; int nseg; int blocks; int threads; };
ScanGeometry bitpar_geometry(const BitparDevice &d, int64_t begin, int64_t end);

// Enqueue the scan of stream range (begin,end] (hit end positions) on `st`.
// out/counter are device pointers; counter must be zeroed by the caller (async memset).
hipError_t bitpar_launch(const BitparDevice &d, const uint8_t *d_text, int64_t n, int64_t begin, int64_t end,
                         pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, hipStream_t st,
                         ScanGeometry *geo_out);
const char *bitpar_kernel_name(int k, bool indels);

// which byte values occur in the stream (pm_util.hip); used to bound the character classes of -w/-W on raw streams
hipError_t stream_presence(const uint8_t *d_text, int64_t n, bool present[256], hipStream_t st);

// ---- window gather (verify stage text access when the stream lives only in HBM) -------------
hipError_t gather_windows(const uint8_t *d_text, int64_t n, const int64_t *d_starts, const int32_t *d_lens,
                          const int64_t *d_offsets, int count, uint8_t *d_out, hipStream_t st);


// ---- device clustering for -K filter_bitvec (pm_cluster.hip) ----------------------------------
// Position sharding: the records cover ends in (guard_lo, guard_hi]; this shard
// reports the clusters whose hit ends in (own_lo, own_hi].  guard_lo <= 0 / guard_hi = INT64_MAX
// mean the true start / end of the stream.
struct OwnedRange { int64_t own_lo, own_hi, guard_lo, guard_hi; int on; };

size_t cluster_temp_bytes(size_t n);
// invalid_level: records with this level chain like any other but are never a chain's hit (-1: none)
hipError_t cluster_device(const pm_hit *d_in, size_t n1, const pm_hit *d_in2, size_t n2, int k, int64_t scanned_to, bool last, int invalid_level,
                          const uint8_t *d_pat_len, const uint32_t *d_pat_id, const OwnedRange &own,
                          uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                          pm_hit *d_out, pm_hit *d_left, unsigned long long *d_counts, hipStream_t st);


// final hits in (end, pid, k) order on the device (pm_cluster.hip): see sort_final_device
hipError_t sort_final_device(const pm_hit *d_in, const unsigned long long *d_count, size_t n_upper, const uint32_t *d_pat_id, uint32_t npat,
                             int idxbits, int keybits, uint64_t *d_keys, uint64_t *d_keys_alt, pm_hit *d_out, void *d_temp, size_t temp_bytes, hipStream_t st);

// records of d_in that end in the owned range, compacted into d_out (pass-through engines, sharded scans)
hipError_t owned_filter_device(const pm_hit *d_in, size_t n, const OwnedRange &own, pm_hit *d_out, unsigned long long *d_count, hipStream_t st);

constexpr uint32_t PM_SEED_HOLE = 0xffffffffu;   // pid of an unused slot in the half-seed record buffer
constexpr int SEED_OUT_BLOCK = 64;               // slots a wave reserves per atomic (exact_halves -k seeds)

// duplicate and hole removal for the edit-distance seed plan (pm_cluster.hip); d_out may alias d_in
hipError_t dedup_device(const pm_hit *d_in, size_t n, uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                        pm_hit *d_out, unsigned long long *d_count, hipStream_t st);

hipError_t cluster_dp_device(const pm_hit *d_in, size_t n1, const pm_hit *d_in2, size_t n2, int k, bool indels, int64_t scanned_to, bool last,
                             const uint8_t *d_text, int64_t ntext, int eos_code,
                             const uint8_t *d_pat_codes, const uint8_t *d_pat_len, const int32_t *d_esb, const int32_t *d_eeb,
                             const uint32_t *d_pat_id, const OwnedRange &own, uint64_t *d_keys, uint64_t *d_keys_alt, void *d_temp, size_t temp_bytes,
                             pm_hit *d_out, pm_hit *d_left, unsigned long long *d_counts, hipStream_t st);

// exact_halves' sequential per-pattern rule on the device (pm_cluster.hip)
size_t halves_temp_bytes(size_t n);
hipError_t halves_rule_device(const pm_hit *d_in, size_t n, bool flags, int slack, const uint8_t *d_pat_len, const uint32_t *d_pat_id,
                              uint64_t *d_keys, uint64_t *d_keys_alt, uint32_t *d_vals, uint32_t *d_vals_alt, void *d_temp, size_t temp_bytes,
                              pm_hit *d_out, unsigned long long *d_counts, hipStream_t st);

// ---- seed extension DP on the GPU (pm_extend.hip) ---------------------------------------------
hipError_t extend_seeds(const uint8_t *d_text, int64_t n, const pm_hit *d_seeds, size_t nseeds,
                        const uint8_t *d_half_codes, const uint8_t *d_half_len, const int32_t *d_esb, const int32_t *d_eeb,
                        int k, int eos_code, pm_hit *d_out, unsigned long long *d_counter, size_t cap, hipStream_t st);

}  // namespace pm

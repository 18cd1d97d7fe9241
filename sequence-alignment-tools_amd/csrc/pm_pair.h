// pm_pair.h -- "pair plan" scan kernel (pm_pair.hip): host tables and launch interface.
//
// Substitution-only candidates (-K 1, -K 2) of 20..32 character A,C,G,T patterns.  The last 20
// bases of a pattern are cut into four fields of five bases (10 bits of the 2-bit packed window);
// <= 2 substitutions leave two fields untouched (<= 1: the first two or the last two), so every
// candidate agrees with its pattern on one of the C(4,2) = 6 field pairs (2 for k = 1).  A pair of
// fields is a 20-bit key: 2^20 keys are exactly the bits of a 128 KiB LDS bitmap, so the first
// stage is an exact membership test (one v_alignbit, one ds_read_b32, one shift) instead of a
// hashed Bloom filter, and it runs over 6 (2) combos instead of the 10 (4) of the byte-piece plan.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "pm_internal.h"

namespace pm {

constexpr int PAIR_THREADS = 1024;                  // 16 waves, one workgroup per CU (LDS bound)
constexpr int PAIR_WAVES = PAIR_THREADS / 64;
constexpr int PAIR_MAX_COMBOS = 6;
constexpr int PAIR_BITMAP_WORDS = 32768;            // 2^20 key bits: key bits 0..14 = row (dword), 15..19 = bit
constexpr int PAIR_QUEUE = 128;                     // per wave: suspect records (16 bytes) waiting to leave in a batch
constexpr int PAIR_LDS_BYTES = PAIR_BITMAP_WORDS * 4 + PAIR_WAVES * PAIR_QUEUE * 16;   // key bitmap + the waves' suspect queues
static_assert(PAIR_LDS_BYTES <= 163840, "160 KiB of LDS per workgroup");

struct PairTables {                                 // one pattern tile
  int k = 0, maxlen = 0, ncombos = 0, eos_code = -1, stride = 0;
  bool ascii = false;
  int fa[PAIR_MAX_COMBOS] = {}, fb[PAIR_MAX_COMBOS] = {};   // key fields of every combo (a < b), the other two are (c < d)
  std::vector<uint32_t> image;                      // [combo][PAIR_BITMAP_WORDS]
  std::vector<uint64_t> slots;                      // [combo][PAIR_BITMAP_WORDS * stride]: slot (row, rank of the key inside its row), see pm_pair.hip slot_pack
  std::vector<uint32_t> row_base;                   // [combo][PAIR_BITMAP_WORDS + 1]: keys that occur in front of every row (rank of a key = row_base[row] + rank in row)
  std::vector<uint32_t> first_pat;                  // [combo][distinct keys + 1]: by rank, first index into order[] of every key
  std::vector<uint32_t> order;                      // [combo][np]: pattern indices sorted by key
  std::vector<uint32_t> olist;                      // [combo][np]: the other 20 window bits of order[]'s patterns (pm_pair_verify walks a key's run here)
  size_t first_off[PAIR_MAX_COMBOS] = {};           // per combo, in elements of first_pat
  std::vector<uint64_t> pat40;                      // last 20 bases, 2 bits each
  std::vector<uint8_t> pat_len;
  std::vector<uint32_t> pat_id;
  std::vector<uint8_t> pat_codes;                   // 32 stream codes per pattern
  std::vector<uint32_t> pat_zone;                   // bit i: pattern character i lies in an exact zone
};

struct PairDevice {
  int k = 0, maxlen = 0, ncombos = 0, eos_code = -1, stride = 0;
  bool ascii = false;
  int fa[PAIR_MAX_COMBOS] = {}, fb[PAIR_MAX_COMBOS] = {};
  size_t first_off[PAIR_MAX_COMBOS] = {};
  size_t np = 0;
  uint64_t *slots = nullptr;
  uint32_t *image = nullptr, *row_base = nullptr, *first_pat = nullptr, *order = nullptr, *olist = nullptr, *pat_id = nullptr;
  uint64_t *pat40 = nullptr;
  uint8_t *pat_len = nullptr, *pat_codes = nullptr;
  uint32_t *pat_zone = nullptr;
  int viol_level = 0;                                 // see PairArgs::viol_level (set by the caller after pair_upload)
  Knobs knobs;                                        // likewise
};

// "" or why the plan does not take this pattern set
// stride_knob: slots per row of the slot table (0 = default)
std::string pair_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids, const Alphabet &alpha, int k,
                       int eos_code, PairTables *out, int stride_knob = 0, int slot_patterns = 3);
hipError_t pair_upload(const PairTables &t, PairDevice *d, hipStream_t st);
void pair_free(PairDevice *d);
ScanGeometry pair_geometry(const PairDevice &d, int64_t begin, int64_t end);
// d_susp: susp_cap 16-byte suspect records between the scan kernel and pm_pair_verify; *d_susp_count is
// zeroed by the caller before the launch and holds the number of suspects afterwards (> susp_cap: the
// buffer was too small and the candidate records are incomplete)
constexpr size_t PAIR_SUSPECT_BYTES = 16;
hipError_t pair_launch(const PairDevice &d, const uint8_t *d_text, const uint32_t *d_packed, int64_t n, int64_t begin, int64_t end,
                       pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, void *d_susp, unsigned long long *d_susp_count, uint64_t susp_cap,
                       hipStream_t st, ScanGeometry *geo_out, unsigned long long *d_stats = nullptr, int floor_mode = 0,
                       uint64_t *seed_out = nullptr, unsigned long long *seed_count = nullptr, uint64_t seed_cap = 0);
// floor_mode 3 = the edit-distance plan on the pair geometry (-k 2): pm_pair_edit_scan<3> (14 tests per window, edit_cost) +
// pm_pair_edit_resolve, seed records (pattern index << 40 | position) into seed_out for pm_edits_verify; tables built
// with slot_patterns = 2.  floor_mode 1, 2: measurement kernels (pm_measure_pair_edit_floor).

}  // namespace pm

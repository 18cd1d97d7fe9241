// pm_seed.h -- seed-filter kernel family (pm_seed.hip): host tables and launch interface.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "pm_internal.h"

namespace pm {

constexpr int SEED_THREADS = 1024;                // 16 waves per workgroup, one workgroup per CU (LDS bound)
constexpr int SEED_QCAP = 128;                    // first survivor queue (Bloom survivors), entries per wave
constexpr int SEED_Q2CAP = 112;                   // second queue (second-level bitmap survivors)
constexpr int SEED_BLOOM_WORDS = 32768;           // 128 KiB blocked Bloom filter per combo
constexpr int SEED_BLOOM_STRIDE = SEED_BLOOM_WORDS + 4;   // + the bytes a 32-bit block that starts at the last byte reaches (16-byte granule)
constexpr int SEED_MAX_COMBOS = 16;
constexpr int SEED_LDS_BYTES = SEED_BLOOM_STRIDE * 4 + (SEED_THREADS / 64) * (SEED_QCAP + SEED_Q2CAP) * 8;   // filter + wave queues

struct SeedTables {
  int k = 0, Lw = 0, pb = 0, r = 0, maxlen = 0, mode = 0;
  bool ascii = false;
  std::vector<std::array<int, 4>> combos;
  std::vector<uint32_t> perm_sel;         // piece indices of every combo
  size_t nslots = 0;
  int idx_bits = 0, bucket_shift = 0;
  std::vector<uint32_t> bloom;                    // [combo][SEED_BLOOM_STRIDE]
  std::vector<uint32_t> slots;                    // [combo][nbuckets][8]
  std::vector<uint32_t> bitmap2;                  // [combo][2^(lb2-5)]
  int lb2 = 0;
  struct P40 { uint32_t lo, hi; };
  std::vector<P40> pat40;
  std::vector<uint8_t> pat_len;
  std::vector<uint32_t> pat_id;
  std::vector<uint8_t> pat_codes;                 // 32 bytes per pattern
  bool halves = false; int hk = 0;                // exact_halves -k mode: partner half per pattern
  int hfast = 0;                                  // see SeedArgs::hfast
  bool exact_filter = false;                      // see SeedArgs::exact_filter
  // exact_halves -k, ranked form (pm_half_scan): key bitmap + rank directory, dense slots by rank, halves by key
  std::vector<uint32_t> hr_image, hr_first, hr_order;
  std::vector<uint64_t> hr_slots, hr_more;
  int edits = 0;                                  // > 0: edit-distance seed plan for this k (records in pat_codes)
  std::vector<uint8_t> etable;                    // edits: per combo, key-hash bit map of pm_edit_scan (2^etable_log bits)
  int etable_log = 0;
  std::vector<uint32_t> eidx;                     // edits: pattern index of every bucket slot (the slots hold pattern pieces + fingerprint)
  int eos_code = -1;
  std::vector<uint32_t> part32;
  std::vector<uint8_t> part_len, part_side;
  uint8_t cmap[256];
};

struct SeedDevice {
  uint32_t *bloom = nullptr, *slots = nullptr, *pat_id = nullptr, *bitmap2 = nullptr;
  int lb2 = 0;
  void *pat40 = nullptr, *d_args = nullptr;
  uint8_t *pat_len = nullptr, *pat_codes = nullptr, *cmap = nullptr, *part_len = nullptr, *part_side = nullptr, *etable = nullptr;
  uint32_t *eidx = nullptr;
  uint32_t *hr_image = nullptr, *hr_first = nullptr, *hr_order = nullptr;   // exact_halves -k: pm_half_scan / pm_half_verify tables
  uint64_t *hr_slots = nullptr, *hr_more = nullptr;
  bool half_ranked = false;
  int etable_log = 0;
  bool edit_tabulated = false;                    // edits: the first stage is pm_edit_scan (PM_EDIT_SCAN=bloom selects the older pm_seed_scan instance)
  uint32_t *part32 = nullptr;
  bool halves = false; int hk = 0, hfast = 0, eos_code = -1, edits = 0;
  bool exact_filter = false;
  uint32_t emask_a[SEED_MAX_COMBOS] = {}, emask_b[SEED_MAX_COMBOS] = {}, evar[SEED_MAX_COMBOS] = {};
  uint32_t mask_lo[SEED_MAX_COMBOS] = {}, mask_hi[SEED_MAX_COMBOS] = {}, perm_sel[SEED_MAX_COMBOS] = {};
  int mode = 0;
  int k = 0, Lw = 0, pb = 0, r = 0, ncombos = 0, maxlen = 0;
  bool ascii = false;
  size_t nslots = 0;
  int idx_bits = 0, bucket_shift = 0;
  Knobs knobs;                                    // set by the caller after seed_upload
};

std::string seed_build(const std::vector<Pattern> &pats, const std::vector<uint32_t> &ids,
                       const Alphabet &alpha, int k, int eos_code, SeedTables *out, int force_lmin = 0,
                       const std::vector<std::string> *partners = nullptr, const std::vector<uint8_t> *sides = nullptr,
                       int halves_k = 0, bool edits = false, const Knobs &knobs = Knobs());
hipError_t seed_upload(const SeedTables &t, SeedDevice *d, hipStream_t st);
void seed_free(SeedDevice *d);
ScanGeometry seed_geometry(const SeedDevice &d, int64_t begin, int64_t end);
// edit-distance plan (and the ranked exact_halves -k plan): where the scan kernel puts its records (the verify kernel reads them)
struct EditStage {
  uint64_t *d_seeds = nullptr;                    // 8-byte records: pattern index << 40 | position
  unsigned long long *d_seed_count = nullptr;     // zeroed by the caller before every launch
  uint64_t seed_cap = 0;
  int tile = 0;
  bool skip_scan = false;                         // the seed records are there already (pattern index << 40 | position: pm_pair.hip's edit plan): run the verify kernel only
  // exact_bases -k (see pm_bases_verify): per pattern of the whole list (seed-plan pattern ids are 1-based indices into it)
  bool bases = false;
  const uint8_t *b_codes = nullptr, *b_len = nullptr;             // 32 stream codes per pattern, length
  const int32_t *b_esb = nullptr, *b_eeb = nullptr;
  int64_t own_lo = 0, own_hi = 0;                                   // seed records with own_lo < end <= own_hi are reported
};
// 2-bit form of the stream for the scan kernels' first stage: d_packed holds (n + 15) / 16 dwords
hipError_t pack_stream(const uint8_t *d_text, int64_t n, bool ascii, uint32_t *d_packed, int64_t npacked, hipStream_t st);
hipError_t seed_launch(const SeedDevice &d, const uint8_t *d_text, const uint32_t *d_packed, int64_t n, int64_t begin, int64_t end,
                       pm_hit *d_out, unsigned long long *d_counter, uint64_t cap, hipStream_t st,
                       ScanGeometry *geo_out, const EditStage *es = nullptr);

}  // namespace pm

// pm_api.cpp -- the C ABI of include/pm_gpu.h: handle life cycle, engine selection, the device
// scan stage and the host verify/cluster stage.
//
// Split of work (DESIGN.md "stages"):
//   device  : scan kernels (pm_bitpar.hip / pm_seed.hip) turn the stream into sparse candidate
//             records -- a pure function of (stream range, patterns), so it shards by position;
//   host    : the order-dependent, sparse part of the reference wrappers -- filter_bitvec's
//             clustering + one verify per cluster (filter_bitvec.cc:88-177), exact_halves'
//             seed extension + per-pattern dedup (exact_halves.cc:140-190), exact_bases
//             (exact_bases.cc:92-121) -- in stream order over the gathered records.
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <chrono>
#include <thread>
#include <unordered_map>

#include "pm_internal.h"
#include "pm_iupac.h"
#include "pm_seed.h"
#include "pm_pair.h"

using namespace pm;

namespace pm {
void Alphabet::set_raw() {
  size = 256;
  for (int i = 0; i < 256; ++i) { ch[i] = (uint8_t)i; nch[i] = i; present[i] = true; }
}
void Alphabet::set_table(const uint8_t *table, int len) {      // char_io.t:222-236
  size = len;
  for (int i = 0; i < 256; ++i) { ch[i] = 0; nch[i] = -1; present[i] = true; }
  for (int i = 0; i < len; ++i) { ch[i] = table[i]; nch[table[i]] = i; }
}
}  // namespace pm

struct pm_handle {
  pm_config cfg{};
  Knobs knobs;                        // the environment's test / measurement knobs as pm_create found them
  std::vector<Pattern> pats;
  std::unordered_map<uint32_t, uint32_t> id2idx;   // caller's pattern id -> index into pats
  Alphabet alpha;
  int eos_code = -1;
  int sem = 0, kern = 0;
  bool inited = false;
  bool host_only = false;             // pm_init_host: host stage only, no stream on the device

  const uint8_t *h_text = nullptr;
  const uint8_t *d_text = nullptr;
  bool own_d_text = false;
  int64_t n = 0;
  hipStream_t stream = nullptr;

  // device stage
  std::vector<Pattern> inner;
  std::vector<uint32_t> inner_ids;
  BitparDevice bp;
  SeedDevice sd;                      // first pattern tile (plan parameters are read from here)
  std::vector<SeedDevice> sd_more;    // further tiles when the pattern set is too large for one LDS filter
  std::vector<PairDevice> pair;       // -K 1 / -K 2 on 20..32 character patterns: the pair plan's tiles (pm_pair.hip) instead of sd
  PairDevice epair;                   // -k 2 (filter_bitvec / shift_and_inexact): the pair geometry as the edit plan's first stage (one pattern tile)
  bool epair_on = false;
  bool seed_flags = false;            // exact_halves on whole-pattern Hamming candidates (aux flags)
  bool bases_flags = false;           // exact_bases -K on whole-pattern Hamming candidates with clean exact zones (pair plan): records are final
  bool zoned = false;                 // some pattern has exact-base constraints
  bool nn_quirk = false;              // -w without -W on a stream that holds N, and some pattern has the letter N (pattern_n_quirk)
  bool wild_seed = false;             // -w/-W on the seed family: primers with <= 2 ambiguity letters expanded into their concrete variants
  bool bases_edits = false;           // exact_bases -k on the seed family: the k-error automaton's candidates -> block seeds (pm_bases_seeds)
  int64_t own_begin = 0, own_end = 0; //   the range the caller asked for (the candidates are scanned a little wider)
  bool halves_dev = false;            // exact_halves -k: half seeds extended by pm_seed_extend on the GPU
  bool half_ranked_any = false;       // ... and some pattern tile runs on pm_half_scan (needs the seed record buffer)
  bool edits_dev = false;             // filter_bitvec / shift_and_inexact -k on the seed kernels: records deduplicated after the scan
  unsigned long long *d_seed_count = nullptr;   // edits: [0] unused, [1+t] seed records of tile t
  uint64_t *d_seeds = nullptr;                  // edits: 8-byte seed records between the scan and the verify kernel
  size_t seed_cap = 0;
  void *d_susp = nullptr;                       // pair plan: 16-byte suspect records between its scan and verify kernels
  size_t susp_cap = 0;
  unsigned long long *h_seed_count = nullptr;   // pinned, 1 + 256 entries: a copy into pageable memory would make the "async" scan call wait for the kernels
  std::vector<pm_hit> start_cache;    // edits: candidates that end in the first Lw+2k+2 characters (whole-prefix scans only)
  bool start_cached = false;
  std::vector<pm_hit> end_cache;      // edits: candidates that end in the last four characters (scans that reach the end of the stream)
  bool end_cached = false;
  std::vector<pm_hit> overhang_cache; // exact_halves / exact_bases -K: hits that hang over the end of the stream
  bool overhang_cached = false;
  std::vector<pm_hit> edge_cache;     // exact_bases -k: block occurrences in the first and last 56 characters
  bool edge_cached = false;
  std::vector<pm_hit> head_cache;     // -K on the automaton's semantics: records of the stream start (stream_start_candidates)
  bool head_cached = false;
  uint8_t *d_dp_codes = nullptr;      // device DP (pm_cluster_dp): 32 stream codes per pattern, exact zones
  int32_t *d_dp_esb = nullptr, *d_dp_eeb = nullptr;
  pm_hit *d_ext = nullptr;            // its output (swapped with d_cands after every scan)
  uint8_t *d_half_codes = nullptr, *d_half_len = nullptr;
  int32_t *d_hesb = nullptr, *d_heeb = nullptr;
  int scan_k = 0, seed_k = 0;
  int64_t scan_end = 0;               // end of the range of the scan in flight / last scan
  bool scan_indels = false;
  pm_hit *d_cands = nullptr;
  unsigned long long *d_counter = nullptr;
  unsigned long long *h_counter = nullptr;     // pinned
  size_t cap = 0;
  size_t last_count = 0;
  size_t overflow_need = 0;           // record count a PM_E_OVERFLOW of the host-added records asks for
  int64_t scan_begin = 0;
  bool scan_pending = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  int last_launches = 0;
  unsigned long long internal_rescans = 0;   // scans repeated inside pm_scan_wait since pm_init (an internal buffer was too small)
  // pm_scan on hit-dense text: a range whose record lists would outgrow dense_bound records is scanned in pieces (scan_range)
  // instead of growing the lists; bound_on is set while pm_scan drives the scan (direct pm_scan_candidates calls keep growing)
  bool bound_on = false, too_dense = false, dense_mode = false;
  int64_t piece_len = 0;                     // pm_scan scans in pieces of at most this many positions (0: whole ranges)
  unsigned long long last_peak = 0;          // longest record list of the last scan
  unsigned long long range_splits = 0;       // times pm_scan halved its piece length since pm_init
  ScanGeometry geo{};

  // host stage state
  std::vector<pm_hit> carry;          // filter_bitvec: candidates whose cluster is not complete yet
  std::vector<int64_t> lasthit;       // exact_halves: last kept end per pattern (exact_halves.cc:163)
  // pm_scan: the final hits of the ranges scanned so far, in (end, pid, k) order, in pinned host memory; handed out
  // by pm_scan (copied) or pm_scan_view (as a span) from land_pos on
  pm_hit *land = nullptr;
  size_t land_cap = 0, land_n = 0, land_pos = 0;
  hipStream_t copy_stream = nullptr;  // copies out of HBM that run beside the next range's scan (non-blocking stream)
  hipEvent_t ev_fin = nullptr;        // "the finalize stage of this range is done" on the handle's stream
  pm_hit *d_fsorted = nullptr;        // final hits after the device sort
  size_t fsorted_cap = 0;
  bool sort_dev = false;              // (end, pid, k) order by one keys-only radix sort on the device (device_sort_plan)
  int sort_idxbits = 0, sort_keybits = 0;
  bool spec = false;                  // pm_scan has the scan of the range it expects next in flight: (spec_b, spec_e]
  int64_t spec_b = 0, spec_e = 0;
  int64_t next_begin = 0;
  AlignScratch scratch;
  // window staging (verify stage)
  std::vector<uint8_t> winbuf;
  int64_t *d_wstart = nullptr; int32_t *d_wlen = nullptr; int64_t *d_woff = nullptr; uint8_t *d_wout = nullptr;
  size_t d_wcap = 0, d_woutcap = 0;

  // device finalize workspace (pm_finalize_device)
  uint64_t *d_keys = nullptr, *d_keys_alt = nullptr;
  void *d_ctemp = nullptr;
  size_t ckeys_cap = 0, ctemp_bytes = 0;
  pm_hit *d_fout = nullptr, *d_fleft = nullptr;
  const pm_hit *d_final = nullptr;    // final hits left in HBM by a finalize call with out == NULL
  size_t n_final = 0;
  float pack_ms = 0.f;                // duration of the last 2-bit re-encoding of the stream (ensure_packed)
  unsigned long long *d_fcounts = nullptr, *h_fcounts = nullptr;
  uint8_t *d_fpat_len = nullptr;
  uint32_t *d_fpat_id = nullptr;
  // exact_halves rule on the device: payload arrays and the pair sort's workspace
  uint32_t *d_packed = nullptr;       // the stream at 2 bits per base (seed family's first stage), rebuilt by every pm_init
  size_t packed_cap = 0;
  pm_hit *d_carry = nullptr;          // carried candidates of an earlier range, uploaded for the device clustering
  size_t d_carry_cap = 0;
  uint32_t *d_vals = nullptr, *d_vals_alt = nullptr;
  void *d_htemp = nullptr;
  size_t vals_cap = 0, htemp_bytes = 0;
  std::vector<uint8_t> in_rest;       // per inner pattern: 1 = scanned by the bit-parallel residue engine beside the seed family
  size_t nrest = 0;
  bool halves_fresh = true;           // no host-side exact_halves state (lasthit, carried seeds) since init / pm_reset

  std::string err;
};

// per-tile record counts between a scan's first-stage and verify kernels: [0] unused, [1 + t] tile t, then 8 measurement counters
constexpr size_t SEEDCOUNT_WORDS = 1 + 256 + 8;

static thread_local std::string g_create_error;

static int fail(pm_handle *h, int code, const std::string &msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
static int hipfail(pm_handle *h, hipError_t e, const char *what) {
  return fail(h, PM_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hipfail((h), e_, #expr); } while (0)

// ---- pick_pattern_index, automatic branch (reference select.cc:31-141, NOPRIMEGEN, -x 0) -----
extern "C" int pm_pick_semantics(int32_t alphabet_size, int32_t acgt_normalized, int32_t k, int32_t wildcards,
                                 int32_t npat, const int32_t *patlen, const int32_t *esb, const int32_t *eeb) {
  long min_exact = INT_MAX, at_least_half = 0, excess = 0, min_inexact = INT_MAX, min_len = INT_MAX;
  for (int i = 0; i < npat; ++i) {
    const int a = esb ? esb[i] : 0, b = eeb ? eeb[i] : 0;
    const int c = a >= b ? a : b;                                  // the larger exact block decides
    min_exact = std::min<long>(min_exact, c);
    excess += c - patlen[i] / 2;
    at_least_half += (c - patlen[i] / 2) >= 0;
    min_inexact = std::min<long>(min_inexact, patlen[i] - c);
    min_len = std::min<long>(min_len, patlen[i]);
  }
  min_inexact = std::min(min_inexact, min_len);
  if (k > 0 && k >= min_inexact) return PM_E_FATAL;                // select.cc:87-90
  int exact_engine;                                                // select.cc:101-115
  if (wildcards) exact_engine = 4;
  else if (alphabet_size < 255) exact_engine = acgt_normalized ? 2 : 3;
  else exact_engine = 3;
  if (k == 0) return exact_engine == 4 ? PM_SEM_SHIFT_AND : PM_SEM_KEYWORD_TREE;
  const bool long_enough = (min_len >= 12 && alphabet_size < 10) || (min_len >= 8 && alphabet_size >= 10);
  if (k == 1 && long_enough && (at_least_half <= 0 || excess <= 0)) return PM_SEM_EXACT_HALVES;   // :121-126
  if (min_exact >= 6) return PM_SEM_EXACT_BASES;                   // :131-133
  return PM_SEM_FILTER_BITVEC;                                     // :137-139
}

// The one place the library looks at its environment (see Knobs, pm_internal.h).
static void read_knobs(Knobs *k) {
  auto num = [](const char *name) -> long long { const char *v = getenv(name); return v && *v ? atoll(v) : 0; };
  auto is = [](const char *name, const char *val) { const char *v = getenv(name); return v && !strcmp(v, val); };
  *k = Knobs();
  k->seed_chunk = num("PM_SEED_CHUNK"); k->seed_group = (int)num("PM_SEED_GROUP"); k->seed_debug = (int)num("PM_SEED_DEBUG");
  k->seed_tile = (long)num("PM_SEED_TILE");
  if (const char *v = getenv("PM_PAIR")) k->pair = atoi(v);
  k->pair_row = (int)num("PM_PAIR_ROW");
  k->half_bloom = is("PM_HALF_SCAN", "bloom"); k->edit_bloom = is("PM_EDIT_SCAN", "bloom"); k->edit_hash = is("PM_EDIT_SCAN", "hash");
  k->edit_table_log = (int)num("PM_EDIT_TABLE_LOG");
  if (const char *v = getenv("PM_BITPAR_TP")) k->bitpar_tp = atoi(v) != 0;
  k->bitpar_seglen = num("PM_BITPAR_SEGLEN");
  k->dense_bound = num("PM_DENSE_BOUND");
  k->debug = getenv("PM_DEBUG") != nullptr;
}

extern "C" int pm_prepare_device(int device) {
  if (hipSetDevice(device) != hipSuccess) return PM_E_HIP;
  return hipFree(nullptr) == hipSuccess ? PM_OK : PM_E_HIP;         // (forces the runtime's lazy initialisation)
}

extern "C" int pm_create(const pm_config *cfg, pm_handle **out) {
  if (!cfg || !out) return fail(nullptr, PM_E_INVALID, "pm_create: null argument");
  if (cfg->abi_version != PM_ABI_VERSION) return fail(nullptr, PM_E_INVALID, "pm_create: ABI version mismatch");
  if (cfg->k < 0) return fail(nullptr, PM_E_INVALID, "pm_create: negative k");
  pm_handle *h = new (std::nothrow) pm_handle();
  if (!h) return fail(nullptr, PM_E_NOMEM, "out of memory");
  h->cfg = *cfg;
  read_knobs(&h->knobs);
  h->alpha.set_raw();
  *out = h;
  return PM_OK;
}

extern "C" int pm_add_pattern(pm_handle *h, const char *pat, size_t len, uint64_t id, int32_t esb, int32_t eeb) {
  if (!h || !pat) return PM_E_INVALID;
  if (h->inited) return fail(h, PM_E_INVALID, "pm_add_pattern after pm_init");
  if (id == 0) id = h->pats.size() + 1;                            // pattern_match.h:92-94
  if (id > 0xffffffffull) return fail(h, PM_E_UNSUPPORTED, "pattern ids above 2^32-1");
  h->id2idx[(uint32_t)id] = (uint32_t)h->pats.size();
  h->pats.push_back(Pattern{std::string(pat, len), id, esb, eeb});
  return PM_OK;
}

static void drain_spec(pm_handle *h);
static int ensure_landing(pm_handle *h, size_t need_more);
static void device_sort_plan(pm_handle *h);

static void free_device(pm_handle *h) {
  drain_spec(h);
  if (h->land) (void)hipHostFree(h->land);
  h->land = nullptr; h->land_cap = h->land_n = h->land_pos = 0;
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->ev_fin) (void)hipEventDestroy(h->ev_fin);
  h->copy_stream = nullptr; h->ev_fin = nullptr;
  if (h->d_fsorted) (void)hipFree(h->d_fsorted);
  h->d_fsorted = nullptr; h->fsorted_cap = 0;
  bitpar_free(&h->bp);
  seed_free(&h->sd);
  for (SeedDevice &d : h->sd_more) seed_free(&d);
  h->sd_more.clear();
  for (PairDevice &d : h->pair) pair_free(&d);
  h->pair.clear();
  pair_free(&h->epair); h->epair_on = false;
  if (h->d_cands) (void)hipFree(h->d_cands);
  { void *hx[] = {h->d_ext, h->d_half_codes, h->d_half_len, h->d_hesb, h->d_heeb, h->d_dp_codes, h->d_dp_esb, h->d_dp_eeb, h->d_seed_count, h->d_seeds, h->d_susp}; for (void *q : hx) if (q) (void)hipFree(q); }
  h->d_seed_count = nullptr; h->d_seeds = nullptr; h->d_susp = nullptr; h->susp_cap = 0;
  h->d_dp_codes = nullptr; h->d_dp_esb = h->d_dp_eeb = nullptr;
  h->d_ext = nullptr; h->d_half_codes = h->d_half_len = nullptr; h->d_hesb = h->d_heeb = nullptr;
  if (h->d_counter) (void)hipFree(h->d_counter);
  if (h->h_counter) (void)hipHostFree(h->h_counter);
  if (h->h_seed_count) (void)hipHostFree(h->h_seed_count);
  h->h_seed_count = nullptr;
  if (h->own_d_text && h->d_text) (void)hipFree((void *)h->d_text);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->d_wstart) (void)hipFree(h->d_wstart);
  if (h->d_wlen) (void)hipFree(h->d_wlen);
  if (h->d_woff) (void)hipFree(h->d_woff);
  if (h->d_wout) (void)hipFree(h->d_wout);
  void *fw[] = {h->d_keys, h->d_keys_alt, h->d_ctemp, h->d_fout, h->d_fleft, h->d_fcounts, h->d_fpat_len, h->d_fpat_id, h->d_vals, h->d_vals_alt, h->d_htemp, h->d_carry, h->d_packed};
  h->d_packed = nullptr; h->packed_cap = 0;
  h->d_vals = h->d_vals_alt = nullptr; h->d_htemp = nullptr; h->vals_cap = 0; h->htemp_bytes = 0; h->d_carry = nullptr; h->d_carry_cap = 0;
  for (void *q : fw) if (q) (void)hipFree(q);
  if (h->h_fcounts) (void)hipHostFree(h->h_fcounts);
  h->d_keys = h->d_keys_alt = nullptr; h->d_ctemp = nullptr; h->d_fout = h->d_fleft = nullptr; h->d_fcounts = nullptr;
  h->h_fcounts = nullptr; h->d_fpat_len = nullptr; h->d_fpat_id = nullptr; h->ckeys_cap = 0; h->ctemp_bytes = 0;
  h->d_cands = nullptr; h->d_counter = nullptr; h->h_counter = nullptr; h->d_text = nullptr;
  h->ev0 = h->ev1 = nullptr; h->d_wstart = nullptr; h->d_wlen = nullptr; h->d_woff = nullptr; h->d_wout = nullptr;
  h->d_wcap = h->d_woutcap = 0;
}

static int ensure_capacity(pm_handle *h, size_t cap, bool exact = false) {
  if (h->d_cands && (exact ? h->cap == cap : h->cap >= cap)) return PM_OK;
  if (h->d_cands) (void)hipFree(h->d_cands);
  h->d_cands = nullptr;
  HIP_TRY(h, hipMalloc((void **)&h->d_cands, cap * sizeof(pm_hit)));
  if (h->d_ext) { (void)hipFree(h->d_ext); h->d_ext = nullptr; }
  h->cap = cap;
  return PM_OK;
}

// Decide the reference engine to reproduce and derive the inner (device) pattern set from it.
static int resolve(pm_handle *h) {
  const int np = (int)h->pats.size();
  std::vector<int32_t> len(np), esb(np), eeb(np);
  for (int i = 0; i < np; ++i) { len[i] = (int)h->pats[i].s.size(); esb[i] = h->pats[i].esb; eeb[i] = h->pats[i].eeb; }
  const Alphabet &A = h->alpha;
  const bool acgt = A.nch['A'] == 0 && A.nch['C'] == 1 && A.nch['G'] == 2 && A.nch['T'] == 3;
  int sem = h->cfg.semantics;
  // the "edits >= inexact bases" check runs for every -N (select.cc:87-90)
  const int autosem = pm_pick_semantics(A.size, acgt, h->cfg.k, h->cfg.wildcards, np, len.data(), esb.data(), eeb.data());
  if (autosem == PM_E_FATAL)
    return fail(h, PM_E_FATAL, "Fatal error: Number of edits >= Minimum number of inexact bases");
  if (sem == PM_SEM_AUTO) sem = autosem;
  switch (sem) {
    case 1: case 2: case 3: sem = PM_SEM_KEYWORD_TREE; break;
    case 4: break;
    case 5: break;
    case 7: case 8: case 9: case 10: sem = PM_SEM_EXACT_BASES; break;
    case 11: case 12: case 13: case 14: sem = PM_SEM_EXACT_HALVES; break;
    case PM_SEM_SHIFT_AND_INEXACT: break;
    default: return fail(h, PM_E_INVALID, "unknown semantics selector");
  }
  h->sem = sem;
  h->inner.clear(); h->inner_ids.clear();
  h->scan_k = 0; h->scan_indels = false;
  switch (sem) {
    case PM_SEM_KEYWORD_TREE: case PM_SEM_SHIFT_AND:
      for (const Pattern &p : h->pats) { h->inner.push_back(p); h->inner_ids.push_back((uint32_t)p.id); }
      break;
    case PM_SEM_SHIFT_AND_INEXACT:
      for (const Pattern &p : h->pats) { h->inner.push_back(p); h->inner_ids.push_back((uint32_t)p.id); }
      h->scan_k = h->cfg.k; h->scan_indels = h->cfg.indels != 0;
      break;
    case PM_SEM_FILTER_BITVEC:                                      // filter_bitvec.cc:185-196
      for (int i = 0; i < np; ++i) { h->inner.push_back(h->pats[i]); h->inner_ids.push_back((uint32_t)(i + 1)); }
      h->scan_k = h->cfg.k; h->scan_indels = h->cfg.indels != 0;
      break;
    case PM_SEM_EXACT_HALVES:                                       // exact_halves.cc:199-224
      for (int i = 0; i < np; ++i) {
        const std::string &s = h->pats[i].s;
        const size_t l1 = s.size() / 2;
        if (l1 == 0) return fail(h, PM_E_UNSUPPORTED, "exact_halves needs patterns of length >= 2");
        h->inner.push_back(Pattern{s.substr(0, l1), 0, 0, 0}); h->inner_ids.push_back((uint32_t)(2 * i + 1));
        h->inner.push_back(Pattern{s.substr(l1), 0, 0, 0});    h->inner_ids.push_back((uint32_t)(2 * i + 2));
      }
      h->lasthit.assign(np + 1, 0);
      h->halves_fresh = true;
      break;
    case PM_SEM_EXACT_BASES:                                        // exact_bases.cc:131-160
      for (int i = 0; i < np; ++i) {
        const Pattern &p = h->pats[i];
        const int L = (int)p.s.size();
        const int cut = p.esb >= p.eeb ? p.esb : p.eeb;
        if (cut <= 0 || cut > L) return fail(h, PM_E_UNSUPPORTED, "exact_bases needs 0 < exact bases <= pattern length");
        h->inner.push_back(Pattern{p.esb >= p.eeb ? p.s.substr(0, p.esb) : p.s.substr(L - p.eeb), 0, 0, 0});
        h->inner_ids.push_back((uint32_t)(i + 1));
      }
      break;
  }
  h->kern = h->cfg.kernel;
  return PM_OK;
}

// -w / -W on the seed family.  The seed kernels compare A,C,G,T exactly, so an ambiguity letter of a
// primer is expanded at table build: the primer becomes its concrete variants (same id), and the
// smallest distance over the variants is the wildcard distance (shift_and.cc:108-117: a pattern
// character accepts every stream letter of its IUPAC compatibility set).  That is only the whole story
// while the STREAM holds nothing but A,C,G,T (and N without -W): a stream letter R would match a
// primer's A under -w.
static std::string acgt_of(unsigned char ch) {                      // the A,C,G,T a pattern character accepts under -w
  std::string r;
  const char *set = iupac_compatible_set(ch);
  if (!set) return r;
  for (const char *q = set; *q; ++q) if (*q == 'A' || *q == 'C' || *q == 'G' || *q == 'T') r.push_back(*q);
  return r;
}
static bool stream_letters_plain(const pm_handle *h) {
  for (int c = 0; c < h->alpha.size && c < 256; ++c) {
    if (!h->alpha.present[c] || c == h->eos_code) continue;
    const unsigned char l = h->alpha.ch[c];
    if (l == 'A' || l == 'C' || l == 'G' || l == 'T') continue;
    if (l == 'N' && !h->cfg.text_n) continue;                       // a text N matches nothing without -W
    return false;
  }
  return true;
}
// The one letter pair on which the reference's automaton and its verify DP disagree: pattern N at a stream N under -w
// without -W.  The mask build leaves the stream's N out of every class (shift_and.cc:112: `wccompat[j]!='N' || _textn`),
// so the automaton counts a substitution there; the DP looks for EQUAL characters first (pattern_alignment.cc:314-316)
// and charges nothing.  filter_bitvec's hit value is the DP's, so for such a pattern it cannot be read off the
// candidate levels (found by tests/test_gpu_adversarial.py against the oracle; both kernel families shared the shortcut).
static bool stream_has_N(const pm_handle *h) {
  const int c = h->alpha.nch[(unsigned char)'N'];
  return c >= 0 && c < 256 && h->alpha.present[c];
}
static bool pattern_n_quirk(const pm_handle *h, const Pattern &p) {
  return h->cfg.wildcards && !h->cfg.text_n && p.s.find('N') != std::string::npos && stream_has_N(h);
}
// concrete variants of a primer (at most `cap`; empty + false when there would be more)
static bool expand_iupac(const std::string &p, size_t cap, std::vector<std::string> *out) {
  out->assign(1, std::string());
  for (unsigned char ch : p) {
    const std::string alts = acgt_of(ch);
    if (alts.empty()) { out->clear(); return true; }                // accepts no base at all: the primer cannot match
    if (out->size() * alts.size() > cap) { out->clear(); return false; }
    std::vector<std::string> next;
    next.reserve(out->size() * alts.size());
    for (const std::string &pre : *out) for (char a : alts) next.push_back(pre + a);
    out->swap(next);
  }
  return true;
}

// Inner pattern set of the seed family: Hamming candidates of the WHOLE patterns; the wrappers'
// rules are then applied to (end, pattern, distance, clean-half flags) records on the host.
static bool seed_eligible(pm_handle *h, std::string *why) {
  const int sem = h->sem;
  if (h->cfg.wildcards && !h->wild_seed) { *why = "IUPAC wildcards run on the bit-parallel family (the seed family takes them on A,C,G,T streams for the exact engines and filter_bitvec)"; return false; }
  if (h->cfg.k > 0 && h->cfg.indels && sem == PM_SEM_EXACT_HALVES) {
    // exact_halves with edits: exact seeds of the halves + partner prefilter on the GPU, DP on the host
    if (h->cfg.k > 2) { *why = "exact_halves -k > 2 runs on the bit-parallel family"; return false; }
    for (const Pattern &p : h->pats) if (p.s.size() > 32 || p.s.size() < 16) { *why = "exact_halves -k on the seed family needs 16..32 character patterns"; return false; }
    return true;
  }
  if (h->cfg.k > 0 && h->cfg.indels && (sem == PM_SEM_FILTER_BITVEC || sem == PM_SEM_SHIFT_AND_INEXACT)) {
    // the automaton's candidates from displaced-piece seeds + a per-seed automaton run (pm_seed.hip, EDITS)
    if (h->cfg.k > 2) { *why = "edit distance > 2 runs on the bit-parallel family"; return false; }
    for (size_t i = 0; i < h->pats.size(); ++i) {
      if (i < h->in_rest.size() && h->in_rest[i]) continue;         // goes to the bit-parallel residue
      const Pattern &p = h->pats[i];
      if (p.s.size() > 32 || p.s.size() < 20) { *why = "the edit-distance seed plan needs 20..32 character patterns"; return false; }
    }
    for (uint32_t id : h->inner_ids) if (id >= (1u << 22)) { *why = "the edit-distance seed plan packs pattern ids into 22 bits"; return false; }
    return true;
  }
  if (h->cfg.k > 0 && h->cfg.indels && sem == PM_SEM_EXACT_BASES) {
    // every exact_bases hit is a window within k edits of the whole pattern: the edit-distance seed plan
    // finds those, pm_bases_seeds turns them into the reference's block seeds (pm_cluster.hip)
    if (h->cfg.k > 2) { *why = "edit distance > 2 runs on the bit-parallel family"; return false; }
    if (h->pats.size() >= ((size_t)1 << 22)) { *why = "the edit-distance seed plan packs pattern ids into 22 bits"; return false; }
    for (const Pattern &p : h->pats) {
      if (p.s.size() > 32 || p.s.size() < 20) { *why = "the edit-distance seed plan needs 20..32 character patterns"; return false; }
      for (unsigned char ch : p.s) if (!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T')) { *why = "pattern with characters other than A,C,G,T"; return false; }
    }
    return true;
  }
  if (h->cfg.k > 0 && h->cfg.indels && sem != PM_SEM_KEYWORD_TREE && sem != PM_SEM_SHIFT_AND) { *why = "edit-distance search (-k) runs on the bit-parallel family"; return false; }
  // substitution-only search of 20..32 character A,C,G,T patterns with k = 1, 2 runs on the pair plan
  // (pm_pair.hip), whose exact verify knows the patterns' exact zones: a substitution there fails
  // the reference's constrained verifies (pattern_alignment.cc:320-323, primer_alignment.cc:155)
  bool pair_ok = !h->cfg.indels && (h->cfg.k == 1 || h->cfg.k == 2) && !h->pats.empty();
  for (const Pattern &p : h->pats) {
    pair_ok = pair_ok && p.s.size() >= 20 && p.s.size() <= 32;
    for (unsigned char ch : p.s) pair_ok = pair_ok && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
  }
  if (h->knobs.pair >= 0) pair_ok = pair_ok && h->knobs.pair != 0;
  if (sem == PM_SEM_EXACT_BASES && !pair_ok) { *why = "exact_bases runs on the bit-parallel family (the seed family takes it for -K 1 / -K 2 on 20..32 character patterns)"; return false; }
  if ((sem == PM_SEM_FILTER_BITVEC || sem == PM_SEM_EXACT_HALVES) && !pair_ok)
    for (const Pattern &p : h->pats) if (p.esb || p.eeb) { *why = "exact-base constraints need the text-based verify of the bit-parallel family"; return false; }
  return true;
}

static int init_common(pm_handle *h, const uint8_t *table, int32_t table_len) {
  drain_spec(h);
  h->land_n = h->land_pos = 0;
  if (table) {
    if (table_len <= 0 || table_len > 256) return fail(h, PM_E_INVALID, "bad alphabet table length");
    h->alpha.set_table(table, table_len);
  } else h->alpha.set_raw();
  // -w/-W on a raw stream: IUPAC classes name up to 16 letters each, the kernels keep 6 character
  // classes in registers -- only the letters that occur in the stream need one
  if (!table && h->cfg.wildcards) {
    if (h->host_only) {                                             // no device copy: look at the host bytes
      for (int i = 0; i < 256; ++i) h->alpha.present[i] = false;
      for (int64_t i = 0; i < h->n; ++i) h->alpha.present[h->h_text[i]] = true;
    } else HIP_TRY(h, stream_presence(h->d_text, h->n, h->alpha.present, h->stream));
  }
  h->eos_code = h->alpha.nch[(uint8_t)h->cfg.eos];                  // shift_and_inexact.cc:131
  int rc = resolve(h);
  if (rc) return rc;
  bitpar_free(&h->bp); seed_free(&h->sd);
  for (SeedDevice &d : h->sd_more) seed_free(&d);
  h->sd_more.clear();
  for (PairDevice &d : h->pair) pair_free(&d);
  h->pair.clear();
  pair_free(&h->epair); h->epair_on = false;
  // tables pm_finalize_device builds on first use depend on the alphabet mapping and the pattern
  // list of THIS init: drop the ones of an earlier init
  { void *lazy[] = {h->d_dp_codes, h->d_dp_esb, h->d_dp_eeb, h->d_fpat_len, h->d_fpat_id}; for (void *q : lazy) if (q) (void)hipFree(q); }
  h->d_dp_codes = nullptr; h->d_dp_esb = h->d_dp_eeb = nullptr; h->d_fpat_len = nullptr; h->d_fpat_id = nullptr;
  h->d_final = nullptr; h->n_final = 0;
  h->seed_flags = false; h->bases_flags = false; h->bases_edits = false; h->half_ranked_any = false;
  h->zoned = false;
  for (const Pattern &p : h->pats) h->zoned = h->zoned || p.esb || p.eeb;
  h->nn_quirk = false;
  for (const Pattern &p : h->pats) h->nn_quirk = h->nn_quirk || pattern_n_quirk(h, p);
  h->start_cached = false; h->start_cache.clear(); h->end_cached = false; h->end_cache.clear(); h->overhang_cached = false; h->overhang_cache.clear(); h->edge_cached = false; h->edge_cache.clear(); h->head_cached = false; h->head_cache.clear();
  std::string why;
  bool want_seed = h->kern == PM_KERNEL_SEED || h->kern == PM_KERNEL_AUTO;
  // A pattern set is rarely uniform: a few primers with an ambiguity letter, one that is too short
  // or too long for the seed plan.  Where the engines' records mean the same (one inner pattern per
  // pattern: keyword_tree, shift_and, shift_and_inexact, filter_bitvec) those few are scanned by
  // the bit-parallel kernels into the same record buffer and the rest stays on the seed family.
  h->in_rest.assign(h->inner.size(), 0);
  h->nrest = 0;
  const bool edits_plan = h->cfg.indels && h->cfg.k > 0 && (h->sem == PM_SEM_FILTER_BITVEC || h->sem == PM_SEM_SHIFT_AND_INEXACT);
  h->wild_seed = h->cfg.wildcards && want_seed && h->kern == PM_KERNEL_AUTO && stream_letters_plain(h) &&
                 (h->sem == PM_SEM_KEYWORD_TREE || h->sem == PM_SEM_SHIFT_AND || h->sem == PM_SEM_FILTER_BITVEC);
  if (want_seed && (!h->cfg.wildcards || h->wild_seed) && h->kern == PM_KERNEL_AUTO &&
      (h->sem == PM_SEM_KEYWORD_TREE || h->sem == PM_SEM_SHIFT_AND || h->sem == PM_SEM_SHIFT_AND_INEXACT || h->sem == PM_SEM_FILTER_BITVEC)) {
    std::vector<std::string> variants;
    for (size_t i = 0; i < h->inner.size(); ++i) {
      const std::string &ps = h->inner[i].s;
      bool ok = ps.size() <= 32 && ps.size() >= (edits_plan ? 20u : 10u);
      if (h->wild_seed) ok = ok && expand_iupac(ps, 16, &variants) &&   // up to two ambiguity letters (16 variants)
                         !(h->sem == PM_SEM_FILTER_BITVEC && pattern_n_quirk(h, h->inner[i]));   // (its value needs the text-based verify)
      else for (unsigned char ch : ps) ok = ok && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
      if (!ok) { h->in_rest[i] = 1; ++h->nrest; }
    }
    if (h->nrest == h->inner.size()) { std::fill(h->in_rest.begin(), h->in_rest.end(), 0); h->nrest = 0; }   // nothing for the seed family: one engine
  }
  if (want_seed && !seed_eligible(h, &why)) {
    if (h->kern == PM_KERNEL_SEED) return fail(h, PM_E_UNSUPPORTED, "seed engine: " + why);
    want_seed = false;
  }
  if (want_seed) {
    std::vector<Pattern> sp; std::vector<uint32_t> sid;
    int sk = h->scan_k;
    std::vector<std::string> partners;
    std::vector<uint8_t> sides;
    const bool halves_mode = h->sem == PM_SEM_EXACT_HALVES && h->cfg.indels && h->cfg.k > 0;
    const bool edits_mode = h->cfg.indels && h->cfg.k > 0 && (h->sem == PM_SEM_FILTER_BITVEC || h->sem == PM_SEM_SHIFT_AND_INEXACT || h->sem == PM_SEM_EXACT_BASES);
    if (halves_mode) {                            // halves as exact patterns (ids 2j+1, 2j+2), partner = other half
      sp = h->inner; sid = h->inner_ids; sk = 0;
      for (size_t i = 0; i + 1 < sp.size(); i += 2) {
        partners.push_back(sp[i + 1].s); sides.push_back(0);      // left half: partner to the right
        partners.push_back(sp[i].s); sides.push_back(1);          // right half: partner to the left
      }
    } else if (h->sem == PM_SEM_EXACT_HALVES) {   // -K: whole patterns, distance <= k, halves decided by flags
      for (size_t i = 0; i < h->pats.size(); ++i) { sp.push_back(h->pats[i]); sid.push_back((uint32_t)(i + 1)); }
      sk = h->cfg.k;
    } else if (h->sem == PM_SEM_EXACT_BASES) {
      // -K: an occurrence of the mandated exact block whose remainder verifies (exact_bases.cc:92-121, the
      // extension DP with indels off walks the diagonal) = a whole-pattern window within k substitutions,
      // none of them in an exact zone; one hit per occurrence, no clustering: the records are final
      // (-k: the records in between are the automaton's candidates of pattern i + 1, see pm_bases_seeds)
      for (size_t i = 0; i < h->pats.size(); ++i) { sp.push_back(h->pats[i]); sid.push_back(h->cfg.indels ? (uint32_t)(i + 1) : (uint32_t)h->pats[i].id); }
      sk = h->cfg.k;
    } else if (h->wild_seed) {                    // every concrete variant of a primer is a seed-family pattern with the primer's id
      std::vector<std::string> variants;
      for (size_t i = 0; i < h->inner.size(); ++i) {
        if (h->in_rest[i]) continue;
        expand_iupac(h->inner[i].s, 16, &variants);
        for (const std::string &v : variants) { Pattern q = h->inner[i]; q.s = v; sp.push_back(q); sid.push_back(h->inner_ids[i]); }
      }
      if (sp.empty()) { sp.push_back(Pattern{std::string(edits_mode ? 20 : 10, 'A'), 0, 0, 0}); sid.push_back(0); why = "no primer the seed family could take"; }
    } else if (h->nrest) {
      for (size_t i = 0; i < h->inner.size(); ++i) if (!h->in_rest[i]) { sp.push_back(h->inner[i]); sid.push_back(h->inner_ids[i]); }
    } else { sp = h->inner; sid = h->inner_ids; }
    // the automata and the keyword tree know nothing of exact_start_bases / exact_end_bases (shift_and_inexact.cc:85-86
    // stores them and never looks): their records are every window within k, whatever zone a substitution falls in
    if (h->sem == PM_SEM_SHIFT_AND_INEXACT || h->sem == PM_SEM_SHIFT_AND || h->sem == PM_SEM_KEYWORD_TREE)
      for (Pattern &p : sp) { p.esb = 0; p.eeb = 0; }
    // One LDS Bloom filter (1 Mbit) stays selective up to ~256k keys: larger pattern sets are cut
    // into tiles with their own tables; the scan launches once per tile into the same record buffer.
    size_t tile_keys = 262144;
    // the plan (window, pieces) must be the same for every tile: it depends on the shortest pattern
    // of the whole set, so every tile is built with that window forced
    int force_lw = 0;
    for (const Pattern &p : sp) force_lw = force_lw == 0 ? (int)p.s.size() : std::min(force_lw, (int)p.s.size());
    // halves of <= 10 bases: the filter is the exact bitmap of their keys (SeedArgs::exact_filter) and
    // has no capacity to exceed -- one pass over the stream for all of them
    if (halves_mode && force_lw > 0 && force_lw <= 10) tile_keys = (size_t)1 << 21;
    // halves of >= 10 bases go through the ranked plan (pm_half_scan): its key bitmap has no capacity either, and a
    // tile of 2^20 halves keeps the "further halves of a key" index inside the slot's 20 bits
    if (halves_mode && force_lw >= 10 && !h->knobs.half_bloom) {
      size_t fits = 0;                                 // seed_build's rule: the ranked plan when >= 90 % of the halves fit its partner test
      for (size_t i = 0; i < sp.size(); ++i) {
        const int L = (int)sp[i].s.size(), plen = (int)partners[i].size();
        if (sides[i] ? L + plen + h->cfg.k <= 32 : plen + h->cfg.k <= 16) ++fits;
      }
      if (fits * 10 >= sp.size() * 9) tile_keys = (size_t)1 << 20;
    }
    if (h->knobs.seed_tile > 0) tile_keys = (size_t)h->knobs.seed_tile;
    const size_t ntile = sp.empty() ? 1 : (sp.size() + tile_keys - 1) / tile_keys;
    size_t per = (sp.size() + ntile - 1) / ntile;
    if (halves_mode) per += per & 1;                // keep (left, right) half pairs together: side = index parity
    // Substitution-only search with k = 1, 2 on patterns of 20..32 characters: the pair plan (exact
    // 20-bit key bitmaps, pm_pair.hip) replaces the Bloom-filter plan of pm_seed.hip
    bool use_pair = !halves_mode && !edits_mode && (sk == 1 || sk == 2) && !sp.empty() && force_lw >= 20;
    if (h->knobs.pair >= 0) use_pair = use_pair && h->knobs.pair != 0;
    for (size_t ti = 0; ti < ntile && why.empty() && use_pair; ++ti) {
      const size_t lo = ti * per, hi = std::min(sp.size(), lo + per);
      std::vector<Pattern> tp(sp.begin() + lo, sp.begin() + hi);
      std::vector<uint32_t> tid(sid.begin() + lo, sid.begin() + hi);
      PairTables pt;
      const std::string msg = pair_build(tp, tid, h->alpha, sk, h->eos_code, &pt, h->knobs.pair_row);
      if (!msg.empty()) {                                            // not for this set: the seed plan below takes it
        for (PairDevice &d : h->pair) pair_free(&d);
        h->pair.clear();
        use_pair = false;
        break;
      }
      h->pair.emplace_back();
      HIP_TRY(h, pair_upload(pt, &h->pair.back(), h->stream));
      h->pair.back().knobs = h->knobs;
      h->pair.back().viol_level = h->sem == PM_SEM_FILTER_BITVEC ? 3 : 0;   // filter_bitvec chains every candidate; the others drop zone violations
      // the plan facts the rest of this file reads from h->sd (no device tables behind them)
      h->sd.k = sk; h->sd.Lw = 20; h->sd.pb = 5; h->sd.r = 4 - sk; h->sd.ncombos = pt.ncombos; h->sd.ascii = pt.ascii;
      h->sd.maxlen = std::max(h->sd.maxlen, pt.maxlen);
    }
    for (size_t ti = 0; ti < ntile && why.empty() && !use_pair; ++ti) {
      const size_t lo = ti * per, hi = std::min(sp.size(), lo + per);
      std::vector<Pattern> tp(sp.begin() + lo, sp.begin() + hi);
      std::vector<uint32_t> tid(sid.begin() + lo, sid.begin() + hi);
      std::vector<std::string> tpart;
      std::vector<uint8_t> tside;
      if (halves_mode) { tpart.assign(partners.begin() + lo, partners.begin() + hi); tside.assign(sides.begin() + lo, sides.begin() + hi); }
      SeedTables st;
      why = seed_build(tp, tid, h->alpha, sk, h->eos_code, &st, force_lw, halves_mode ? &tpart : nullptr,
                       halves_mode ? &tside : nullptr, h->cfg.k, edits_mode, h->knobs);
      if (why.empty() && h->kern == PM_KERNEL_AUTO && st.Lw < (halves_mode ? 8 : 10)) why = "patterns too short for the seed family";
      if (!why.empty()) break;
      SeedDevice *dst = &h->sd;
      if (ti > 0) { h->sd_more.emplace_back(); dst = &h->sd_more.back(); }
      HIP_TRY(h, seed_upload(st, dst, h->stream));
      dst->knobs = h->knobs;
      dst->maxlen = std::max(dst->maxlen, h->sd.maxlen);
    }
    if (!why.empty()) {
      seed_free(&h->sd);
      for (SeedDevice &d : h->sd_more) seed_free(&d);
      h->sd_more.clear();
      if (h->kern == PM_KERNEL_SEED) return fail(h, PM_E_UNSUPPORTED, "seed engine: " + why);
      want_seed = false;
    } else {
      if (h->nrest) {
        std::vector<Pattern> rp; std::vector<uint32_t> rid;
        for (size_t i = 0; i < h->inner.size(); ++i) if (h->in_rest[i]) { rp.push_back(h->inner[i]); rid.push_back(h->inner_ids[i]); }
        BitparTables tabs;
        std::string msg = bitpar_build(rp, rid, h->alpha, h->scan_k, h->eos_code, &tabs, h->cfg.wildcards != 0, h->cfg.text_n != 0);
        if (!msg.empty()) return fail(h, PM_E_UNSUPPORTED, "bit-parallel engine (residue): " + msg);
        HIP_TRY(h, bitpar_upload(tabs, h->scan_indels, &h->bp, h->stream));
        h->bp.knobs = h->knobs;
      }
      int mx = h->sd.maxlen;
      for (SeedDevice &d : h->sd_more) mx = std::max(mx, d.maxlen);
      h->sd.maxlen = mx;
      h->kern = PM_KERNEL_SEED;
      h->seed_flags = h->sem == PM_SEM_EXACT_HALVES && !halves_mode;
      h->bases_flags = h->sem == PM_SEM_EXACT_BASES && !h->cfg.indels;
      h->bases_edits = h->sem == PM_SEM_EXACT_BASES && h->cfg.indels;
      h->halves_dev = halves_mode;
      h->half_ranked_any = h->sd.half_ranked;
      for (const SeedDevice &d : h->sd_more) h->half_ranked_any = h->half_ranked_any || d.half_ranked;
      h->edits_dev = edits_mode;
      // -k 2 on one pattern tile: the first stage runs on the PAIR geometry (pm_pair.hip, edit plan: 14 field-pair tests per
      // window instead of 34 hashed piece triples; round 4) -- its tables are the substitution plan's with two patterns per
      // slot, over the same pattern list as the automaton records of h->sd (seed records carry indices into it)
      if (edits_mode && h->cfg.k == 2 && !h->bases_edits && h->sd_more.empty() && !h->knobs.edit_bloom && !h->knobs.edit_hash && !sp.empty()) {
        PairTables pt;
        const std::string msg = pair_build(sp, sid, h->alpha, 2, h->eos_code, &pt, h->knobs.pair_row, 2);
        if (msg.empty()) {
          HIP_TRY(h, pair_upload(pt, &h->epair, h->stream));
          h->epair.knobs = h->knobs;
          h->epair_on = true;
        }
      }
      if (halves_mode) {
        const size_t nh = h->inner.size();
        std::vector<uint8_t> codes(nh * 16, 0), lens(nh, 0);
        for (size_t i = 0; i < nh; ++i) {
          lens[i] = (uint8_t)h->inner[i].s.size();
          for (size_t q = 0; q < h->inner[i].s.size() && q < 16; ++q) codes[i * 16 + q] = (uint8_t)h->alpha.nch[(unsigned char)h->inner[i].s[q]];
        }
        std::vector<int32_t> es(h->pats.size()), ee(h->pats.size());
        for (size_t i = 0; i < h->pats.size(); ++i) { es[i] = h->pats[i].esb; ee[i] = h->pats[i].eeb; }
        { void *hx[] = {h->d_half_codes, h->d_half_len, h->d_hesb, h->d_heeb}; for (void *q : hx) if (q) (void)hipFree(q); }
        HIP_TRY(h, hipMalloc((void **)&h->d_half_codes, codes.size() ? codes.size() : 16));
        HIP_TRY(h, hipMalloc((void **)&h->d_half_len, lens.size() ? lens.size() : 16));
        HIP_TRY(h, hipMalloc((void **)&h->d_hesb, es.size() ? es.size() * 4 : 16));
        HIP_TRY(h, hipMalloc((void **)&h->d_heeb, ee.size() ? ee.size() * 4 : 16));
        if (nh) {
          HIP_TRY(h, hipMemcpy(h->d_half_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
          HIP_TRY(h, hipMemcpy(h->d_half_len, lens.data(), lens.size(), hipMemcpyHostToDevice));
          HIP_TRY(h, hipMemcpy(h->d_hesb, es.data(), es.size() * 4, hipMemcpyHostToDevice));
          HIP_TRY(h, hipMemcpy(h->d_heeb, ee.data(), ee.size() * 4, hipMemcpyHostToDevice));
        }
      }
      h->seed_k = sk;
    }
  }
  if (!want_seed) {
    h->halves_dev = false; h->edits_dev = false; h->half_ranked_any = false;
    std::fill(h->in_rest.begin(), h->in_rest.end(), 0); h->nrest = 0;
    h->kern = PM_KERNEL_BITPAR;
    BitparTables tabs;
    std::string msg = bitpar_build(h->inner, h->inner_ids, h->alpha, h->scan_k, h->eos_code, &tabs,
                                   h->cfg.wildcards != 0, h->cfg.text_n != 0);
    if (!msg.empty()) return fail(h, PM_E_UNSUPPORTED, "bit-parallel engine: " + msg);
    HIP_TRY(h, bitpar_upload(tabs, h->scan_indels, &h->bp, h->stream));
    h->bp.knobs = h->knobs;
  }
  if (!h->d_counter) HIP_TRY(h, hipMalloc((void **)&h->d_counter, sizeof(unsigned long long)));
  if (!h->h_counter) HIP_TRY(h, hipHostMalloc((void **)&h->h_counter, sizeof(unsigned long long), hipHostMallocDefault));
  if (!h->h_seed_count) { HIP_TRY(h, hipHostMalloc((void **)&h->h_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long), hipHostMallocDefault)); memset(h->h_seed_count, 0, SEEDCOUNT_WORDS * sizeof(unsigned long long)); }
  if (!h->ev0) HIP_TRY(h, hipEventCreate(&h->ev0));
  if (!h->ev1) HIP_TRY(h, hipEventCreate(&h->ev1));
  if (!h->host_only) {
    if (!h->ev_fin) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fin, hipEventDisableTiming));
    if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  }
  if (!h->d_cands) { rc = ensure_capacity(h, (size_t)1 << 20); if (rc) return rc; }
  if (h->edits_dev && !h->host_only) {                              // every candidate is reported by several seeds before the dedup
    const size_t want = std::min<size_t>(std::max<size_t>((size_t)(h->n / 24), (size_t)1 << 22), (size_t)1 << 28);
    if (h->cap < want) { rc = ensure_capacity(h, want); if (rc) return rc; }
  }
  device_sort_plan(h);
  // pm_scan's landing buffer for a database-sized stream: pinned now, beside the upload, rather than inside the first range
  // (2^20 records = 16 MiB take ~5 ms to pin; a 1 GiB range of uniform text hands back 3.3e5 -K 2 hits, 1.2e6 with -k 2)
  if (!h->host_only && h->n >= ((int64_t)1 << 28) && h->land_cap < ((size_t)1 << 20)) { rc = ensure_landing(h, (size_t)1 << 20); if (rc) return rc; }
  h->internal_rescans = 0; h->range_splits = 0; h->dense_mode = false; h->piece_len = 0;
  h->inited = true;
  return pm_reset(h);
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Host stream -> HBM.  A single hipMemcpy from pageable memory runs at the speed of one core's copy
// into the runtime's staging buffer (≈11 GB/s here); several threads, each staging its slice
// through two pinned buffers of its own on its own HIP stream, fill the PCIe link instead.
static hipError_t upload_stream(int device, void *d, const uint8_t *text, size_t n) {
  constexpr size_t CH = (size_t)8 << 20;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int T = (int)std::min<size_t>(std::min<unsigned>(6u, hw), (n + 4 * CH - 1) / (4 * CH));
  if (T <= 1) return hipMemcpy(d, text, n, hipMemcpyHostToDevice);
  std::vector<hipError_t> errs((size_t)T, hipSuccess);
  std::vector<std::thread> th;
  const size_t per = ((n + (size_t)T - 1) / (size_t)T + CH - 1) / CH * CH;
  for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
    hipError_t e = hipSetDevice(device);
    const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
    hipStream_t st = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
      e = hipHostMalloc(&pin[b], CH, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
    }
    int b = 0;
    for (size_t off = lo; off < hi && e == hipSuccess; off += CH, b ^= 1) {
      const size_t len = std::min(CH, hi - off);
      e = hipEventSynchronize(ev[b]);                               // the copy that last used this buffer is done
      if (e != hipSuccess) break;
      memcpy(pin[b], text + off, len);
      e = hipMemcpyAsync((char *)d + off, pin[b], len, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipEventRecord(ev[b], st);
    }
    if (st) { const hipError_t e2 = hipStreamSynchronize(st); if (e == hipSuccess) e = e2; }
    for (int q = 0; q < 2; ++q) { if (ev[q]) (void)hipEventDestroy(ev[q]); if (pin[q]) (void)hipHostFree(pin[q]); }
    if (st) (void)hipStreamDestroy(st);
    errs[(size_t)t] = e;
  });
  for (std::thread &x : th) x.join();
  for (hipError_t e : errs) if (e != hipSuccess) return e;
  return hipSuccess;
}

// The seed family's first stage reads the stream at 2 bits per base (pm_seed.hip pack_stream): one
// pass over the stream per pm_init, on the handle's stream, after the bytes are in HBM.
static int ensure_packed(pm_handle *h) {
  if (h->kern != PM_KERNEL_SEED) return PM_OK;
  const size_t words = (size_t)((h->n + 15) / 16);
  if (!h->d_packed || h->packed_cap < words) {
    if (h->d_packed) (void)hipFree(h->d_packed);
    h->d_packed = nullptr;
    h->packed_cap = words;
    HIP_TRY(h, hipMalloc((void **)&h->d_packed, (words + 128) * sizeof(uint32_t)));   // + padding the scan kernels' block prefetch may read (zeroed below)
  }
  if (h->n >= 16) HIP_TRY(h, pack_stream(h->d_text, 16, h->sd.ascii, h->d_packed + words + 64, 1, h->stream));   // the kernel's code is resident before the timed pass
  HIP_TRY(h, hipStreamSynchronize(h->stream));                      // table uploads of this init are not part of the figure
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->d_packed + words, 0, 128 * sizeof(uint32_t), h->stream));
  HIP_TRY(h, pack_stream(h->d_text, h->n, h->sd.ascii, h->d_packed, (int64_t)words, h->stream));
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  (void)hipEventElapsedTime(&h->pack_ms, h->ev0, h->ev1);
  return PM_OK;
}

extern "C" int pm_init(pm_handle *h, const uint8_t *text, int64_t n, const uint8_t *table, int32_t table_len) {
  if (!h || (!text && n > 0) || n < 0) return fail(h, PM_E_INVALID, "pm_init: bad arguments");
  const double ti0 = now_ms();
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (h->own_d_text && h->d_text) { (void)hipFree((void *)h->d_text); h->d_text = nullptr; }
  void *d = nullptr;
  HIP_TRY(h, hipMalloc(&d, (size_t)(n > 0 ? n : 1) + 16));
  const double ti1 = now_ms();
  h->d_text = (const uint8_t *)d; h->own_d_text = true; h->h_text = text; h->n = n; h->stream = nullptr; h->host_only = false;
  // the stream crosses PCIe (≈0.3 s for 3 GB of pageable memory) while this thread builds the
  // pattern tables; raw streams with wildcards look at the stream on the device first, so they wait
  const bool overlap = n > ((int64_t)1 << 24) && !(!table && h->cfg.wildcards);
  hipError_t copy_err = hipSuccess;
  std::thread copier;
  if (n > 0) {
    if (overlap) {
      const int dev = h->cfg.device;
      copier = std::thread([=, &copy_err]() {
        copy_err = hipSetDevice(dev);
        if (copy_err == hipSuccess) copy_err = upload_stream(dev, d, text, (size_t)n);
      });
    } else {
      HIP_TRY(h, hipMemcpy(d, text, (size_t)n, hipMemcpyHostToDevice));
    }
  }
  const int rc = init_common(h, table, table_len);
  const double ti2 = now_ms();
  if (copier.joinable()) copier.join();
  if (h->knobs.debug) fprintf(stderr, "[pm] init: runtime + stream buffer %.0f ms, tables %.0f ms, then %.0f ms more for the stream upload\n", ti1 - ti0, ti2 - ti1, now_ms() - ti2);
  if (copy_err != hipSuccess) return fail(h, PM_E_HIP, std::string("pm_init: stream upload: ") + hipGetErrorString(copy_err));
  if (rc) return rc;
  return ensure_packed(h);
}

extern "C" int pm_init_host(pm_handle *h, const uint8_t *text, int64_t n, const uint8_t *table, int32_t table_len) {
  if (!h || (!text && n > 0) || n < 0) return fail(h, PM_E_INVALID, "pm_init_host: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (h->own_d_text && h->d_text) (void)hipFree((void *)h->d_text);
  h->d_text = nullptr; h->own_d_text = false; h->h_text = text; h->n = n; h->stream = nullptr;
  h->host_only = true;
  return init_common(h, table, table_len);
}

extern "C" int pm_init_device(pm_handle *h, const void *d_text, int64_t n, const uint8_t *table, int32_t table_len,
                              void *hip_stream) {
  if (!h || (!d_text && n > 0) || n < 0) return fail(h, PM_E_INVALID, "pm_init_device: bad arguments");
  if (((uintptr_t)d_text) & 3) return fail(h, PM_E_INVALID, "pm_init_device: stream must be 4-byte aligned");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (h->own_d_text && h->d_text) (void)hipFree((void *)h->d_text);
  h->d_text = (const uint8_t *)d_text; h->own_d_text = false; h->h_text = nullptr; h->n = n; h->host_only = false;
  h->stream = (hipStream_t)hip_stream;
  const int rc = init_common(h, table, table_len);
  if (rc) return rc;
  return ensure_packed(h);
}

extern "C" int pm_set_capacity(pm_handle *h, size_t max_candidates) {
  if (!h || max_candidates == 0) return PM_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  drain_spec(h);
  return ensure_capacity(h, max_candidates, true);
}

extern "C" int pm_reset(pm_handle *h) {
  if (!h) return PM_E_INVALID;
  drain_spec(h);
  if (h->scan_pending) { (void)hipStreamSynchronize(h->stream); h->scan_pending = false; }
  h->carry.clear(); h->land_n = h->land_pos = 0; h->next_begin = 0;
  std::fill(h->lasthit.begin(), h->lasthit.end(), 0);
  h->halves_fresh = true;
  h->last_count = 0;
  return PM_OK;
}

extern "C" void pm_destroy(pm_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  free_device(h);
  delete h;
}

extern "C" const char *pm_last_error(const pm_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }
extern "C" int pm_selected_semantics(const pm_handle *h) { return h && h->inited ? h->sem : PM_E_INVALID; }
extern "C" int pm_selected_kernel(const pm_handle *h) { return h && h->inited ? h->kern : PM_E_INVALID; }

extern "C" int pm_describe(const pm_handle *h, char *buf, size_t buflen) {
  if (!h || !buf || !h->inited) return PM_E_INVALID;
  if (h->kern == PM_KERNEL_SEED && !h->pair.empty()) {
    snprintf(buf, buflen, "kernel=pm_pair_scan tiles=%d combos=%d fields=2-of-4 x 5 bases window=20 row_slots=%d chunk=%lld nchunks=%d grid=%d block=%d lds=%d",
             (int)h->pair.size(), h->pair[0].ncombos, h->pair[0].stride, (long long)h->geo.seg_len, h->geo.nseg, h->geo.blocks, h->geo.threads, PAIR_LDS_BYTES);
    if (h->nrest) {
      const size_t at = strlen(buf);
      if (at < buflen) snprintf(buf + at, buflen - at, " + %s for %zu patterns the seed plan does not take", bitpar_kernel_name(h->scan_k, h->scan_indels), h->nrest);
    }
  }
  else if (h->kern == PM_KERNEL_SEED && h->edits_dev && h->epair_on) {
    snprintf(buf, buflen, "kernel=pm_pair_edit_scan+pm_pair_edit_resolve+pm_edits_verify tiles=1 tests=14 fields=2-of-4 x 5 bases, displaced by |d| <= 2 window=20 row_slots=%d slot_patterns=2 chunk=%lld nchunks=%d grid=%d block=%d lds=%d",
             h->epair.stride, (long long)h->geo.seg_len, h->geo.nseg, h->geo.blocks, h->geo.threads, PAIR_LDS_BYTES);
    if (h->nrest) {
      const size_t at = strlen(buf);
      if (at < buflen) snprintf(buf + at, buflen - at, " + %s for %zu patterns the seed plan does not take", bitpar_kernel_name(h->scan_k, h->scan_indels), h->nrest);
    }
  }
  else if (h->kern == PM_KERNEL_SEED) {
    snprintf(buf, buflen, "kernel=%s tiles=%d combos=%d pieces=%d-of-%d x %d bases window=%d slots=%zu chunk=%lld nchunks=%d grid=%d block=%d lds=%d",
             h->sd.edits && h->sd.edit_tabulated ? "pm_edit_scan+pm_edits_verify" : h->sd.edits ? "pm_seed_scan+pm_edits_verify" :
             h->sd.halves && h->sd.half_ranked ? "pm_half_scan+pm_half_verify" : "pm_seed_scan", 1 + (int)h->sd_more.size(), h->sd.ncombos, h->sd.r, h->sd.k + h->sd.r, h->sd.pb, h->sd.Lw, h->sd.nslots, (long long)h->geo.seg_len, h->geo.nseg,
             h->geo.blocks, h->geo.threads, SEED_LDS_BYTES);
    if (h->nrest) {
      const size_t at = strlen(buf);
      if (at < buflen) snprintf(buf + at, buflen - at, " + %s for %zu patterns the seed plan does not take", bitpar_kernel_name(h->scan_k, h->scan_indels), h->nrest);
    }
  }
  else
    snprintf(buf, buflen, "kernel=%s tiles=%d lanes_per_tile=64 words_per_lane=%d seg_len=%lld nseg=%d grid=%d block=%d",
             bitpar_kernel_name(h->scan_k, h->scan_indels), h->bp.ntiles, BP_WPL, (long long)h->geo.seg_len, h->geo.nseg,
             h->geo.blocks, h->geo.threads);
  return PM_OK;
}

// ---- device stage ------------------------------------------------------------------------------
static int ensure_dp_tables(pm_handle *h);

extern "C" int pm_scan_candidates_async(pm_handle *h, int64_t begin, int64_t end) {
  if (!h || !h->inited) return fail(h, PM_E_INVALID, "pm_scan_candidates: handle not initialised");
  if (h->host_only) return fail(h, PM_E_INVALID, "pm_scan_candidates: this handle runs the host stage only (pm_init_host)");
  if (begin < 0 || end < begin) return fail(h, PM_E_INVALID, "pm_scan_candidates: bad range");
  if (end > h->n) end = h->n;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  drain_spec(h);
  HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(unsigned long long), h->stream));
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  h->own_begin = begin; h->own_end = end;
  if (h->bases_edits) { int rcd = ensure_dp_tables(h); if (rcd) return rcd; }
  if (h->kern == PM_KERNEL_SEED && h->bases_edits) {                // a block seed comes from candidates up to maxlen + k away
    const int64_t reach = (int64_t)h->sd.maxlen + 2 * h->cfg.k + 4;
    begin = begin > reach ? begin - reach : 0;
    end = std::min<int64_t>(h->n, end + reach);
  }
  if (h->kern == PM_KERNEL_SEED && (h->edits_dev || (h->halves_dev && h->half_ranked_any)))
  {
    // per pattern tile: scan kernel -> seed records in d_ext, verify kernel -> candidates in d_cands
    // ~1 seed record per 8 bases at 200k patterns (key matches that pass the four-base-word test)
    // (sized for the range being scanned; a denser stream overflows once and the buffer grows)
    const size_t want_seeds = (size_t)((end - begin) / 6) + ((size_t)1 << 20);
    if (!h->d_seeds || h->seed_cap < want_seeds) {
      if (h->d_seeds) { (void)hipFree(h->d_seeds); h->d_seeds = nullptr; }
      h->seed_cap = std::max(h->seed_cap, want_seeds);
      HIP_TRY(h, hipMalloc((void **)&h->d_seeds, h->seed_cap * sizeof(uint64_t)));
    }
    if (!h->d_seed_count) HIP_TRY(h, hipMalloc((void **)&h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long)));
    const int ntiles = 1 + (int)h->sd_more.size();
    if (ntiles > 256) return fail(h, PM_E_UNSUPPORTED, "too many pattern tiles for the edit-distance plan");
    HIP_TRY(h, hipMemsetAsync(h->d_seed_count, 0, SEEDCOUNT_WORDS * sizeof(unsigned long long), h->stream));
    for (int t = 0; t < ntiles; ++t) {
      EditStage es; es.d_seeds = h->d_seeds; es.d_seed_count = h->d_seed_count + 1 + t; es.seed_cap = h->seed_cap; es.tile = t;
      if (h->bases_edits) {
        es.bases = true; es.b_codes = h->d_dp_codes; es.b_len = h->d_fpat_len; es.b_esb = h->d_dp_esb; es.b_eeb = h->d_dp_eeb;
        es.own_lo = h->own_begin; es.own_hi = h->own_end;
      }
      const SeedDevice &d = t == 0 ? h->sd : h->sd_more[t - 1];
      if (h->epair_on && h->edits_dev && t == 0) {
        // first stage on the pair geometry: windows whose frame can hold an end in (begin, end] are the positions begin - 3 .. end + 1
        const size_t want_susp = (size_t)((end - begin) / 10) + ((size_t)1 << 20);
        if (!h->d_susp || h->susp_cap < want_susp) {
          if (h->d_susp) { (void)hipFree(h->d_susp); h->d_susp = nullptr; }
          h->susp_cap = std::max(h->susp_cap, want_susp);
          HIP_TRY(h, hipMalloc(&h->d_susp, h->susp_cap * PAIR_SUSPECT_BYTES));
        }
        HIP_TRY(h, pair_launch(h->epair, h->d_text, h->d_packed, h->n, std::max<int64_t>(0, begin - 3), std::min<int64_t>(h->n, end + 2), h->d_cands, h->d_counter, h->cap,
                               h->d_susp, h->d_seed_count + 260, h->susp_cap, h->stream, &h->geo, nullptr, 3, h->d_seeds, h->d_seed_count + 1, h->seed_cap));
        es.skip_scan = true;
      }
      HIP_TRY(h, seed_launch(d, h->d_text, h->d_packed, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, t == 0 && !es.skip_scan ? &h->geo : nullptr, &es));
    }
    HIP_TRY(h, hipMemcpyAsync(h->h_seed_count, h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    h->last_launches = 2 * ntiles + (h->epair_on && h->edits_dev ? 1 : 0);
    if (h->nrest) { HIP_TRY(h, bitpar_launch(h->bp, h->d_text, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, nullptr)); ++h->last_launches; }
  }
  else if (h->kern == PM_KERNEL_SEED && !h->pair.empty())
  {
    // suspects (windows within k of a pattern on the packed bases, before the exact check): a few per
    // thousand positions on random streams; a denser stream overflows once and the buffer grows
    // (+ up to 15 unused slots per wave of the scan grid: slots are reserved 16 at a time)
    const size_t want = (size_t)((end - begin) / 128) + (size_t)((end - begin) / 8192) * h->pair[0].ncombos + ((size_t)1 << 20);
    if (!h->d_susp || h->susp_cap < want) {
      if (h->d_susp) { (void)hipFree(h->d_susp); h->d_susp = nullptr; }
      h->susp_cap = std::max(h->susp_cap, want);
      HIP_TRY(h, hipMalloc(&h->d_susp, h->susp_cap * PAIR_SUSPECT_BYTES));
    }
    if (!h->d_seed_count) HIP_TRY(h, hipMalloc((void **)&h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long)));
    if (h->pair.size() > 256) return fail(h, PM_E_UNSUPPORTED, "too many pattern tiles");
    HIP_TRY(h, hipMemsetAsync(h->d_seed_count, 0, SEEDCOUNT_WORDS * sizeof(unsigned long long), h->stream));
    for (size_t t = 0; t < h->pair.size(); ++t)
      HIP_TRY(h, pair_launch(h->pair[t], h->d_text, h->d_packed, h->n, begin, end, h->d_cands, h->d_counter, h->cap,
                             h->d_susp, h->d_seed_count + 1 + t, h->susp_cap, h->stream, t == 0 ? &h->geo : nullptr, h->d_seed_count + 257));
    HIP_TRY(h, hipMemcpyAsync(h->h_seed_count, h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    h->last_launches = 2 * (int)h->pair.size();
    if (h->nrest) { HIP_TRY(h, bitpar_launch(h->bp, h->d_text, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, nullptr)); ++h->last_launches; }
  }
  else if (h->kern == PM_KERNEL_SEED)
  {
    HIP_TRY(h, seed_launch(h->sd, h->d_text, h->d_packed, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, &h->geo));
    for (SeedDevice &d : h->sd_more)
      HIP_TRY(h, seed_launch(d, h->d_text, h->d_packed, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, nullptr));
    h->last_launches = 1 + (int)h->sd_more.size();
    if (h->nrest) { HIP_TRY(h, bitpar_launch(h->bp, h->d_text, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, nullptr)); ++h->last_launches; }
  }
  else
    HIP_TRY(h, bitpar_launch(h->bp, h->d_text, h->n, begin, end, h->d_cands, h->d_counter, h->cap, h->stream, &h->geo));
  h->scan_begin = begin; h->scan_end = end;
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  if (h->kern != PM_KERNEL_SEED) h->last_launches = 1;
  h->scan_pending = true;
  return PM_OK;
}


// sort workspace shared by pm_finalize_device (clustering) and the edit-distance dedup
static int ensure_sort_workspace(pm_handle *h, size_t n, bool with_out) {
  if (h->ckeys_cap < n || !h->d_keys) {
    void *fw[] = {h->d_keys, h->d_keys_alt, h->d_ctemp, h->d_fout, h->d_fleft};
    for (void *q : fw) if (q) (void)hipFree(q);
    h->d_keys = h->d_keys_alt = nullptr; h->d_ctemp = nullptr; h->d_fout = h->d_fleft = nullptr;
    h->ckeys_cap = std::max<size_t>(n + n / 4, (size_t)1 << 16);
    HIP_TRY(h, hipMalloc((void **)&h->d_keys, h->ckeys_cap * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_keys_alt, h->ckeys_cap * 8));
    h->ctemp_bytes = cluster_temp_bytes(h->ckeys_cap);
    HIP_TRY(h, hipMalloc(&h->d_ctemp, h->ctemp_bytes ? h->ctemp_bytes : 16));
  }
  if (with_out && !h->d_fout) {
    HIP_TRY(h, hipMalloc((void **)&h->d_fout, h->ckeys_cap * sizeof(pm_hit)));
    HIP_TRY(h, hipMalloc((void **)&h->d_fleft, h->ckeys_cap * sizeof(pm_hit)));
  }
  if (!h->d_fcounts) {
    HIP_TRY(h, hipMalloc((void **)&h->d_fcounts, 4 * sizeof(unsigned long long)));
    HIP_TRY(h, hipHostMalloc((void **)&h->h_fcounts, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  return PM_OK;
}

// Edit-distance seed plan: windows that do not fit in front of the stream start are not seeded, so
// every candidate that ends in the first Lw+2k+2 characters is produced here by running the
// automaton itself (shift_and_inexact.cc:249-352, rows start with l prefix bits :162-164) for each
// pattern over those few characters; the kernel's records for the same ends are duplicates and
// leave with the dedup.  Whatever range holds such an end gets them -- a caller's first range may be shorter than that
// (found by scripts/fuzz_families.py --dense-bound, seed 605103, at the other end of the stream).
static int edits_start_candidates(pm_handle *h, std::vector<pm_hit> *out) {
  const int k = h->cfg.k;
  const int64_t Tfull = std::min<int64_t>(h->n, h->sd.Lw + 2 * k + 2);
  const int64_t T = Tfull;
  if (T <= 0 || h->scan_begin >= Tfull) return PM_OK;
  if (h->start_cached) {                                            // same stream, same patterns: computed once
    for (const pm_hit &x : h->start_cache) if (x.end > h->scan_begin && x.end <= h->scan_end) out->push_back(x);
    return PM_OK;
  }
  std::vector<pm_hit> all, *extra = &all;
  uint8_t head[64] = {0};
  if (h->h_text) memcpy(head, h->h_text, (size_t)T);
  else HIP_TRY(h, hipMemcpy(head, h->d_text, (size_t)T, hipMemcpyDeviceToHost));
  for (size_t j = 0; j < h->inner.size(); ++j) {
    if (j < h->in_rest.size() && h->in_rest[j]) continue;           // the residue engine reports its own
    const std::string &s = h->inner[j].s;
    const int L = (int)s.size();
    uint64_t R[3] = {0, 1, 3};
    const uint64_t last = 1ull << (L - 1);
    uint64_t M[4] = {0, 0, 0, 0};                       // positions of A, C, G, T (the only pattern characters of the seed family)
    int code[4];
    for (int q = 0; q < 4; ++q) code[q] = h->alpha.nch[(unsigned char)"ACGT"[q]];
    for (int i = 0; i < L; ++i) for (int q = 0; q < 4; ++q)
      if (h->wild_seed ? acgt_of((unsigned char)s[i]).find("ACGT"[q]) != std::string::npos : s[i] == "ACGT"[q]) M[q] |= 1ull << i;
    for (int64_t t = 0; t < T; ++t) {
      const int c = head[t];
      if (c == h->eos_code) { R[0] = R[1] = R[2] = 0; continue; }
      const uint64_t U = c == code[0] ? M[0] : c == code[1] ? M[1] : c == code[2] ? M[2] : c == code[3] ? M[3] : 0;
      const uint64_t x0 = (R[0] << 1) | 1, m1 = x0 | R[0], n0 = x0 & U;
      const uint64_t x1 = (R[1] << 1) | 1, n1 = (x1 & U) | m1 | (n0 << 1) | 1 | n0, m2 = x1 | R[1];
      const uint64_t x2 = (R[2] << 1) | 1, n2 = (x2 & U) | m2 | (n1 << 1) | 1 | n1;
      R[0] = n0; R[1] = n1; R[2] = n2;
      const int lvl = (R[0] & last) ? 0 : (R[1] & last) ? 1 : (k >= 2 && (R[2] & last)) ? 2 : -1;
      if (lvl >= 0) {
        pm_hit x; x.end = t + 1; x.pid = h->inner_ids[j]; x.k = (uint8_t)lvl; x.aux[0] = x.aux[1] = x.aux[2] = 0;
        extra->push_back(x);
      }
    }
  }
  h->start_cache = all; h->start_cached = true;
  for (const pm_hit &x : all) if (x.end > h->scan_begin && x.end <= h->scan_end) out->push_back(x);
  return PM_OK;
}

// Edit-distance seed plan, the other end: a match whose clean pieces are followed by deleted pattern characters is
// seeded by the window one or two positions BEHIND its end (the seed's place is where the pattern's last base
// would be), and behind the last character of the stream there are no windows.  The candidates that end in the
// last four characters are therefore produced here as well: the automaton from the empty state over the last
// L + k + 8 characters -- its last bit depends on the last L + k only -- (from the stream-start state when that
// is the whole stream); duplicates of the kernel's records leave with the dedup.  Computed once per stream.
// Found by scripts/fuzz_families.py (seed 1308).
static int edits_end_candidates(pm_handle *h, std::vector<pm_hit> *extra) {
  const int k = h->cfg.k;
  const int64_t n = h->n;
  if (h->scan_end <= n - 4 || n <= 0) return PM_OK;                 // (the range that holds such an end gets it: the last range may be shorter than four positions)
  if (h->end_cached) {
    for (const pm_hit &x : h->end_cache) if (x.end > h->scan_begin && x.end <= h->scan_end) extra->push_back(x);
    return PM_OK;
  }
  const int64_t T = std::min<int64_t>(n, 32 + k + 8);
  uint8_t tail[64] = {0};
  if (h->h_text) memcpy(tail, h->h_text + (n - T), (size_t)T);
  else HIP_TRY(h, hipMemcpy(tail, h->d_text + (n - T), (size_t)T, hipMemcpyDeviceToHost));
  std::vector<pm_hit> all;
  for (size_t j = 0; j < h->inner.size(); ++j) {
    if (j < h->in_rest.size() && h->in_rest[j]) continue;           // the residue engine reports its own
    const std::string &s = h->inner[j].s;
    const int L = (int)s.size();
    const int64_t Tj = std::min<int64_t>(T, L + k + 8);
    uint64_t R[3] = {0, 0, 0};
    if (Tj == n) { R[1] = 1; R[2] = 3; }                            // the whole stream: rows start with l prefix bits
    const uint64_t last = 1ull << (L - 1);
    uint64_t M[4] = {0, 0, 0, 0};
    int code[4];
    for (int q = 0; q < 4; ++q) code[q] = h->alpha.nch[(unsigned char)"ACGT"[q]];
    for (int i = 0; i < L; ++i) for (int q = 0; q < 4; ++q)
      if (h->wild_seed ? acgt_of((unsigned char)s[i]).find("ACGT"[q]) != std::string::npos : s[i] == "ACGT"[q]) M[q] |= 1ull << i;
    for (int64_t t = T - Tj; t < T; ++t) {
      const int c = tail[t];
      if (c == h->eos_code) { R[0] = R[1] = R[2] = 0; continue; }
      const uint64_t U = c == code[0] ? M[0] : c == code[1] ? M[1] : c == code[2] ? M[2] : c == code[3] ? M[3] : 0;
      const uint64_t x0 = (R[0] << 1) | 1, m1 = x0 | R[0], n0 = x0 & U;
      const uint64_t x1 = (R[1] << 1) | 1, n1 = (x1 & U) | m1 | (n0 << 1) | 1 | n0, m2 = x1 | R[1];
      const uint64_t x2 = (R[2] << 1) | 1, n2 = (x2 & U) | m2 | (n1 << 1) | 1 | n1;
      R[0] = n0; R[1] = n1; R[2] = n2;
      const int lvl = (R[0] & last) ? 0 : (R[1] & last) ? 1 : (k >= 2 && (R[2] & last)) ? 2 : -1;
      const int64_t end = n - T + t + 1;
      if (lvl >= 0 && end > n - 4) {
        pm_hit x; x.end = end; x.pid = h->inner_ids[j]; x.k = (uint8_t)lvl; x.aux[0] = x.aux[1] = x.aux[2] = 0;
        all.push_back(x);
      }
    }
  }
  h->end_cache = all; h->end_cached = true;
  for (const pm_hit &x : all) if (x.end > h->scan_begin && x.end <= h->scan_end) extra->push_back(x);
  return PM_OK;
}

// The k-error automaton starts with the first l bits of every pattern set in row l
// (shift_and_inexact.cc:162-164), so at the very start of the stream a pattern whose first
// d <= k characters are "missing" is reported at end = L-d with level d + mismatches.  The seed
// kernel only sees whole windows; these few records are produced here and appended in HBM.
static int stream_start_candidates(pm_handle *h) {
  const int k = h->seed_k;
  if (!h->head_cached) {                                            // same stream, same patterns: computed once (0.3 ms per scan at 200k patterns)
  const int64_t need = std::min<int64_t>(h->n, 32);
  uint8_t head[32] = {0};
  if (need > 0) {
    if (h->h_text) memcpy(head, h->h_text, (size_t)need);
    else HIP_TRY(h, hipMemcpy(head, h->d_text, (size_t)need, hipMemcpyDeviceToHost));
  }
  std::vector<pm_hit> &extra = h->head_cache;
  extra.clear();
  for (size_t j = 0; j < h->inner.size(); ++j) {
    if (j < h->in_rest.size() && h->in_rest[j]) continue;           // the residue engine reports its own
    const std::string &s = h->inner[j].s;
    const int L = (int)s.size();
    for (int d = 1; d <= k && d < L; ++d) {
      const int e = L - d;
      if (e > h->n) continue;
      int lvl = d;
      bool dead = false;
      for (int i = 0; i < e && !dead; ++i) {
        if ((int)head[i] == h->eos_code) dead = true;                // EOS clears every row
        else if (h->wild_seed) { if (acgt_of((unsigned char)s[d + i]).find((char)h->alpha.ch[head[i]]) == std::string::npos) ++lvl; }
        else if ((int)head[i] != h->alpha.nch[(unsigned char)s[d + i]]) ++lvl;
      }
      if (!dead && lvl <= k) {
        pm_hit x; x.end = e; x.pid = h->inner_ids[j]; x.k = (uint8_t)lvl; x.aux[0] = x.aux[1] = x.aux[2] = 0;
        extra.push_back(x);
      }
    }
  }
  h->head_cached = true;
  }
  std::vector<pm_hit> extra;                                        // (the range that holds the end gets the record: a first range may be shorter than a pattern)
  for (const pm_hit &x : h->head_cache) if (x.end > h->own_begin && x.end <= h->own_end) extra.push_back(x);
  if (extra.empty()) return PM_OK;
  if (h->last_count + extra.size() > h->cap) {
    h->overflow_need = h->last_count + extra.size();
    h->last_count = 0;
    return fail(h, PM_E_OVERFLOW, "candidate buffer too small (pm_set_capacity)");
  }
  HIP_TRY(h, hipMemcpy(h->d_cands + h->last_count, extra.data(), extra.size() * sizeof(pm_hit), hipMemcpyHostToDevice));
  h->last_count += extra.size();
  return PM_OK;
}

// exact_halves reads past the end of the stream: a left half found as an exact seed right at the end is
// extended by primer_alignment_lmatch over len2 + k characters from the seed on (primer_alignment.cc:573-574)
// whether or not the stream still has them, and what char_io hands out there is the zero padding of the
// mapped file's last page (mapFile.h:49-57) -- code 0, the table's first character on a normalized stream.
// So a pattern hangs over the end by t <= len2 characters when its first L - t characters are the stream's
// last ones (left half exact, the rest within k substitutions, none in an exact zone) and its last t
// characters against code 0 keep the total within k: hit at end n + t.  The whole-pattern windows of the
// seed kernels end inside the stream; these few records (flag 1: found through the left seed) are made here
// and appended in HBM, like the records of the stream start.  With indels the half seeds are extended by
// pm_seed_extend, which reads code 0 past the end itself.  exact_bases -K is the same with the mandated first
// block in the left half's place (its records are final: the primer's id, no flag).
static int stream_end_overhang_candidates(pm_handle *h, bool bases) {
  const int k = h->cfg.k;
  const int64_t n = h->n;
  if (!h->overhang_cached) {                                        // same stream, same patterns: computed once
    std::vector<pm_hit> &all = h->overhang_cache;
    all.clear();
    const int64_t need = std::min<int64_t>(n, 32);
    uint8_t tail[32] = {0};                                         // tail[32 - need .. 32) = the last `need` characters
    if (need > 0) {
      if (h->h_text) memcpy(tail + 32 - need, h->h_text + (n - need), (size_t)need);
      else HIP_TRY(h, hipMemcpy(tail + 32 - need, h->d_text + (n - need), (size_t)need, hipMemcpyDeviceToHost));
    }
    auto differs = [&](unsigned char pc, int code) -> bool {
      if (h->cfg.wildcards) return acgt_of(pc).find((char)h->alpha.ch[code]) == std::string::npos;
      return code != h->alpha.nch[pc];
    };
    if (h->eos_code != 0) {                                         // (code 0 = EOS: the extension's first character past the end is a violation)
      for (size_t j = 0; j < h->pats.size(); ++j) {
        const Pattern &p = h->pats[j];
        const int L = (int)p.s.size();
        const int es = std::max(0, std::min(L, p.esb)), ee = std::max(0, std::min(L, p.eeb));
        // the part found as an exact seed inside the stream: the left half (exact_halves), the mandated first block
        // (exact_bases with esb >= eeb, exact_bases.cc:139-150; a last block lies on the stream with all of its pattern)
        if (bases && (es < ee || es == 0)) continue;
        const int len1 = bases ? es : L / 2, len2 = L - len1;
        for (int t = 1; t <= len2; ++t) {
          const int inside = L - t;                                 // pattern characters 0 .. inside-1 lie on the stream's last ones
          if (inside > need) continue;                              // (more of the pattern than the stream holds: a larger overhang may still fit)
          int lvl = 0;
          bool dead = false;
          for (int i = 0; i < L && !dead; ++i) {
            const int code = i < inside ? (int)tail[32 - inside + i] : 0;
            if (i < inside && code == h->eos_code) dead = true;
            else if (differs((unsigned char)p.s[i], code)) {
              if (i < len1 || i < es || i >= L - ee) dead = true;   // the seed part is exact; a substitution in an exact zone is a violation
              else ++lvl;
            }
          }
          if (!dead && lvl <= k) {
            pm_hit x; x.end = n + t; x.pid = bases ? (uint32_t)p.id : (uint32_t)(j + 1); x.k = (uint8_t)lvl; x.aux[0] = bases ? 0 : 1; x.aux[1] = x.aux[2] = 0;
            all.push_back(x);
          }
        }
      }
    }
    h->overhang_cached = true;
  }
  const std::vector<pm_hit> &extra = h->overhang_cache;
  if (extra.empty()) return PM_OK;
  if (h->last_count + extra.size() > h->cap) {
    h->overflow_need = h->last_count + extra.size();
    h->last_count = 0;
    return fail(h, PM_E_OVERFLOW, "candidate buffer too small (pm_set_capacity)");
  }
  HIP_TRY(h, hipMemcpy(h->d_cands + h->last_count, extra.data(), extra.size() * sizeof(pm_hit), hipMemcpyHostToDevice));
  h->last_count += extra.size();
  return PM_OK;
}

// per pattern (index = position in the pattern list): 32 stream codes, length, exact zones -- what the
// device-side DPs (pm_cluster_dp) and pm_bases_seeds read
static int ensure_dp_tables(pm_handle *h) {
  const size_t np = h->pats.size();
  if (!h->d_dp_codes) {
    std::vector<uint8_t> codes(np * 32, 0);
    std::vector<int32_t> es(np), ee(np);
    for (size_t i = 0; i < np; ++i) {
      for (size_t q = 0; q < h->pats[i].s.size() && q < 32; ++q) codes[i * 32 + q] = (uint8_t)h->alpha.nch[(unsigned char)h->pats[i].s[q]];
      es[i] = h->pats[i].esb; ee[i] = h->pats[i].eeb;
    }
    HIP_TRY(h, hipMalloc((void **)&h->d_dp_codes, codes.size() ? codes.size() : 32));
    HIP_TRY(h, hipMalloc((void **)&h->d_dp_esb, np ? np * 4 : 4));
    HIP_TRY(h, hipMalloc((void **)&h->d_dp_eeb, np ? np * 4 : 4));
    if (np) {
      HIP_TRY(h, hipMemcpy(h->d_dp_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_dp_esb, es.data(), np * 4, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_dp_eeb, ee.data(), np * 4, hipMemcpyHostToDevice));
    }
  }
  if (!h->d_fpat_len) {
    std::vector<uint8_t> pl(np); std::vector<uint32_t> pi(np);
    for (size_t i = 0; i < np; ++i) { pl[i] = (uint8_t)std::min<size_t>(h->pats[i].s.size(), 255); pi[i] = (uint32_t)h->pats[i].id; }
    HIP_TRY(h, hipMalloc((void **)&h->d_fpat_len, pl.size() ? pl.size() : 1));
    HIP_TRY(h, hipMalloc((void **)&h->d_fpat_id, pi.size() ? pi.size() * 4 : 4));
    if (!pl.empty()) {
      HIP_TRY(h, hipMemcpy(h->d_fpat_len, pl.data(), pl.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_fpat_id, pi.data(), pi.size() * 4, hipMemcpyHostToDevice));
    }
  }
  return PM_OK;
}

// An internal buffer between two kernels of one scan (seed records, suspects) was too small: it has
// been enlarged and the same range must be scanned again.  That is the library's own business --
// pm_scan_wait does it -- and never reaches the caller, whose record buffer (pm_set_capacity) was
// not the problem and must not be reallocated for it.
static const int SCAN_AGAIN = 1000;

// Records one list of a scan may hold before pm_scan cuts the range in two instead of growing it (candidates 16 B, suspects
// 16 B, seed records 8 B, and the finalize stage's sort workspace of ~70 B per candidate): 2^30 by default, PM_DENSE_BOUND
// for tests.  Always below the 2^31 items the device sorts count in an int.
static size_t dense_bound(const pm_handle *h) {
  const long long v = h->knobs.dense_bound;
  return v > 0 ? (size_t)std::min<long long>(v, (1ll << 31) - 1) : (size_t)1 << 30;
}
static int dense_fail(pm_handle *h, const char *what, unsigned long long n) {
  h->too_dense = true;
  h->last_count = 0;
  return fail(h, PM_E_UNSUPPORTED, std::string(what) + ": " + std::to_string(n) + " records in one range -- scan the stream in smaller ranges");
}

// Waiting for the handle's stream.  (Polling hipStreamQuery instead of hipStreamSynchronize measured no difference
// around the 15 ms scan kernels: 15.47 against 15.49 ms per step.)
static hipError_t stream_wait(pm_handle *h) { return hipStreamSynchronize(h->stream); }

static int scan_wait_once(pm_handle *h, size_t *n_out) {
  HIP_TRY(h, stream_wait(h));
  h->scan_pending = false;
  (void)hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1);
  const size_t cnt = (size_t)*h->h_counter;
  if (n_out) *n_out = cnt;
  h->last_peak = cnt;
  if (h->h_seed_count && h->kern == PM_KERNEL_SEED && (h->edits_dev || (h->halves_dev && h->half_ranked_any) || !h->pair.empty())) {
    const size_t tiles = !h->pair.empty() ? h->pair.size() : 1 + h->sd_more.size();
    for (size_t t = 0; t < tiles && t < 256; ++t) h->last_peak = std::max(h->last_peak, h->h_seed_count[1 + t]);
    h->last_peak = std::max(h->last_peak, h->h_seed_count[260]);
  }
  if (h->bound_on && cnt > dense_bound(h)) return dense_fail(h, "candidate records", cnt);
  if (cnt > h->cap) { h->last_count = 0; return fail(h, PM_E_OVERFLOW, "candidate buffer too small (pm_set_capacity)"); }
  h->last_count = cnt;
  if (h->edits_dev || (h->halves_dev && h->half_ranked_any)) {
    // the seed buffer of a tile must have held all its seed records
    unsigned long long worst = 0;
    for (int t = 0; t < 1 + (int)h->sd_more.size(); ++t) worst = std::max(worst, h->h_seed_count[1 + t]);
    if (h->knobs.debug) fprintf(stderr, "[pm] %s: %llu seed records (tile with most), seed cap %zu, candidates %zu\n", h->edits_dev ? "edits" : "halves", worst, h->seed_cap, cnt);
    if (h->bound_on && worst > dense_bound(h)) return dense_fail(h, "seed records of the edit-distance plan", worst);
    if (worst > h->seed_cap) {                                     // grow the seed buffer and tell the caller to scan again
      (void)hipFree(h->d_seeds); h->d_seeds = nullptr;
      h->seed_cap = (size_t)worst + (size_t)worst / 8 + 1024;
      h->last_count = 0;
      return SCAN_AGAIN;
    }
    if (h->bound_on && h->epair_on && h->edits_dev && h->h_seed_count[260] > dense_bound(h))
      return dense_fail(h, "suspects of the edit-distance plan", h->h_seed_count[260]);
    if (h->epair_on && h->edits_dev && h->h_seed_count[260] > h->susp_cap) {   // the pair geometry's suspect list between its two kernels
      (void)hipFree(h->d_susp); h->d_susp = nullptr;
      h->susp_cap = (size_t)h->h_seed_count[260] + (size_t)h->h_seed_count[260] / 8 + 1024;
      h->last_count = 0;
      return SCAN_AGAIN;
    }
  }
  if (h->halves_dev) {
    // second device pass: banded DP next to every surviving seed (pm_extend.hip); the records
    // handed on are the successful extensions
    if (!h->d_ext) HIP_TRY(h, hipMalloc((void **)&h->d_ext, h->cap * sizeof(pm_hit)));
    HIP_TRY(h, extend_seeds(h->d_text, h->n, h->d_cands, cnt, h->d_half_codes, h->d_half_len, h->d_hesb, h->d_heeb,
                            h->cfg.k, h->eos_code, h->d_ext, h->d_counter, h->cap, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, stream_wait(h));
    std::swap(h->d_cands, h->d_ext);
    h->last_count = (size_t)*h->h_counter;
    if (n_out) *n_out = h->last_count;
    h->last_launches += 1;
  }
  if (h->kern == PM_KERNEL_SEED && !h->pair.empty()) {
    // the suspect buffer must have held every tile's suspects
    unsigned long long worst = 0;
    for (size_t t = 0; t < h->pair.size(); ++t) worst = std::max(worst, h->h_seed_count[1 + t]);
    if (h->knobs.debug) fprintf(stderr, "[pm] pair plan: %llu suspects (tile with most), capacity %zu, candidates %zu\n", worst, h->susp_cap, cnt);
    if (h->bound_on && worst > dense_bound(h)) return dense_fail(h, "suspects of the pair plan", worst);
    if (worst > h->susp_cap) {                                     // grow it and tell the caller to scan again
      (void)hipFree(h->d_susp); h->d_susp = nullptr;
      h->susp_cap = (size_t)worst + (size_t)worst / 8 + 1024;
      h->last_count = 0;
      return SCAN_AGAIN;
    }
  }
  if (h->edits_dev) {
    // records of the stream start (host), then sort + unique on the device: several seeds report each candidate
    size_t tot = cnt;
    if (h->bases_edits && (h->own_begin < 56 || h->own_end > h->n - 56)) {
      // exact_bases -k: the records are occurrences of the mandated block, found through windows within k edits of
      // the whole pattern.  A pattern that hangs over the end of the stream has no such window (the extension DP reads
      // code 0 there, see stream_end_overhang_candidates), and at the start of the stream the windows that do not
      // fit are not seeded (a pattern whose first characters are deleted there ends in them).  Every block occurrence
      // in the first and last 56 characters therefore comes from the host as well -- more than needed: the
      // reference extends EVERY occurrence, the records are verified by the same DP, duplicates leave with the dedup.
      const int64_t n = h->n, E = std::min<int64_t>(n, 56);
      if (!h->edge_cached) {                                        // same stream, same patterns: computed once
        uint8_t edge[2][64] = {{0}, {0}};
        if (E > 0) {
          if (h->h_text) { memcpy(edge[0], h->h_text, (size_t)E); memcpy(edge[1], h->h_text + (n - E), (size_t)E); }
          else {
            HIP_TRY(h, hipMemcpy(edge[0], h->d_text, (size_t)E, hipMemcpyDeviceToHost));
            HIP_TRY(h, hipMemcpy(edge[1], h->d_text + (n - E), (size_t)E, hipMemcpyDeviceToHost));
          }
        }
        h->edge_cache.clear();
        for (size_t j = 0; j < h->pats.size(); ++j) {
          const Pattern &p = h->pats[j];
          const int L = (int)p.s.size();
          const int es = std::max(0, std::min(L, p.esb)), ee = std::max(0, std::min(L, p.eeb));
          const bool prefix = es >= ee;                              // exact_bases.cc:139-150: the larger block decides
          const int blk = prefix ? es : ee;
          if (blk <= 0) continue;
          const char *bs = p.s.data() + (prefix ? 0 : L - blk);
          for (int side = 0; side < 2; ++side) {
            const int64_t base = side == 0 ? 0 : n - E;               // stream index of edge[side][0]
            for (int64_t o = 0; o + blk <= E; ++o) {
              if (side == 1 && base + o + blk <= E) continue;         // (a short stream: already taken from its start)
              bool ok = true;
              for (int q = 0; q < blk && ok; ++q) ok = (int)edge[side][o + q] == h->alpha.nch[(unsigned char)bs[q]];
              if (ok) { pm_hit x; x.end = base + o + blk; x.pid = (uint32_t)(j + 1); x.k = 0; x.aux[0] = x.aux[1] = x.aux[2] = 0; h->edge_cache.push_back(x); }
            }
          }
        }
        h->edge_cached = true;
      }
      std::vector<pm_hit> extra;
      for (const pm_hit &x : h->edge_cache) {
        if (x.end > h->own_begin && x.end <= h->own_end) extra.push_back(x);     // (whatever range holds the end)
      }
      if (tot + extra.size() > h->cap) { h->last_count = 0; if (n_out) *n_out = tot + extra.size(); return fail(h, PM_E_OVERFLOW, "candidate buffer too small (pm_set_capacity)"); }
      if (!extra.empty()) HIP_TRY(h, hipMemcpy(h->d_cands + tot, extra.data(), extra.size() * sizeof(pm_hit), hipMemcpyHostToDevice));
      tot += extra.size();
    }
    if (!h->bases_edits) {                                          // (exact_bases: the records are block seeds, not automaton ends)
      std::vector<pm_hit> extra;
      int rc = edits_start_candidates(h, &extra);
      if (rc) return rc;
      rc = edits_end_candidates(h, &extra);
      if (rc) return rc;
      if (tot + extra.size() > h->cap) { h->last_count = 0; if (n_out) *n_out = tot + extra.size(); return fail(h, PM_E_OVERFLOW, "candidate buffer too small (pm_set_capacity)"); }
      if (!extra.empty()) HIP_TRY(h, hipMemcpy(h->d_cands + tot, extra.data(), extra.size() * sizeof(pm_hit), hipMemcpyHostToDevice));
      tot += extra.size();
    }
    if (tot >= ((size_t)1 << 31)) return dense_fail(h, "edit-distance plan, 2^31 or more candidate", tot);
    int rc = ensure_sort_workspace(h, tot, false);
    if (rc) return rc;
    const double td0 = now_ms();
    HIP_TRY(h, dedup_device(h->d_cands, tot, h->d_keys, h->d_keys_alt, h->d_ctemp, h->ctemp_bytes, h->d_cands, h->d_fcounts, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->h_fcounts, h->d_fcounts, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, stream_wait(h));
    h->last_count = (size_t)h->h_fcounts[0];
    if (h->knobs.debug) fprintf(stderr, "[pm] edits: %zu raw records (with holes) -> %zu unique candidates, dedup %.1f ms\n", tot, h->last_count, now_ms() - td0);
    if (n_out) *n_out = h->last_count;
    h->last_launches += 3;
    return PM_OK;
  }
  if (h->kern == PM_KERNEL_SEED && h->own_begin < 32 && h->seed_k > 0 &&
      (h->sem == PM_SEM_FILTER_BITVEC || h->sem == PM_SEM_SHIFT_AND_INEXACT)) {
    int rc = stream_start_candidates(h);
    if (rc) { if (rc == PM_E_OVERFLOW && n_out) *n_out = h->overflow_need; return rc; }
    if (n_out) *n_out = h->last_count;
  }
  if (h->kern == PM_KERNEL_SEED && (h->seed_flags || h->bases_flags) && h->own_end >= h->n) {
    int rc = stream_end_overhang_candidates(h, h->bases_flags);
    if (rc) { if (rc == PM_E_OVERFLOW && n_out) *n_out = h->overflow_need; return rc; }
    if (n_out) *n_out = h->last_count;
  }
  return PM_OK;
}

extern "C" int pm_scan_wait(pm_handle *h, size_t *n_out) {
  if (!h || !h->scan_pending) return fail(h, PM_E_INVALID, "pm_scan_wait: no scan in flight");
  float spent_ms = 0.f;                                             // kernel time of the attempts that had to be repeated
  for (int tries = 0;; ++tries) {
    const int rc = scan_wait_once(h, n_out);
    if (rc != SCAN_AGAIN) { h->last_ms += spent_ms; return rc; }
    spent_ms += h->last_ms;
    ++h->internal_rescans;
    if (tries >= 8) return fail(h, PM_E_NOMEM, "pm_scan_wait: the scan's internal record buffers keep overflowing");
    const int ra = pm_scan_candidates_async(h, h->own_begin, h->own_end);   // the caller's range, with the enlarged buffer
    if (ra) return ra;
  }
}

// The scan pm_scan enqueued for the range it expected next, and nobody asked for after all (another range, pm_reset,
// a direct scan or finalize call, pm_destroy): let it finish and forget it.
static void drain_spec(pm_handle *h) {
  if (!h->spec) return;
  h->spec = false;
  if (h->scan_pending) { (void)hipStreamSynchronize(h->stream); h->scan_pending = false; }
  h->last_count = 0;
}

extern "C" int pm_scan_candidates(pm_handle *h, int64_t begin, int64_t end, pm_hit *out, size_t cap, size_t *n_out) {
  int rc = pm_scan_candidates_async(h, begin, end);
  if (rc) return rc;
  size_t cnt = 0;
  rc = pm_scan_wait(h, &cnt);
  if (n_out) *n_out = cnt;
  if (rc) return rc;
  if (out) {
    if (cnt > cap) return fail(h, PM_E_OVERFLOW, "pm_scan_candidates: out buffer too small");
    if (cnt) HIP_TRY(h, hipMemcpy(out, h->d_cands, cnt * sizeof(pm_hit), hipMemcpyDeviceToHost));
  }
  return PM_OK;
}

extern "C" int pm_candidates_device(pm_handle *h, void **d_records, size_t *n) {
  if (!h || !h->inited) return PM_E_INVALID;
  if (d_records) *d_records = h->d_cands;
  if (n) *n = h->last_count;
  return PM_OK;
}

extern "C" int pm_scan_stats(pm_handle *h, uint64_t *out, int n) {
  if (!h || !h->inited || !out || n < 0) return PM_E_INVALID;
  uint64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  v[0] = h->last_count;
  if (h->h_seed_count) {
    const size_t tiles = !h->pair.empty() ? h->pair.size() : 1 + h->sd_more.size();
    for (size_t t = 0; t < tiles && t < 256; ++t) v[1] = std::max<uint64_t>(v[1], h->h_seed_count[1 + t]);
    v[3] = h->h_seed_count[257]; v[4] = h->h_seed_count[258]; v[5] = h->h_seed_count[259];
  }
  v[2] = h->internal_rescans;
  v[6] = h->range_splits;
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return PM_OK;
}

// Measurement (no reference counterpart): time the pair geometry as the FIRST STAGE of an edit-distance
// plan on this handle's tables -- a -K 2 handle on the pair plan -- over the whole stream: 14 (field pair, displacement)
// tests per window (scripts/edit_pair_cover.py), mode 1 = with the substitution compare (a lower bound of such a kernel),
// mode 2 = with the five-shift necessary condition for "<= 2 edits on the other ten bases" on two patterns per slot.
// Produces no hits: *ms is the kernel's duration, *suspects the records it would hand to a verify kernel.
extern "C" int pm_measure_pair_edit_floor(pm_handle *h, int mode, float *ms, uint64_t *suspects) {
  if (!h || !h->inited || !ms || !suspects || (mode != 1 && mode != 2)) return fail(h, PM_E_INVALID, "pm_measure_pair_edit_floor: bad arguments");
  if (h->kern != PM_KERNEL_SEED || h->pair.size() != 1 || h->cfg.k != 2 || h->cfg.indels)
    return fail(h, PM_E_UNSUPPORTED, "pm_measure_pair_edit_floor: needs a -K 2 handle on the pair plan (one pattern tile)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  drain_spec(h);
  const size_t want = (size_t)(h->n / 16) + ((size_t)1 << 20);
  if (!h->d_susp || h->susp_cap < want) {
    if (h->d_susp) { (void)hipFree(h->d_susp); h->d_susp = nullptr; }
    h->susp_cap = want;
    HIP_TRY(h, hipMalloc(&h->d_susp, h->susp_cap * PAIR_SUSPECT_BYTES));
  }
  if (!h->d_seed_count) HIP_TRY(h, hipMalloc((void **)&h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMemsetAsync(h->d_seed_count, 0, SEEDCOUNT_WORDS * sizeof(unsigned long long), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(unsigned long long), h->stream));
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  HIP_TRY(h, pair_launch(h->pair[0], h->d_text, h->d_packed, h->n, 0, h->n, h->d_cands, h->d_counter, h->cap, h->d_susp, h->d_seed_count + 1, h->susp_cap,
                         h->stream, nullptr, nullptr, mode));
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->h_seed_count, h->d_seed_count, SEEDCOUNT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  (void)hipEventElapsedTime(ms, h->ev0, h->ev1);
  *suspects = h->h_seed_count[1];
  return PM_OK;
}

extern "C" int pm_pack_time(pm_handle *h, float *ms) {
  if (!h || !h->inited || !ms) return PM_E_INVALID;
  *ms = h->kern == PM_KERNEL_SEED ? h->pack_ms : 0.f;
  return PM_OK;
}

extern "C" int pm_final_hits_device(pm_handle *h, void **d_hits, size_t *n) {
  if (!h || !h->inited) return PM_E_INVALID;
  if (d_hits) *d_hits = const_cast<pm_hit *>(h->d_final);
  if (n) *n = h->n_final;
  return PM_OK;
}

extern "C" int pm_copy_records(pm_handle *h, const void *d_src, size_t n, pm_hit *out) {
  if (!h || !h->inited || (n && (!d_src || !out))) return fail(h, PM_E_INVALID, "pm_copy_records: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  if (n) HIP_TRY(h, hipMemcpyAsync(out, d_src, n * sizeof(pm_hit), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, stream_wait(h));
  return PM_OK;
}

extern "C" int pm_last_kernel_time(pm_handle *h, float *ms, int *launches) {
  if (!h) return PM_E_INVALID;
  if (ms) *ms = h->last_ms;
  if (launches) *launches = h->last_launches;
  return PM_OK;
}

// ---- host stage --------------------------------------------------------------------------------
namespace {

struct Window { int64_t start; int32_t len; int64_t off; };

// Raw characters (cp.getch()) of a batch of stream windows, from the host copy of the stream
// when there is one, else gathered from HBM by one kernel + one copy.  Bytes outside [0,n) read
// as code 0, like the reference's unchecked mmap reads of the zero-padded last page.
int fetch_windows(pm_handle *h, std::vector<Window> &wins) {
  int64_t total = 0;
  for (Window &w : wins) { w.off = total; total += w.len; }
  h->winbuf.resize((size_t)total + 1);
  if (wins.empty()) return PM_OK;
  if (h->h_text) {
    // host copy of the stream: copy + code -> character in one pass, slices of windows per thread
    auto copy = [&](size_t lo, size_t hi) {
      for (size_t wi = lo; wi < hi; ++wi) {
        const Window &w = wins[wi];
        for (int i = 0; i < w.len; ++i) {
          const int64_t p = w.start + i;
          h->winbuf[w.off + i] = h->alpha.ch[(p >= 0 && p < h->n) ? h->h_text[p] : 0];
        }
      }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = wins.size() < 4096 ? 1 : std::min<size_t>(hw ? hw : 1, 16);
    if (nthreads <= 1) copy(0, wins.size());
    else {
      std::vector<std::thread> pool;
      const size_t per = (wins.size() + nthreads - 1) / nthreads;
      for (size_t t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() { copy(std::min(wins.size(), t * per), std::min(wins.size(), (t + 1) * per)); });
      for (std::thread &th : pool) th.join();
    }
    return PM_OK;
  } else {
    const size_t cnt = wins.size();
    if (h->d_wcap < cnt) {
      if (h->d_wstart) { (void)hipFree(h->d_wstart); (void)hipFree(h->d_wlen); (void)hipFree(h->d_woff); }
      h->d_wcap = cnt * 2;
      HIP_TRY(h, hipMalloc((void **)&h->d_wstart, h->d_wcap * 8));
      HIP_TRY(h, hipMalloc((void **)&h->d_wlen, h->d_wcap * 4));
      HIP_TRY(h, hipMalloc((void **)&h->d_woff, h->d_wcap * 8));
    }
    if (h->d_woutcap < (size_t)total) {
      if (h->d_wout) (void)hipFree(h->d_wout);
      h->d_woutcap = (size_t)total * 2;
      HIP_TRY(h, hipMalloc((void **)&h->d_wout, h->d_woutcap));
    }
    std::vector<int64_t> st(cnt), of(cnt); std::vector<int32_t> ln(cnt);
    for (size_t i = 0; i < cnt; ++i) { st[i] = wins[i].start; ln[i] = wins[i].len; of[i] = wins[i].off; }
    // (while pm_scan has the next range's scan on the handle's stream, the windows go through the copy stream: the
    // stream text is read-only and the staging buffers are used by this function alone)
    hipStream_t ws = h->spec && h->copy_stream ? h->copy_stream : h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_wstart, st.data(), cnt * 8, hipMemcpyHostToDevice, ws));
    HIP_TRY(h, hipMemcpyAsync(h->d_wlen, ln.data(), cnt * 4, hipMemcpyHostToDevice, ws));
    HIP_TRY(h, hipMemcpyAsync(h->d_woff, of.data(), cnt * 8, hipMemcpyHostToDevice, ws));
    HIP_TRY(h, gather_windows(h->d_text, h->n, h->d_wstart, h->d_wlen, h->d_woff, (int)cnt, h->d_wout, ws));
    HIP_TRY(h, hipMemcpyAsync(h->winbuf.data(), h->d_wout, (size_t)total, hipMemcpyDeviceToHost, ws));
    HIP_TRY(h, hipStreamSynchronize(ws));
  }
  for (int64_t i = 0; i < total; ++i) h->winbuf[i] = h->alpha.ch[h->winbuf[i]];
  return PM_OK;
}

bool by_end_pid(const pm_hit &a, const pm_hit &b) {
  if (a.end != b.end) return a.end < b.end;
  if (a.pid != b.pid) return a.pid < b.pid;
  return a.k < b.k;
}

// (end, pid, k) order for large hit lists: ends are spread over the scanned range, so split the
// range of ends into equal slices, one thread per slice (std::sort of 10^6 records took 25 ms)
void sort_hits(pm_hit *v, size_t n) {
  if (n < ((size_t)1 << 16)) { std::sort(v, v + n, by_end_pid); return; }
  int64_t lo = v[0].end, hi = v[0].end;
  for (size_t i = 1; i < n; ++i) { lo = std::min(lo, v[i].end); hi = std::max(hi, v[i].end); }
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const int64_t span = (hi - lo) / T + 1;
  std::vector<size_t> cnt((size_t)T + 1, 0);
  for (size_t i = 0; i < n; ++i) ++cnt[(size_t)((v[i].end - lo) / span) + 1];
  for (int t = 0; t < T; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
  std::vector<pm_hit> tmp(n);
  std::vector<size_t> at(cnt.begin(), cnt.end() - 1);
  for (size_t i = 0; i < n; ++i) tmp[at[(size_t)((v[i].end - lo) / span)]++] = v[i];
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
    std::sort(tmp.begin() + (ptrdiff_t)cnt[(size_t)t], tmp.begin() + (ptrdiff_t)cnt[(size_t)t + 1], by_end_pid);
    std::copy(tmp.begin() + (ptrdiff_t)cnt[(size_t)t], tmp.begin() + (ptrdiff_t)cnt[(size_t)t + 1], v + cnt[(size_t)t]);
  });
  for (std::thread &x : th) x.join();
}

pm_hit make_hit(int64_t end, uint64_t pid, int k) {
  pm_hit x; x.end = end; x.pid = (uint32_t)pid; x.k = (uint8_t)k; x.aux[0] = x.aux[1] = x.aux[2] = 0;
  return x;
}

// filter_bitvec::find_patterns (filter_bitvec.cc:88-177) over carried + new candidates.
// The reference sorts all candidates by position and, for every still-live candidate, chains
// later candidates of the same pattern while they lie within 2k+1 of the last one chained
// (:103-116).  Chains never mix patterns, so the same clusters fall out of grouping the
// candidates by pattern (counting sort, O(n)) and cutting each pattern's ends at gaps > 2k+1.
int finalize_filter_bitvec(pm_handle *h, const pm_hit *cands, size_t n, int64_t scanned_to, bool last,
                           std::vector<pm_hit> &outv) {
  const double tf0 = now_ms();
  const size_t np = h->pats.size();
  const int k = h->cfg.k, win = 2 * k + 1;
  const bool indels = h->cfg.indels != 0;
  const size_t total = h->carry.size() + n;
  if (total == 0) return PM_OK;
  // group by inner pattern id (1..np)
  std::vector<uint32_t> first(np + 2, 0);
  for (const pm_hit &c : h->carry) ++first[c.pid + 1];
  for (size_t i = 0; i < n; ++i) ++first[cands[i].pid + 1];
  for (size_t j = 1; j <= np + 1; ++j) first[j] += first[j - 1];
  std::vector<pm_hit> g(total);
  {
    std::vector<uint32_t> at(first.begin(), first.end() - 1);
    for (const pm_hit &c : h->carry) g[at[c.pid]++] = c;
    for (size_t i = 0; i < n; ++i) g[at[cands[i].pid]++] = cands[i];
  }
  std::vector<pm_hit> keep;
  struct Cluster { int64_t first, last; uint32_t pid; };
  std::vector<Cluster> need_dp;
  for (size_t pid = 1; pid <= np; ++pid) {
    const size_t lo = first[pid], hi = first[pid + 1];
    if (lo == hi) continue;
    if (hi - lo > 1) std::sort(g.begin() + lo, g.begin() + hi, [](const pm_hit &a, const pm_hit &b) { return a.end < b.end; });
    const Pattern &p = h->pats[pid - 1];
    const int L = (int)p.s.size();
    size_t i = lo;
    while (i < hi) {
      size_t j = i + 1;
      int best_lvl = g[i].k; int64_t best_end = g[i].end;
      while (j < hi && g[j].end <= g[j - 1].end + win) {
        if (g[j].k < best_lvl) { best_lvl = g[j].k; best_end = g[j].end; }
        ++j;
      }
      const int64_t c_first = g[i].end, c_last = g[j - 1].end;
      if (!last && scanned_to < c_last + win) {                     // :118-121: the cluster may still grow
        keep.insert(keep.end(), g.begin() + i, g.begin() + j);
      } else if (!indels && p.esb == 0 && p.eeb == 0 && c_first >= L && !(h->nn_quirk && pattern_n_quirk(h, p))) {
        // Substitution-only search without exact-base constraints needs no text: the DP
        // (pattern_alignment.cc with b = 0) walks diagonals only, so the cluster's value is the
        // smallest Hamming distance among its windows -- the smallest candidate level, windows that
        // are not candidates having distance > k -- and the column rule (:461-475) keeps the
        // left-most such window.
        outv.push_back(make_hit(best_end, p.id, best_lvl));
      } else need_dp.push_back(Cluster{c_first, c_last, (uint32_t)pid});
      i = j;
    }
  }
  h->carry.swap(keep);
  if (need_dp.empty()) return PM_OK;
  std::vector<Window> wins;
  wins.reserve(need_dp.size());
  for (const Cluster &c : need_dp) {
    const int L = (int)h->pats[c.pid - 1].s.size();
    int64_t ws = 0;
    if (c.first > (int64_t)L + k) ws = c.first - L - k;             // pattern_alignment.cc:137-139
    wins.push_back(Window{ws, (int32_t)(c.last - ws), 0});
  }
  const double tf1 = now_ms();
  int rc = fetch_windows(h, wins);
  if (rc) return rc;
  const double tf2 = now_ms();
  AlignParams prm; prm.k = k; prm.indels = indels; prm.eos = (uint8_t)h->cfg.eos;
  prm.wc = h->cfg.wildcards != 0; prm.tn = h->cfg.text_n != 0;
  // one DP per cluster; the clusters are independent: slices per host thread, results kept in order
  std::vector<AlignResult> res(need_dp.size());
  auto run = [&](size_t lo, size_t hi, AlignScratch &scratch) {
    for (size_t wi = lo; wi < hi; ++wi) {
      const Cluster &c = need_dp[wi];
      const Pattern &p = h->pats[c.pid - 1];
      res[wi] = editdist_align(h->winbuf.data() + wins[wi].off, wins[wi].start, c.first, c.last,
                               p.s.data(), (int)p.s.size(), p.esb, p.eeb, prm, scratch);
    }
  };
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nthreads = need_dp.size() < 4096 ? 1 : std::min<size_t>(hw ? hw : 1, 16);
  if (nthreads <= 1) run(0, need_dp.size(), h->scratch);
  else {
    std::vector<std::thread> pool;
    const size_t per = (need_dp.size() + nthreads - 1) / nthreads;
    for (size_t t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t]() { AlignScratch sc; run(std::min(need_dp.size(), t * per), std::min(need_dp.size(), (t + 1) * per), sc); });
    for (std::thread &th : pool) th.join();
  }
  for (size_t wi = 0; wi < need_dp.size(); ++wi)
    if (res[wi].ok) outv.push_back(make_hit(res[wi].end, h->pats[need_dp[wi].pid - 1].id, res[wi].value));   // :135
  if (h->knobs.debug) fprintf(stderr, "[pm] filter_bitvec host stage: cluster %.1f ms, windows %.1f ms, %zu DPs %.1f ms\n", tf1 - tf0, tf2 - tf1, need_dp.size(), now_ms() - tf2);
  return PM_OK;
}

bool seed_order(const pm_hit &a, const pm_hit &b) {                  // exact_halves.cc:114-118
  if (a.end != b.end) return a.end < b.end;
  return a.pid > b.pid;
}

// Sort records into seed_order.  std::sort on 16-byte records was most of the host stage
// (5 ms per 80k records); positions < 2^40 and ids < 2^24 make one 64-bit key, sorted by an LSD
// radix sort of (key, index) pairs.
void sort_seed_order(std::vector<pm_hit> &v) {
  const size_t n = v.size();
  bool fits = n < ((size_t)1 << 31);
  for (size_t i = 0; i < n && fits; ++i) fits = v[i].end >= 0 && v[i].end < ((int64_t)1 << 40) && v[i].pid < (1u << 24);
  if (!fits || n < 64) { std::sort(v.begin(), v.end(), seed_order); return; }
  struct KI { uint64_t key; uint32_t idx; };
  std::vector<KI> a(n), b(n);
  uint64_t all_or = 0;
  for (size_t i = 0; i < n; ++i) {
    a[i].key = ((uint64_t)v[i].end << 24) | (uint64_t)(0xffffffu - v[i].pid);   // id descending
    a[i].idx = (uint32_t)i;
    all_or |= a[i].key;
  }
  constexpr int BITS = 11;
  std::vector<uint32_t> cnt((size_t)1 << BITS);
  for (int shift = 0; shift < 64 && (all_or >> shift) != 0; shift += BITS) {
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (size_t i = 0; i < n; ++i) ++cnt[(a[i].key >> shift) & ((1u << BITS) - 1)];
    uint32_t run = 0;
    for (uint32_t &c : cnt) { const uint32_t t = c; c = run; run += t; }
    for (size_t i = 0; i < n; ++i) b[cnt[(a[i].key >> shift) & ((1u << BITS) - 1)]++] = a[i];
    a.swap(b);
  }
  std::vector<pm_hit> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = v[a[i].idx];
  v.swap(out);
}

// exact_halves::find_patterns (exact_halves.cc:140-190) / exact_bases (exact_bases.cc:92-121)
int finalize_seeds(pm_handle *h, const pm_hit *cands, size_t n, bool halves, std::vector<pm_hit> &outv) {
  std::vector<pm_hit> seeds(cands, cands + n);
  if (halves) sort_seed_order(seeds);
  else std::sort(seeds.begin(), seeds.end(), by_end_pid);
  const int k = h->cfg.k;
  const bool indels = h->cfg.indels != 0;
  struct Job { int pat; bool left; int len1, len2; };
  std::vector<Job> jobs(seeds.size());
  std::vector<Window> wins(seeds.size());
  for (size_t i = 0; i < seeds.size(); ++i) {
    const uint32_t hid = seeds[i].pid;
    Job j;
    if (halves) {
      j.pat = (int)((hid - 1) / 2);
      const int L = (int)h->pats[j.pat].s.size();
      j.len1 = L / 2; j.len2 = L - j.len1; j.left = (hid % 2) == 1;
    } else {
      j.pat = (int)hid - 1;
      const Pattern &p = h->pats[j.pat];
      const int L = (int)p.s.size();
      j.left = p.esb >= p.eeb;
      j.len1 = j.left ? p.esb : L - p.eeb; j.len2 = L - j.len1;
    }
    jobs[i] = j;
    if (j.left) wins[i] = Window{seeds[i].end, j.len2 + k, 0};                       // primer_alignment.cc:573-574
    else {
      int64_t ws = 0;
      const int plen = j.len1 + j.len2 + k;
      if (seeds[i].end > (int64_t)plen) ws = seeds[i].end - plen;                      // :657-662
      wins[i] = Window{ws, (int32_t)std::max<int64_t>(0, seeds[i].end - j.len2 - ws), 0};
    }
  }
  int rc = fetch_windows(h, wins);
  if (rc) return rc;
  AlignParams prm; prm.k = k; prm.indels = indels; prm.eos = (uint8_t)h->cfg.eos;
  prm.wc = h->cfg.wildcards != 0; prm.tn = h->cfg.text_n != 0;
  for (size_t i = 0; i < seeds.size(); ++i) {
    const Job &j = jobs[i];
    const Pattern &p = h->pats[j.pat];
    int64_t end = 0; int val = 0; bool ok;
    if (j.left) ok = lmatch_extend(h->winbuf.data() + wins[i].off, seeds[i].end, j.len1, p.s.data() + j.len1, j.len2,
                                   p.esb, p.eeb, prm, h->scratch, &end, &val);
    else ok = rmatch_extend(h->winbuf.data() + wins[i].off, wins[i].len, seeds[i].end, p.s.data(), j.len1, j.len2,
                            p.esb, p.eeb, prm, h->scratch, &end, &val);
    if (!ok) continue;
    if (halves) {
      if (end > h->lasthit[j.pat + 1] + (indels ? 2 * k : 0)) {      // exact_halves.cc:163,178
        outv.push_back(make_hit(end, p.id, val));
        h->lasthit[j.pat + 1] = end;
      }
    } else outv.push_back(make_hit(end, p.id, val));
  }
  return PM_OK;
}

// exact_halves on whole-pattern Hamming candidates (seed family, -K only): a candidate whose left
// half is clean was found by the reference through the left seed at end-len2 (inner id 2j-1), one
// whose right half is clean through the right seed at end (inner id 2j); both report
// (end, distance) (primer_alignment.cc:568-617, 651-704 with indels off).  Seeds are then
// replayed in the reference's order (position asc, inner id desc) through the per-pattern
// "end > last kept end" rule (exact_halves.cc:114-118,163,178).
int finalize_halves_flags(pm_handle *h, const pm_hit *cands, size_t n, int64_t scanned_to, bool last,
                          std::vector<pm_hit> &outv) {
  // a seed is kept in h->carry as {end = seed position, pid = inner id, k = value, aux[0] = end - position}
  std::vector<pm_hit> &seeds = h->carry;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t j = cands[i].pid;                                // 1-based pattern index
    const int L = (int)h->pats[j - 1].s.size();
    const int len2 = L - L / 2;
    if (cands[i].aux[0] & 1) { pm_hit x = make_hit(cands[i].end - len2, 2 * j - 1, cands[i].k); x.aux[0] = (uint8_t)len2; seeds.push_back(x); }
    if (cands[i].aux[0] & 2) seeds.push_back(make_hit(cands[i].end, 2 * j, cands[i].k));
  }
  sort_seed_order(seeds);
  // candidates still to come end beyond scanned_to, so their seeds lie beyond scanned_to - maxlen:
  // everything at or before that is in its final order
  const int64_t safe = last ? INT64_MAX : scanned_to - h->sd.maxlen;
  size_t i = 0;
  for (; i < seeds.size() && seeds[i].end <= safe; ++i) {
    const uint32_t j = (seeds[i].pid + 1) / 2;
    const int64_t end = seeds[i].end + seeds[i].aux[0];
    if (end > h->lasthit[j]) {
      outv.push_back(make_hit(end, h->pats[j - 1].id, seeds[i].k));
      h->lasthit[j] = end;
    }
  }
  seeds.erase(seeds.begin(), seeds.begin() + i);
  return PM_OK;
}

// exact_halves -k with the extension done on the GPU (pm_extend.hip): records are
// {end = seed position, pid = inner id, k = value, aux[0] = hit end - seed position}.  What is left
// is the reference's sequential rule (exact_halves.cc:142,163,178): seeds in (position asc, inner id
// desc) order, a hit is kept if its end exceeds the pattern's last kept end by more than 2k.
int finalize_extended(pm_handle *h, const pm_hit *cands, size_t n, std::vector<pm_hit> &outv) {
  std::vector<pm_hit> seeds(cands, cands + n);
  sort_seed_order(seeds);
  const int slack = h->cfg.indels ? 2 * h->cfg.k : 0;
  for (const pm_hit &s : seeds) {
    const uint32_t j = (s.pid + 1) / 2;
    const int64_t end = s.end + s.aux[0];
    if (end > h->lasthit[j] + slack) {
      outv.push_back(make_hit(end, h->pats[j - 1].id, s.k));
      h->lasthit[j] = end;
    }
  }
  return PM_OK;
}

}  // namespace

static int finalize_into(pm_handle *h, const pm_hit *cands, size_t n, int64_t scanned_to, bool last,
                         std::vector<pm_hit> &outv) {
  int rc = PM_OK;
  switch (h->sem) {
    case PM_SEM_KEYWORD_TREE: case PM_SEM_SHIFT_AND: case PM_SEM_SHIFT_AND_INEXACT:
      outv.insert(outv.end(), cands, cands + n);
      for (size_t i = outv.size() - n; i < outv.size(); ++i) outv[i].aux[0] = outv[i].aux[1] = outv[i].aux[2] = 0;   // engine-private bytes
      break;
    case PM_SEM_FILTER_BITVEC: rc = finalize_filter_bitvec(h, cands, n, scanned_to, last, outv); break;
    case PM_SEM_EXACT_HALVES:
      h->halves_fresh = false;
      rc = h->seed_flags ? finalize_halves_flags(h, cands, n, scanned_to, last, outv)
         : h->halves_dev ? finalize_extended(h, cands, n, outv) : finalize_seeds(h, cands, n, true, outv);
      break;
    case PM_SEM_EXACT_BASES:
      if (h->bases_flags) {                                          // seed family: the records are the hits
        outv.insert(outv.end(), cands, cands + n);
        for (size_t i = outv.size() - n; i < outv.size(); ++i) outv[i].aux[0] = outv[i].aux[1] = outv[i].aux[2] = 0;
      }
      else rc = finalize_seeds(h, cands, n, false, outv);
      break;
    default: return fail(h, PM_E_INVALID, "finalize: bad semantics");
  }
  return rc;
}

extern "C" int pm_finalize(pm_handle *h, const pm_hit *cands, size_t n, int64_t scanned_to, int flags,
                           pm_hit *out, size_t cap, size_t *n_out) {
  const int last = flags & PM_FINALIZE_LAST;
  if (!h || !h->inited || (!cands && n)) return fail(h, PM_E_INVALID, "pm_finalize: bad arguments");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  std::vector<pm_hit> outv;
  int rc = finalize_into(h, cands, n, scanned_to, last != 0, outv);
  if (rc) return rc;
  if (flags & PM_FINALIZE_SORTED) sort_hits(outv.data(), outv.size());
  if (n_out) *n_out = outv.size();
  if (outv.size() > cap) return fail(h, PM_E_OVERFLOW, "pm_finalize: out buffer too small");
  if (!outv.empty()) memcpy(out, outv.data(), outv.size() * sizeof(pm_hit));
  return PM_OK;
}

// Device form of pm_finalize for the option sets whose host stage needs no stream text and no
// order-dependent state: exact engines and bare shift_and_inexact (pass-through) and
// filter_bitvec with -K and no exact-base constraints (sort + segmented pass, pm_cluster.hip).
// d_cands == NULL means "the records of the last pm_scan_candidates".  Final hits are copied to
// `out` (host).  The few clusters the device cannot decide (still growing at scanned_to, or
// starting inside the first L characters) go through the host stage.
// `next` != NULL is pm_scan's form: the final hits are sorted on the device and land, in (end, pid, k) order, behind the
// records the handle's landing buffer already holds; with next->on the scan of the range (next->b, next->e] is enqueued
// behind the finalize kernels before the host waits for them, so that the copies out of HBM and everything the caller
// does with the hits run beside it.
struct ScanNext { bool on; int64_t b, e; };
static int finalize_device_impl(pm_handle *h, const void *d_cands, size_t n, int64_t scanned_to, int flags,
                                const OwnedRange &own, pm_hit *out, size_t cap, size_t *n_out, const ScanNext *next = nullptr);

// filter_bitvec with -K and no exact-base constraints: its verify is "smallest level, left-most end"
// (pm_cluster.hip), no stream text needed
static bool device_cluster_plain(const pm_handle *h) {
  if (!(h->sem == PM_SEM_FILTER_BITVEC && !h->cfg.indels && h->cfg.k <= 3 && h->pats.size() < ((size_t)1 << 22))) return false;
  if (h->nn_quirk) return false;                                    // pattern N at a stream N under -w: the value is the DP's (pattern_n_quirk)
  // with exact-base constraints only on the pair plan, whose records mark zone violations (level 3, k <= 2)
  if (h->zoned) return h->kern == PM_KERNEL_SEED && !h->pair.empty() && h->nrest == 0 && h->cfg.k <= 2;
  return true;
}

extern "C" int pm_finalize_device(pm_handle *h, const void *d_cands, size_t n, int64_t scanned_to, int flags,
                                  pm_hit *out, size_t cap, size_t *n_out) {
  const OwnedRange all = {0, 0, 0, 0, 0};
  if (h) drain_spec(h);
  return finalize_device_impl(h, d_cands, n, scanned_to, flags, all, out, cap, n_out);
}

// One shard of a position-sharded scan: the records hold every candidate that ends
// in (guard_lo, guard_hi] and this call reports the hits that end in (own_lo, own_hi], so the shards'
// outputs concatenate to the single-scan result without a merge stage and each shard's cluster DPs
// read the text the shard itself holds.  guard_lo <= 0 / guard_hi == INT64_MAX: the stream really
// starts / ends there.  A same-pattern chain of candidates that reaches from a guard edge into the
// owned range (a tandem repeat longer than the guard band) cannot be decided here: PM_E_UNSUPPORTED.
extern "C" int pm_finalize_device_owned(pm_handle *h, const void *d_cands, size_t n, int64_t own_lo, int64_t own_hi,
                                        int64_t guard_lo, int64_t guard_hi, int flags, pm_hit *out, size_t cap, size_t *n_out) {
  if (!(guard_lo <= own_lo && own_lo <= own_hi && own_hi <= guard_hi))
    return fail(h, PM_E_INVALID, "pm_finalize_device_owned: need guard_lo <= own_lo <= own_hi <= guard_hi");
  // the shard that holds the true end of the stream also owns the hits that end beyond it (extensions that read
  // past the end, stream_end_overhang_candidates)
  if (h) drain_spec(h);
  const bool to_the_end = h && guard_hi == INT64_MAX && own_hi >= h->n;
  const OwnedRange own = {own_lo, to_the_end ? INT64_MAX : own_hi, guard_lo, guard_hi, 1};
  return finalize_device_impl(h, d_cands, n, guard_hi == INT64_MAX ? h->n : guard_hi, flags | PM_FINALIZE_LAST, own, out, cap, n_out);
}

// room for `need` records in the landing buffer, the ones not handed out yet moved to its front
static int ensure_landing(pm_handle *h, size_t need_more) {
  const size_t live = h->land_n - h->land_pos;
  if (h->land_cap >= live + need_more) {
    if (h->land_pos) { if (live) memmove(h->land, h->land + h->land_pos, live * sizeof(pm_hit)); h->land_n = live; h->land_pos = 0; }
    return PM_OK;
  }
  const size_t want = live + need_more;
  const size_t cap = std::max<size_t>(want + want / 2, (size_t)1 << 16);
  pm_hit *p = nullptr;
  HIP_TRY(h, hipHostMalloc((void **)&p, cap * sizeof(pm_hit), hipHostMallocDefault));
  if (live) memcpy(p, h->land + h->land_pos, live * sizeof(pm_hit));
  if (h->land) (void)hipHostFree(h->land);
  h->land = p; h->land_cap = cap; h->land_n = live; h->land_pos = 0;
  return PM_OK;
}

// Can the device put final hits in (end, pid, k) order with one keys-only radix sort (pm_cluster.hip sort_final_device)?
// key = end | pattern index | k in <= 63 bits, pattern ids growing with the index.  Decided once per init.
static void device_sort_plan(pm_handle *h) {
  h->sort_dev = false;
  if (h->cfg.k > 3 || h->pats.empty() || h->pats.size() >= ((size_t)1 << 22)) return;
  for (size_t i = 1; i < h->pats.size(); ++i) if (h->pats[i].id <= h->pats[i - 1].id) return;
  int ib = 1, eb = 0;
  while (((size_t)1 << ib) < h->pats.size()) ++ib;
  const uint64_t maxend = (uint64_t)h->n + 512;                     // (extensions may end a few characters beyond the stream)
  while (eb < 64 && (maxend >> eb)) ++eb;
  h->sort_idxbits = ib; h->sort_keybits = eb + ib + 2;
  h->sort_dev = h->sort_keybits <= 63;
}

// pattern lengths and ids by pattern index, on the device (clustering, the halves rule, the final sort)
static int ensure_fpat(pm_handle *h) {
  if (h->d_fpat_len) return PM_OK;
  std::vector<uint8_t> pl(h->pats.size()); std::vector<uint32_t> pi(h->pats.size());
  for (size_t i = 0; i < h->pats.size(); ++i) { pl[i] = (uint8_t)std::min<size_t>(h->pats[i].s.size(), 255); pi[i] = (uint32_t)h->pats[i].id; }
  HIP_TRY(h, hipMalloc((void **)&h->d_fpat_len, pl.size() ? pl.size() : 1));
  HIP_TRY(h, hipMalloc((void **)&h->d_fpat_id, pi.size() ? pi.size() * 4 : 4));
  if (!pl.empty()) {
    HIP_TRY(h, hipMemcpy(h->d_fpat_len, pl.data(), pl.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_fpat_id, pi.data(), pi.size() * 4, hipMemcpyHostToDevice));
  }
  return PM_OK;
}

// pm_scan's landing: n_upper bounds the number of final hits at d_hits (their count is d_count on the device when the
// kernels in front produce it, else n_upper itself).  Enqueues the sort; *sorted tells whether the device could.
static int land_enqueue_sort(pm_handle *h, const pm_hit *d_hits, const unsigned long long *d_count, size_t n_upper, bool *sorted) {
  *sorted = false;
  if (n_upper == 0 || n_upper > (size_t)INT_MAX / 2 || !h->sort_dev) return PM_OK;
  { int rcw = ensure_sort_workspace(h, n_upper, false); if (rcw) return rcw; }
  { int rcp = ensure_fpat(h); if (rcp) return rcp; }
  if (h->fsorted_cap < n_upper) {
    if (h->d_fsorted) (void)hipFree(h->d_fsorted);
    h->d_fsorted = nullptr;
    h->fsorted_cap = std::max<size_t>(n_upper + n_upper / 4, (size_t)1 << 16);
    HIP_TRY(h, hipMalloc((void **)&h->d_fsorted, h->fsorted_cap * sizeof(pm_hit)));
  }
  HIP_TRY(h, sort_final_device(d_hits, d_count, n_upper, h->d_fpat_id, (uint32_t)h->pats.size(), h->sort_idxbits, h->sort_keybits,
                               h->d_keys, h->d_keys_alt, h->d_fsorted, h->d_ctemp, h->ctemp_bytes, h->stream));
  *sorted = true;
  return PM_OK;
}

// Wait for the finalize kernels enqueued so far.  pm_scan's form (next): the scan of the next range goes in behind them first.
static int finalize_sync(pm_handle *h, const ScanNext *next) {
  if (!next) { HIP_TRY(h, stream_wait(h)); return PM_OK; }
  HIP_TRY(h, hipEventRecord(h->ev_fin, h->stream));
  if (next->on && !h->spec) {
    if (pm_scan_candidates_async(h, next->b, next->e) == PM_OK) { h->spec = true; h->spec_b = next->b; h->spec_e = next->e; }
  }
  HIP_TRY(h, hipEventSynchronize(h->ev_fin));
  return PM_OK;
}

// nfin final hits (sorted on the device: d_sorted, else as they are at d_hits) + the host-decided `extra` -> landing buffer,
// in (end, pid, k) order.  The copy runs on the copy stream: the handle's stream may hold the next range's scan.
static int land_collect(pm_handle *h, const pm_hit *d_hits, bool sorted, size_t nfin, std::vector<pm_hit> &extra, size_t *n_out) {
  { int rcl = ensure_landing(h, nfin + extra.size()); if (rcl) return rcl; }
  pm_hit *dst = h->land + h->land_n;
  if (nfin) {
    HIP_TRY(h, hipMemcpyAsync(dst, sorted ? h->d_fsorted : d_hits, nfin * sizeof(pm_hit), hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    if (!sorted) {
      for (size_t i = 0; i < nfin; ++i) dst[i].aux[0] = dst[i].aux[1] = dst[i].aux[2] = 0;
      sort_hits(dst, nfin);
    }
  }
  if (!extra.empty()) {
    sort_hits(extra.data(), extra.size());
    memcpy(dst + nfin, extra.data(), extra.size() * sizeof(pm_hit));
    std::inplace_merge(dst, dst + nfin, dst + nfin + extra.size(), by_end_pid);
  }
  h->land_n += nfin + extra.size();
  if (n_out) *n_out = nfin + extra.size();
  return PM_OK;
}

static int finalize_device_impl(pm_handle *h, const void *d_cands, size_t n, int64_t scanned_to, int flags,
                                const OwnedRange &own, pm_hit *out, size_t cap, size_t *n_out, const ScanNext *next) {
  if (!h || !h->inited) return fail(h, PM_E_INVALID, "pm_finalize_device: handle not initialised");
  if (h->host_only) return fail(h, PM_E_UNSUPPORTED, "pm_finalize_device: this handle runs the host stage only (use pm_finalize)");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const pm_hit *src = d_cands ? (const pm_hit *)d_cands : h->d_cands;
  if (!d_cands) n = h->last_count;
  const bool last = flags & PM_FINALIZE_LAST;
  if (n_out) *n_out = 0;
  // out == NULL: the final hits stay in HBM (pm_final_hits_device) for the exchange step of a sharded scan
  const bool land = next != nullptr;
  const bool keep = out == nullptr && !land;
  h->d_final = nullptr; h->n_final = 0;
  if (keep && (flags & PM_FINALIZE_SORTED)) return fail(h, PM_E_INVALID, "pm_finalize_device: PM_FINALIZE_SORTED needs a host buffer");
  const bool passthrough = h->sem == PM_SEM_KEYWORD_TREE || h->sem == PM_SEM_SHIFT_AND || h->sem == PM_SEM_SHIFT_AND_INEXACT ||
                           (h->sem == PM_SEM_EXACT_BASES && h->bases_flags);
  // edits on the seed family (A,C,G,T patterns of <= 32 characters): clusters and their DPs on the device
  const bool cluster_dp = h->sem == PM_SEM_FILTER_BITVEC && h->edits_dev && !h->cfg.wildcards && h->pats.size() < ((size_t)1 << 22);
  const bool cluster = cluster_dp || device_cluster_plain(h);
  // exact_halves on the seed family: its per-pattern sequential rule as a sort + one walk per pattern
  // (pm_halves_rule).  Stateless, so only for a complete range on a fresh engine state.
  const bool halves = h->sem == PM_SEM_EXACT_HALVES && (h->seed_flags || h->halves_dev) && h->pats.size() < ((size_t)1 << 22) - 1;
  std::vector<pm_hit> none;
  if (halves) {
    if (!last || own.on || !h->halves_fresh || !h->carry.empty())
      return fail(h, PM_E_UNSUPPORTED, "pm_finalize_device: exact_halves on the device needs the whole range in one call after pm_reset (use pm_finalize)");
    const size_t m = h->seed_flags ? 2 * n : n;
    { int rcw = ensure_sort_workspace(h, m, true); if (rcw) return rcw; }
    if (h->vals_cap < h->ckeys_cap) {
      void *old[] = {h->d_vals, h->d_vals_alt, h->d_htemp};
      for (void *q : old) if (q) (void)hipFree(q);
      h->d_vals = h->d_vals_alt = nullptr; h->d_htemp = nullptr;
      h->vals_cap = h->ckeys_cap;
      HIP_TRY(h, hipMalloc((void **)&h->d_vals, h->vals_cap * 4));
      HIP_TRY(h, hipMalloc((void **)&h->d_vals_alt, h->vals_cap * 4));
      h->htemp_bytes = halves_temp_bytes(h->vals_cap);
      HIP_TRY(h, hipMalloc(&h->d_htemp, h->htemp_bytes ? h->htemp_bytes : 16));
    }
    { int rcp = ensure_fpat(h); if (rcp) return rcp; }
    HIP_TRY(h, halves_rule_device(src, n, h->seed_flags, h->cfg.indels ? 2 * h->cfg.k : 0, h->d_fpat_len, h->d_fpat_id, h->d_keys, h->d_keys_alt,
                                  h->d_vals, h->d_vals_alt, h->d_htemp, h->htemp_bytes, h->d_fout, h->d_fcounts, h->stream));
    bool sorted = false;
    if (land) { int rcs = land_enqueue_sort(h, h->d_fout, h->d_fcounts, m, &sorted); if (rcs) return rcs; }
    HIP_TRY(h, hipMemcpyAsync(h->h_fcounts, h->d_fcounts, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    { int rcy = finalize_sync(h, next); if (rcy) return rcy; }
    const size_t nfin = (size_t)h->h_fcounts[0];
    if (n_out) *n_out = nfin;
    if (land) return land_collect(h, h->d_fout, sorted, nfin, none, n_out);
    if (keep) { h->d_final = h->d_fout; h->n_final = nfin; return PM_OK; }
    if (nfin > cap) return fail(h, PM_E_OVERFLOW, "pm_finalize_device: out buffer too small");
    if (nfin) HIP_TRY(h, hipMemcpy(out, h->d_fout, nfin * sizeof(pm_hit), hipMemcpyDeviceToHost));
    if (flags & PM_FINALIZE_SORTED) sort_hits(out, nfin);
    return PM_OK;
  }
  if (!passthrough && !cluster) return fail(h, PM_E_UNSUPPORTED, "pm_finalize_device: this option set needs the host stage (pm_finalize)");
  if (passthrough && land) {                                        // the records are the hits: sort, [next scan], copy
    bool sorted = false;
    { int rcs = land_enqueue_sort(h, src, nullptr, n, &sorted); if (rcs) return rcs; }
    { int rcy = finalize_sync(h, next); if (rcy) return rcy; }
    return land_collect(h, src, sorted, n, none, n_out);
  }
  if (passthrough && keep) {
    if (own.on) {                                                   // the records that end in the owned range, compacted on the device
      { int rcw = ensure_sort_workspace(h, n, true); if (rcw) return rcw; }
      HIP_TRY(h, owned_filter_device(src, n, own, h->d_fout, h->d_fcounts, h->stream));
      HIP_TRY(h, hipMemcpyAsync(h->h_fcounts, h->d_fcounts, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, stream_wait(h));
      h->d_final = h->d_fout; h->n_final = (size_t)h->h_fcounts[0];
    } else { h->d_final = src; h->n_final = n; }
    if (n_out) *n_out = h->n_final;
    return PM_OK;
  }
  if (passthrough) {
    if (n > cap) return fail(h, PM_E_OVERFLOW, "pm_finalize_device: out buffer too small");
    if (n) HIP_TRY(h, hipMemcpyAsync(out, src, n * sizeof(pm_hit), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, stream_wait(h));
    if (own.on) n = (size_t)(std::remove_if(out, out + n, [&](const pm_hit &x) { return !(x.end > own.own_lo && x.end <= own.own_hi); }) - out);
    if (flags & PM_FINALIZE_SORTED) sort_hits(out, n);
    if (n_out) *n_out = n;
    return PM_OK;
  }
  // (the device sorts count their items in an int: a hit-dense stream can hand one range more records than that)
  if (n + h->carry.size() >= ((size_t)1 << 31)) return dense_fail(h, "pm_finalize_device, 2^31 or more candidate", n + h->carry.size());
  { int rcw = ensure_sort_workspace(h, n + h->carry.size(), true); if (rcw) return rcw; }
  { int rcp = ensure_fpat(h); if (rcp) return rcp; }
  if (cluster_dp) { int rcd = ensure_dp_tables(h); if (rcd) return rcd; }
  // candidates an earlier range left undecided (clusters that could still grow) join this batch on
  // the device: their chains continue here
  std::vector<pm_hit> hostpart;
  const size_t ncarry = h->carry.size();
  if (ncarry) {
    if (h->d_carry_cap < ncarry) {
      if (h->d_carry) (void)hipFree(h->d_carry);
      h->d_carry = nullptr;
      h->d_carry_cap = ncarry + ncarry / 2 + 1024;
      HIP_TRY(h, hipMalloc((void **)&h->d_carry, h->d_carry_cap * sizeof(pm_hit)));
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_carry, h->carry.data(), ncarry * sizeof(pm_hit), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, stream_wait(h));
    h->carry.clear();
  }
  const double tfd0 = now_ms();
  if (cluster_dp)
    HIP_TRY(h, cluster_dp_device(src, n, h->d_carry, ncarry, h->cfg.k, true, scanned_to, last, h->d_text, h->n, h->eos_code, h->d_dp_codes, h->d_fpat_len,
                                 h->d_dp_esb, h->d_dp_eeb, h->d_fpat_id, own, h->d_keys, h->d_keys_alt, h->d_ctemp, h->ctemp_bytes,
                                 h->d_fout, h->d_fleft, h->d_fcounts, h->stream));
  else
  HIP_TRY(h, cluster_device(src, n, h->d_carry, ncarry, h->cfg.k, scanned_to, last, h->zoned ? 3 : -1, h->d_fpat_len, h->d_fpat_id, own, h->d_keys, h->d_keys_alt,
                            h->d_ctemp, h->ctemp_bytes, h->d_fout, h->d_fleft, h->d_fcounts, h->stream));
  bool sorted = false;
  if (land) { int rcs = land_enqueue_sort(h, h->d_fout, h->d_fcounts, n + ncarry, &sorted); if (rcs) return rcs; }
  h->h_fcounts[2] = 0;
  HIP_TRY(h, hipMemcpyAsync(h->h_fcounts, h->d_fcounts, ((n + ncarry) ? 3 : 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  { int rcy = finalize_sync(h, next); if (rcy) return rcy; }
  const size_t nfin = (size_t)h->h_fcounts[0], nleft = (size_t)h->h_fcounts[1];
  if (own.on && h->h_fcounts[2])
    return fail(h, PM_E_UNSUPPORTED, "pm_finalize_device_owned: a chain of candidates reaches from the guard edge into the owned range (repeat longer than the guard band)");
  const double tfd1 = now_ms();
  if (nleft) {
    const size_t at = hostpart.size();
    hostpart.resize(at + nleft);
    if (land) {                                                     // (the handle's stream may hold the next range's scan)
      HIP_TRY(h, hipMemcpyAsync(hostpart.data() + at, h->d_fleft, nleft * sizeof(pm_hit), hipMemcpyDeviceToHost, h->copy_stream));
      HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    } else
    HIP_TRY(h, hipMemcpy(hostpart.data() + at, h->d_fleft, nleft * sizeof(pm_hit), hipMemcpyDeviceToHost));
  }
  std::vector<pm_hit> extra;
  if (!hostpart.empty()) {
    int rc = finalize_into(h, hostpart.data(), hostpart.size(), scanned_to, last, extra);
    if (rc) return rc;
    if (own.on) extra.erase(std::remove_if(extra.begin(), extra.end(), [&](const pm_hit &x) { return !(x.end > own.own_lo && x.end <= own.own_hi); }), extra.end());
  }
  if (land) {
    const int rcl = land_collect(h, h->d_fout, sorted, nfin, extra, n_out);
    if (h->knobs.debug) fprintf(stderr, "[pm] finalize_device (pm_scan): %zu records, device %.1f ms (%zu finals, %zu left for the host), host part + copies %.1f ms%s\n",
                                n, tfd1 - tfd0, nfin, nleft, now_ms() - tfd1, h->spec ? ", next range's scan in flight" : "");
    return rcl;
  }
  if (keep) {                                                       // the few host-decided hits join the device's in HBM
    if (nfin + extra.size() > h->ckeys_cap) return fail(h, PM_E_OVERFLOW, "pm_finalize_device: device output buffer too small");
    if (!extra.empty()) HIP_TRY(h, hipMemcpy(h->d_fout + nfin, extra.data(), extra.size() * sizeof(pm_hit), hipMemcpyHostToDevice));
    h->d_final = h->d_fout; h->n_final = nfin + extra.size();
    if (n_out) *n_out = h->n_final;
    return PM_OK;
  }
  if (nfin + extra.size() > cap) return fail(h, PM_E_OVERFLOW, "pm_finalize_device: out buffer too small");
  if (nfin) HIP_TRY(h, hipMemcpy(out, h->d_fout, nfin * sizeof(pm_hit), hipMemcpyDeviceToHost));
  if (!extra.empty()) memcpy(out + nfin, extra.data(), extra.size() * sizeof(pm_hit));
  const size_t tot = nfin + extra.size();
  if (h->knobs.debug) fprintf(stderr, "[pm] finalize_device: %zu records, device %.1f ms (%zu finals, %zu left for the host), host part + copies %.1f ms\n", n, tfd1 - tfd0, nfin, nleft, now_ms() - tfd1);
  if (flags & PM_FINALIZE_SORTED) sort_hits(out, tot);
  if (n_out) *n_out = tot;
  return PM_OK;
}

// primer_match's per-hit re-alignment (reference primer_match.cc:1135-1151): exact_alignment
// (pattern_alignment.cc:29-43) for k == 0, else editdist_alignment(key, key, k, eos, wc, tn,
// indels, dm, esb, eeb, yesno=false) with its traceback; editdist == INT32_MAX is the CLI's
// "Bogus hit" (constraint violation).
static int align_hits_impl(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out, char *ops, char *text, size_t stride);

extern "C" int pm_align_hits(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out) {
  return align_hits_impl(h, hits, n, out, nullptr, nullptr, 0);
}

extern "C" int pm_align_hits_text(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out, char *ops, char *text, size_t stride) {
  if (!ops || !text || stride == 0) return fail(h, PM_E_INVALID, "pm_align_hits_text: bad arguments");
  return align_hits_impl(h, hits, n, out, ops, text, stride);
}

static int align_hits_impl(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out, char *ops, char *text, size_t stride) {
  if (!h || !h->inited || (!hits && n) || (!out && n)) return fail(h, PM_E_INVALID, "pm_align_hits: bad arguments");
  const bool wc_exact = h->cfg.wildcards && h->cfg.k == 0;      // exact_wc_alignment (pattern_alignment.cc:70-93) reads the text
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int k = h->cfg.k;
  std::vector<Window> wins(n);
  std::vector<const Pattern *> pp(n);
  for (size_t i = 0; i < n; ++i) {
    auto it = h->id2idx.find(hits[i].pid);
    if (it == h->id2idx.end()) return fail(h, PM_E_INVALID, "pm_align_hits: unknown pattern id");
    pp[i] = &h->pats[it->second];
    const int L = (int)pp[i]->s.size();
    int64_t ws = 0;
    if (hits[i].end > (int64_t)L + k) ws = hits[i].end - L - k;       // pattern_alignment.cc:137-139
    wins[i] = Window{ws, (k == 0 && !wc_exact) ? 0 : (int32_t)(hits[i].end - ws), 0};
  }
  if (k > 0 || wc_exact) { int rc = fetch_windows(h, wins); if (rc) return rc; }
  AlignParams prm; prm.k = k; prm.indels = h->cfg.indels != 0; prm.eos = (uint8_t)h->cfg.eos;
  prm.wc = h->cfg.wildcards != 0; prm.tn = h->cfg.text_n != 0;
  // the hits are independent: worker threads take contiguous slices (own DP scratch each)
  auto run = [&](size_t lo, size_t hi, AlignScratch &scratch) -> bool {
  for (size_t i = lo; i < hi; ++i) {
    const int L = (int)pp[i]->s.size();
    if (wc_exact) {
      // start = end - L; per character: equal, IUPAC-compatible (text N only with -W), or substitution
      const int64_t st = hits[i].end - L;
      if ((size_t)L + 1 > stride && ops) return false;
      int subs = 0;
      for (int q = 0; q < L; ++q) {
        const int64_t tp = st + q;
        const unsigned char tc = tp >= wins[i].start && tp < hits[i].end ? h->winbuf[wins[i].off + (tp - wins[i].start)] : 0;
        const unsigned char pc = (unsigned char)pp[i]->s[q];
        char op;
        if (tc == pc) op = '|';
        else {
          const char *set = tc < 128 ? iupac_compatible_set(tc) : nullptr;
          if (set && pc && strchr(set, pc) && (h->cfg.text_n || tc != 'N')) op = '+';
          else { op = '*'; ++subs; }
        }
        if (ops) { ops[i * stride + q] = op; text[i * stride + q] = (char)tc; }
      }
      if (ops) { ops[i * stride + L] = 0; text[i * stride + L] = 0; }
      out[i].start = st; out[i].end = hits[i].end; out[i].editdist = subs; out[i].value = 0;
      continue;
    }
    if (k == 0) {                                                    // exact_alignment (pattern_alignment.cc:29-43)
      out[i].start = hits[i].end - L; out[i].end = hits[i].end; out[i].editdist = 0; out[i].value = 0;
      if (ops) {
        if ((size_t)L + 1 > stride) return false;
        memset(ops + i * stride, '|', (size_t)L); ops[i * stride + L] = 0;
        memcpy(text + i * stride, pp[i]->s.data(), (size_t)L); text[i * stride + L] = 0;
      }
      continue;
    }
    std::string opstr;
    AlignResult r = editdist_align(h->winbuf.data() + wins[i].off, wins[i].start, hits[i].end, hits[i].end,
                                   pp[i]->s.data(), L, pp[i]->esb, pp[i]->eeb, prm, scratch, ops ? &opstr : nullptr);
    out[i].start = r.start; out[i].end = r.end; out[i].editdist = r.editdist; out[i].value = r.value;
    if (ops) {
      const int64_t tl = r.end - r.start;                              // matching text (pattern_alignment.cc:603-606)
      if (opstr.size() + 1 > stride || tl < 0 || (size_t)tl + 1 > stride) return false;
      memcpy(ops + i * stride, opstr.data(), opstr.size()); ops[i * stride + opstr.size()] = 0;
      for (int64_t q = 0; q < tl; ++q) text[i * stride + q] = (char)h->winbuf[wins[i].off + (r.start - wins[i].start) + q];
      text[i * stride + tl] = 0;
    }
  }
  return true;
  };
  bool ok = true;
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nthreads = n < 4096 ? 1 : std::min<size_t>(hw ? hw : 1, 16);
  if (nthreads <= 1) ok = run(0, n, h->scratch);
  else {
    std::vector<std::thread> pool;
    std::vector<char> good(nthreads, 1);
    const size_t per = (n + nthreads - 1) / nthreads;
    for (size_t t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t]() { AlignScratch sc; good[t] = run(std::min(n, t * per), std::min(n, (t + 1) * per), sc) ? 1 : 0; });
    for (std::thread &th : pool) th.join();
    for (char g : good) ok = ok && g;
  }
  if (!ok) return fail(h, PM_E_INVALID, "pm_align_hits_text: stride too small");
  return PM_OK;
}

// One range of PatternMatch::find_patterns: scan, finalize, final hits in (end, pid, k) order behind whatever the landing
// buffer still holds.  With a device finalize stage the scan of the range expected next -- the same number of stream
// bytes, the way the reference's callers walk a stream chunk by chunk -- is already on the GPU when this returns.
static int scan_piece(pm_handle *h, int64_t begin, int64_t end) {
  if (begin != h->next_begin) return fail(h, PM_E_INVALID, "pm_scan: ranges must be consecutive (pm_reset to restart)");
  size_t cnt = 0;
  int rc;
  h->too_dense = false;
  if (h->spec && h->spec_b == begin && h->spec_e == end && h->scan_pending) { h->spec = false; rc = pm_scan_wait(h, &cnt); }
  else rc = pm_scan_candidates(h, begin, end, nullptr, 0, &cnt);       // (drains a speculative scan of another range)
  while (rc == PM_E_OVERFLOW && cnt > h->cap) {                     // grow and redo this range
    if (cnt > dense_bound(h)) return dense_fail(h, "candidate records", cnt);
    rc = ensure_capacity(h, cnt + cnt / 4 + 1024);
    if (rc) return rc;
    rc = pm_scan_candidates(h, begin, end, nullptr, 0, &cnt);
  }
  if (rc) return rc;
  const bool halves_whole = h->sem == PM_SEM_EXACT_HALVES && (h->seed_flags || h->halves_dev) && begin == 0 && end >= h->n &&
                            h->halves_fresh && h->carry.empty() && h->pats.size() < ((size_t)1 << 22) - 1;
  const bool passthrough = h->sem == PM_SEM_KEYWORD_TREE || h->sem == PM_SEM_SHIFT_AND || h->sem == PM_SEM_SHIFT_AND_INEXACT ||
                           (h->sem == PM_SEM_EXACT_BASES && h->bases_flags);
  if ((h->edits_dev && h->sem == PM_SEM_FILTER_BITVEC && !h->cfg.wildcards && h->pats.size() < ((size_t)1 << 22)) || halves_whole ||
      (device_cluster_plain(h) && h->kern == PM_KERNEL_SEED) || passthrough) {
    // sort, clusters and their DPs on the device; only what it hands back goes through the host stage
    ScanNext next = {end < h->n && !h->dense_mode, end, std::min<int64_t>(h->n, end + (end - begin))};   // (no guess after a cut: the pieces are not the caller's ranges)
    const OwnedRange all = {0, 0, 0, 0, 0};
    rc = finalize_device_impl(h, nullptr, 0, end, end >= h->n ? PM_FINALIZE_LAST : 0, all, nullptr, 0, nullptr, &next);
    if (rc) return rc;
  } else {
    std::vector<pm_hit> cands(cnt), outv;
    if (cnt) HIP_TRY(h, hipMemcpy(cands.data(), h->d_cands, cnt * sizeof(pm_hit), hipMemcpyDeviceToHost));
    rc = finalize_into(h, cands.data(), cnt, end, end >= h->n, outv);
    if (rc) return rc;
    sort_hits(outv.data(), outv.size());
    rc = ensure_landing(h, outv.size());
    if (rc) return rc;
    if (!outv.empty()) memcpy(h->land + h->land_n, outv.data(), outv.size() * sizeof(pm_hit));
    h->land_n += outv.size();
  }
  h->next_begin = end;
  return PM_OK;
}

// A range whose lists would outgrow the bound (hit-dense text: DESIGN 7c) is not grown for but scanned in pieces: consecutive
// ranges give the hits of the whole (filter_bitvec.cc:118-121: what a range cannot decide yet waits for the next), so the
// answer does not change; memory stays bounded and so does the 2^31-item limit of the device sorts.  The piece length halves
// when a piece was too dense (that attempt is thrown away: every check sits before the finalize stage touches the engine's
// state), stays for the ranges that follow -- text that was dense a moment ago mostly still is -- and doubles again after
// a piece whose lists stayed under a quarter of the bound.
static int scan_range(pm_handle *h, int64_t begin, int64_t end) {
  const size_t before = h->land_n - h->land_pos;                    // (ensure_landing may move the live hits to the front of the buffer)
  int pieces = 0;
  for (int64_t pos = begin; pos < end;) {
    const int64_t len = h->piece_len ? std::min<int64_t>(h->piece_len, end - pos) : end - pos;
    h->dense_mode = len < end - begin;                              // (no guess of the next range while in pieces: they are not the caller's ranges)
    h->bound_on = true;
    const int rc = scan_piece(h, pos, pos + len);
    h->bound_on = false;
    if (rc != PM_OK) {
      if (!h->too_dense || len <= 256) return rc;                   // (the message says what was too many)
      h->piece_len = std::max<int64_t>(256, len / 2);
      ++h->range_splits;
      drain_spec(h);
      if (h->knobs.debug) fprintf(stderr, "[pm] pm_scan: (%lld, %lld] in pieces of %lld: %s\n", (long long)pos, (long long)(pos + len), (long long)h->piece_len, h->err.c_str());
      continue;
    }
    pos += len;
    ++pieces;
    if (h->piece_len && len == h->piece_len && h->last_peak < dense_bound(h) / 4) {   // (a full piece: the short one at the end of a range says nothing)
      h->piece_len *= 2;
      if (h->piece_len >= end - begin) h->piece_len = 0;            // whole ranges again
    }
  }
  if (pieces > 1)                                                   // the pieces' hits, each in order, as one run in (end, pid, k) order
    sort_hits(h->land + h->land_pos + before, h->land_n - h->land_pos - before);
  h->dense_mode = false;
  return PM_OK;
}

extern "C" int pm_scan(pm_handle *h, int64_t begin, int64_t end, pm_hit *out, size_t cap, size_t *n_out, int *more) {
  if (!h || !h->inited) return fail(h, PM_E_INVALID, "pm_scan: handle not initialised");
  if (n_out) *n_out = 0;
  if (more) *more = 0;
  if (end > h->n) end = h->n;
  if (end > begin) { const int rc = scan_range(h, begin, end); if (rc) return rc; }
  const size_t avail = h->land_n - h->land_pos;
  const size_t take = out ? std::min(avail, cap) : 0;
  if (take) memcpy(out, h->land + h->land_pos, take * sizeof(pm_hit));
  h->land_pos += take;
  if (h->land_pos == h->land_n) h->land_pos = h->land_n = 0;
  if (n_out) *n_out = take;
  if (more) *more = h->land_pos < h->land_n;
  return PM_OK;
}

extern "C" int pm_scan_view(pm_handle *h, int64_t begin, int64_t end, const pm_hit **hits, size_t *n) {
  if (!h || !h->inited || !hits || !n) return fail(h, PM_E_INVALID, "pm_scan_view: bad arguments");
  *hits = nullptr; *n = 0;
  if (end > h->n) end = h->n;
  if (h->land_pos == h->land_n) h->land_pos = h->land_n = 0;       // the span of the last call is given up
  if (end > begin) { const int rc = scan_range(h, begin, end); if (rc) return rc; }
  *hits = h->land + h->land_pos;
  *n = h->land_n - h->land_pos;
  h->land_pos = h->land_n;                                          // handed out; the memory stays valid until the next call
  return PM_OK;
}

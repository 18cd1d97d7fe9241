// gpu_pattern_match.cc -- see gpu_pattern_match.h.
#include "gpu_pattern_match.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>

namespace pmgpu {

BufferChars::BufferChars(std::vector<unsigned char> bytes, std::string table)
    : bytes_(std::move(bytes)), table_(std::move(table)) {
  data_ = bytes_.data();
  n_ = (int64_t)bytes_.size();
  build_inverse();
}

BufferChars::BufferChars(const unsigned char *data, size_t n, std::string table)
    : data_(data), n_((int64_t)n), table_(std::move(table)) {
  build_inverse();
}

void BufferChars::build_inverse() {
  for (int i = 0; i < 256; ++i) inv_[i] = table_.empty() ? i : -1;
  for (size_t i = 0; i < table_.size(); ++i) inv_[(unsigned char)table_[i]] = (int)i;
}

// a shard holds its own bytes, a guard band either side (filter_bitvec chains of candidates that
// straddle a shard edge are decided by the shard that owns the hit, pm_finalize_device_owned) and a
// halo of text the windows / seed extensions of the outermost candidates read
static const int64_t SHARD_GUARD = 1 << 16, SHARD_HALO = 256;

GpuPatternMatch::GpuPatternMatch(int kernel, unsigned int k, char eos, bool wc, bool tn, bool indels,
                                 bool dna_mut, int semantics, int device, RankGroup *group) {
  if (dna_mut) { fprintf(stderr, "Fatal error: DNA mutation scoring is not available in the GPU engine.\n"); exit(1); }
  group_ = group && !group->single() ? group : nullptr;
  pm_config cfg = {};
  cfg.abi_version = PM_ABI_VERSION;
  cfg.semantics = semantics; cfg.kernel = kernel; cfg.k = (int32_t)k; cfg.indels = indels ? 1 : 0;
  cfg.wildcards = wc ? 1 : 0; cfg.text_n = tn ? 1 : 0; cfg.eos = (unsigned char)eos;
  cfg.device = group_ ? group_->device() : device;
  if (pm_create(&cfg, &h_) != PM_OK) { fprintf(stderr, "Fatal error: %s\n", pm_last_error(nullptr)); exit(1); }
  if (group_ && group_->rank() == 0 && pm_create(&cfg, &merge_) != PM_OK) { fprintf(stderr, "Fatal error: %s\n", pm_last_error(nullptr)); exit(1); }
}

GpuPatternMatch::~GpuPatternMatch() {
  if (comm_) pm_comm_destroy(comm_);
  if (merge_) pm_destroy(merge_);
  pm_destroy(h_);
}

void GpuPatternMatch::fatal(const char *what) const {
  fprintf(stderr, "Fatal error: %s: %s\n", what, pm_last_error(h_));          // timestamp()+exit(1) convention
  exit(1);
}

unsigned long GpuPatternMatch::add_pattern(std::string const &pat, unsigned long id, int esb, int eeb) {
  if (id == 0) id = ++next_id_;                                               // pattern_match.h:92-94
  if (pm_add_pattern(h_, pat.data(), pat.size(), id, esb, eeb) != PM_OK) fatal("add_pattern");
  if (merge_ && pm_add_pattern(merge_, pat.data(), pat.size(), id, esb, eeb) != PM_OK) fatal("add_pattern");
  return id;
}

void GpuPatternMatch::init(CharacterProducer &cp) {
  // the engine needs cp only for nch()/ch()/size() and the bytes (SURVEY 8b "Text access")
  std::string table;
  if (cp.size() < 256) for (unsigned i = 0; i < cp.size(); ++i) table.push_back(cp.ch((unsigned char)i));
  const unsigned char *bytes;
  if (cp.has_filename() && cp.c_str()) {                                      // mmap path: char_io.h:167-169
    bytes = reinterpret_cast<const unsigned char *>(cp.c_str());
    n_ = cp.length();
  } else {                                                                    // BufferedFileChars: drain once
    const int64_t save = cp.pos();
    cp.pos(0);
    owned_.clear();
    while (!cp.eof()) owned_.push_back(cp.getnch());
    cp.pos(save);
    bytes = owned_.data();
    n_ = (int64_t)owned_.size();
  }
  const uint8_t *tb = table.empty() ? nullptr : reinterpret_cast<const uint8_t *>(table.data());
  if (!group_) {
    if (pm_init(h_, bytes, n_, tb, (int32_t)table.size()) != PM_OK) fatal("init");
    return;
  }
  // position shard of this rank: its GPU holds stream bytes [glo_, ghi_) only, and
  // scans in local indices; rank 0 also keeps a host-stage handle over the whole stream
  const int world = group_->world(), rank = group_->rank();
  shard_ = (n_ + world - 1) / world;
  lo_ = std::min<int64_t>(n_, (int64_t)rank * shard_); hi_ = std::min<int64_t>(n_, lo_ + shard_);
  glo_ = std::max<int64_t>(0, lo_ - SHARD_GUARD - SHARD_HALO); ghi_ = std::min<int64_t>(n_, hi_ + SHARD_GUARD + SHARD_HALO);
  if (pm_init(h_, bytes + glo_, ghi_ - glo_, tb, (int32_t)table.size()) != PM_OK) fatal("init");
  if (merge_ && pm_init_host(merge_, bytes, n_, tb, (int32_t)table.size()) != PM_OK) {
    fprintf(stderr, "Fatal error: init (merge rank): %s\n", pm_last_error(merge_));
    exit(1);
  }
  if (group_->rccl()) {
    unsigned char id[PM_COMM_ID_BYTES] = {0};
    int rc = rank == 0 ? pm_comm_unique_id(id) : PM_OK;
    if (rc != PM_OK) { fprintf(stderr, "Fatal error: RCCL: %s\n", pm_comm_last_error(nullptr)); exit(1); }
    group_->broadcast(id, sizeof(id));
    if (pm_comm_create(group_->device(), rank, world, id, &comm_) != PM_OK) { fprintf(stderr, "Fatal error: RCCL: %s\n", pm_comm_last_error(nullptr)); exit(1); }
  }
}

// One sharded pass over the whole stream.  Every rank: device stage over its shard; for
// filter_bitvec option sets also the clustering / verify of what it owns (the shards' outputs then
// concatenate to the serial result); count exchange; records to rank 0 (RCCL out of HBM, or the
// launcher's pipes when ranks share a card).  Rank 0: local -> global stream indices, host stage for
// the other option sets (on the handle that knows the whole stream), hits in (end, id) order.
bool GpuPatternMatch::sharded_scan(CharacterProducer &cp, pattern_hit_vector &hits) {
  const int world = group_->world(), rank = group_->rank();
  const int64_t begin = lo_ - glo_, end = hi_ - glo_, nloc = ghi_ - glo_;
  const int64_t g_lo = glo_ == 0 ? 0 : begin - SHARD_GUARD, g_hi = ghi_ == n_ ? nloc : end + SHARD_GUARD;
  const bool own_path = pm_selected_semantics(h_) == PM_SEM_FILTER_BITVEC;
  auto scan = [&](int64_t a, int64_t b) -> size_t {
    size_t cnt = 0;
    int rc = pm_scan_candidates(h_, a, b, nullptr, 0, &cnt);
    while (rc == PM_E_OVERFLOW) {                                  // grow and scan again: local, the exchange comes later
      if (pm_set_capacity(h_, cnt + cnt / 4 + 1024) != PM_OK) fatal("set_capacity");
      rc = pm_scan_candidates(h_, a, b, nullptr, 0, &cnt);
    }
    if (rc != PM_OK) fatal("scan");
    return cnt;
  };
  const uint64_t FAILED = ~0ull, FALLBACK = ~0ull - 1;
  void *d_rec = nullptr;
  size_t nrec = 0;
  bool owned = own_path;
  uint64_t mine = 0;
  if (own_path) {
    if (hi_ > lo_) {
      scan(g_lo, g_hi);
      const int rc = pm_finalize_device_owned(h_, nullptr, 0, begin, end, g_lo, ghi_ == n_ ? INT64_MAX : g_hi, 0, nullptr, 0, &nrec);
      if (rc == PM_OK) { if (pm_final_hits_device(h_, &d_rec, &nrec) != PM_OK) fatal("final hits"); mine = nrec; }
      // not decidable on the shard (an option set whose verify runs on the host, or a repeat longer
      // than the guard band across the shard edge): every rank falls back to sending candidates,
      // and rank 0 clusters and verifies them against the whole stream
      else if (rc == PM_E_UNSUPPORTED) mine = FALLBACK;
      else { fprintf(stderr, "Fatal error: rank %d: %s\n", rank, pm_last_error(h_)); mine = FAILED; }
    }
  }
  std::vector<uint64_t> counts;
  if (own_path) {
    group_->all_gather(mine, &counts);
    for (uint64_t c : counts) if (c == FAILED) group_->leave(1);
    for (uint64_t c : counts) if (c == FALLBACK) owned = false;
  }
  if (!owned) {
    d_rec = nullptr; nrec = 0;
    if (hi_ > lo_) {
      scan(begin, end);
      if (pm_candidates_device(h_, &d_rec, &nrec) != PM_OK) fatal("candidates");
    }
    group_->all_gather(nrec, &counts);
  }
  std::vector<pm_hit> all;
  if (comm_) {
    size_t total = 0;
    for (uint64_t c : counts) total += (size_t)c;
    if (rank == 0) all.resize(total);
    if (pm_comm_gather(comm_, d_rec, nrec, counts.data(), rank == 0 ? all.data() : nullptr) != PM_OK) {
      fprintf(stderr, "Fatal error: rank %d: RCCL gather: %s\n", rank, pm_comm_last_error(comm_));
      group_->leave(1);
    }
  } else {
    std::vector<pm_hit> minev(nrec);
    if (nrec && pm_copy_records(h_, d_rec, nrec, minev.data()) != PM_OK) fatal("copy records");
    group_->gather_host(minev.data(), nrec, counts, &all);
  }
  sharded_done_ = true;
  cp.pos(n_);
  if (rank != 0) return false;
  size_t at = 0;
  for (int r = 0; r < world; ++r) {                                // local -> global stream index
    const int64_t off = std::max<int64_t>(0, std::min<int64_t>(n_, (int64_t)r * shard_) - SHARD_GUARD - SHARD_HALO);
    for (size_t i = 0; i < (size_t)counts[(size_t)r]; ++i) all[at + i].end += off;
    at += (size_t)counts[(size_t)r];
  }
  std::vector<pm_hit> fin;
  if (owned) fin.swap(all);
  else {
    fin.resize(all.size() * 2 + 16);                               // exact_halves -K: up to two seeds per record
    size_t nout = 0;
    if (pm_finalize(merge_, all.data(), all.size(), n_, PM_FINALIZE_LAST, fin.data(), fin.size(), &nout) != PM_OK) {
      fprintf(stderr, "Fatal error: finalize (merge rank): %s\n", pm_last_error(merge_));
      group_->leave(1);
    }
    fin.resize(nout);
  }
  std::sort(fin.begin(), fin.end(), [](const pm_hit &a, const pm_hit &b) { return a.end != b.end ? a.end < b.end : (a.pid != b.pid ? a.pid < b.pid : a.k < b.k); });
  for (const pm_hit &x : fin) hits.push_back(pattern_hit{x.end, x.pid, x.k});
  return !fin.empty();
}

bool GpuPatternMatch::find_patterns(CharacterProducer &cp, pattern_hit_vector &hits, unsigned long minka) {
  // Resumable like the reference engines: scan on from cp.pos(), stop once >= minka hits were
  // appended or the stream ends; cp.pos() is left at the scanned-to position and every hit
  // returned has key <= cp.pos() (primer_match.cc:1121, filter_bitvec.cc:91).
  if (group_) return sharded_done_ ? false : sharded_scan(cp, hits);
  if (cp.eof()) return false;
  unsigned long got = 0;
  while (true) {
    const int64_t begin = cp.pos();
    const int64_t end = begin + chunk_ < n_ ? begin + chunk_ : n_;
    const pm_hit *recs = nullptr;
    size_t cnt = 0;
    // the records stay in the library's buffer; the GPU already scans the next range while they are pushed
    if (pm_scan_view(h_, begin, end, &recs, &cnt) != PM_OK) fatal("find_patterns");
    cp.pos(end);
    for (size_t i = 0; i < cnt; ++i) hits.push_back(pattern_hit{recs[i].end, recs[i].pid, recs[i].k});
    got += cnt;
    if (got >= minka || cp.eof()) return got > 0;
  }
}

void GpuPatternMatch::reset() { if (pm_reset(h_) != PM_OK) fatal("reset"); }
int GpuPatternMatch::selected_semantics() const { return pm_selected_semantics(h_); }
int GpuPatternMatch::selected_kernel() const { return pm_selected_kernel(h_); }

}  // namespace pmgpu

// gpu_pattern_match.cc -- see gpu_pattern_match.h.
#include "gpu_pattern_match.h"

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

GpuPatternMatch::GpuPatternMatch(int kernel, unsigned int k, char eos, bool wc, bool tn, bool indels,
                                 bool dna_mut, int semantics, int device) {
  if (dna_mut) { fprintf(stderr, "Fatal error: DNA mutation scoring is not available in the GPU engine.\n"); exit(1); }
  pm_config cfg = {};
  cfg.abi_version = PM_ABI_VERSION;
  cfg.semantics = semantics; cfg.kernel = kernel; cfg.k = (int32_t)k; cfg.indels = indels ? 1 : 0;
  cfg.wildcards = wc ? 1 : 0; cfg.text_n = tn ? 1 : 0; cfg.eos = (unsigned char)eos; cfg.device = device;
  if (pm_create(&cfg, &h_) != PM_OK) { fprintf(stderr, "Fatal error: %s\n", pm_last_error(nullptr)); exit(1); }
}

GpuPatternMatch::~GpuPatternMatch() { pm_destroy(h_); }

void GpuPatternMatch::fatal(const char *what) const {
  fprintf(stderr, "Fatal error: %s: %s\n", what, pm_last_error(h_));          // timestamp()+exit(1) convention
  exit(1);
}

unsigned long GpuPatternMatch::add_pattern(std::string const &pat, unsigned long id, int esb, int eeb) {
  if (id == 0) id = ++next_id_;                                               // pattern_match.h:92-94
  if (pm_add_pattern(h_, pat.data(), pat.size(), id, esb, eeb) != PM_OK) fatal("add_pattern");
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
  if (pm_init(h_, bytes, n_, table.empty() ? nullptr : reinterpret_cast<const uint8_t *>(table.data()),
              (int32_t)table.size()) != PM_OK)
    fatal("init");
}

bool GpuPatternMatch::find_patterns(CharacterProducer &cp, pattern_hit_vector &hits, unsigned long minka) {
  // Resumable like the reference engines: scan on from cp.pos(), stop once >= minka hits were
  // appended or the stream ends; cp.pos() is left at the scanned-to position and every hit
  // returned has key <= cp.pos() (primer_match.cc:1121, filter_bitvec.cc:91).
  if (cp.eof()) return false;
  unsigned long got = 0;
  std::vector<pm_hit> buf((size_t)1 << 16);
  while (true) {
    const int64_t begin = cp.pos();
    const int64_t end = begin + chunk_ < n_ ? begin + chunk_ : n_;
    size_t cnt = 0;
    int more = 0;
    if (pm_scan(h_, begin, end, buf.data(), buf.size(), &cnt, &more) != PM_OK) fatal("find_patterns");
    cp.pos(end);
    for (;;) {
      for (size_t i = 0; i < cnt; ++i) hits.push_back(pattern_hit{buf[i].end, buf[i].pid, buf[i].k});
      got += cnt;
      if (!more) break;
      if (pm_scan(h_, end, end, buf.data(), buf.size(), &cnt, &more) != PM_OK) fatal("find_patterns");
    }
    if (got >= minka || cp.eof()) return got > 0;
  }
}

void GpuPatternMatch::reset() { if (pm_reset(h_) != PM_OK) fatal("reset"); }
int GpuPatternMatch::selected_semantics() const { return pm_selected_semantics(h_); }
int GpuPatternMatch::selected_kernel() const { return pm_selected_kernel(h_); }

}  // namespace pmgpu

// gpu_pattern_match.cc -- see gpu_pattern_match.h (plugin of the reference tree over include/pm_gpu.h).
#include "gpu_pattern_match.h"

#include <cstdlib>
#include <string>

#include "util.h"            // reference: timestamp()

gpu_pattern_match::gpu_pattern_match(int kernel, unsigned int k, char eos, bool wc, bool tn, bool indels, bool dna_mut)
    : h_(0), n_(0), base_(0), chunk_((FILE_POSITION_TYPE)1 << 30) {
  if (dna_mut) {
    timestamp("Fatal error: DNA mutation scoring is not available in the GPU engine.");
    exit(1);
  }
  pm_config cfg = pm_config();
  cfg.abi_version = PM_ABI_VERSION;
  cfg.semantics = PM_SEM_AUTO;           // reproduce the hit set of the engine pick_pattern_index would choose
  cfg.kernel = kernel;
  cfg.k = (int32_t)k;
  cfg.indels = indels ? 1 : 0;
  cfg.wildcards = wc ? 1 : 0;
  cfg.text_n = tn ? 1 : 0;
  cfg.eos = (unsigned char)eos;
  if (const char *dev = getenv("PM_GPU_DEVICE")) cfg.device = atoi(dev);
  if (pm_create(&cfg, &h_) != PM_OK) {
    std::string msg = std::string("Fatal error: ") + pm_last_error(0);
    timestamp(msg.c_str());
    exit(1);
  }
  if (const char *c = getenv("PM_GPU_CHUNK")) { const long long v = atoll(c); if (v > 0) chunk_ = v; }
}

gpu_pattern_match::~gpu_pattern_match() { pm_destroy(h_); }

void gpu_pattern_match::fatal(const char *what) {                 // pattern_match.h:122-123 convention
  std::string msg = std::string("Fatal error: GPU engine, ") + what + ": " + pm_last_error(h_);
  timestamp(msg.c_str());
  exit(1);
}

long unsigned int gpu_pattern_match::add_pattern(std::string const &pat, unsigned long id, int esb, int eeb) {
  pattern_list::const_iterator it = add_pattern_(pat, id, esb, eeb);            // assigns id when 0 (pattern_match.h:89-103)
  if (by_id_.size() <= id) by_id_.resize(id + 1, patterns().end());
  by_id_[id] = it;
  if (pm_add_pattern(h_, pat.data(), pat.size(), id, esb, eeb) != PM_OK) fatal("add_pattern");
  return id;
}

void gpu_pattern_match::init(CharacterProducer &cp) {
  // alphabet: cp.ch(0..size-1) of a Normalized<> stream (char_io.t:216-246); raw streams have 256 codes
  std::string table;
  if (cp.size() < 256) for (unsigned int i = 0; i < cp.size(); ++i) table.push_back(cp.ch((unsigned char)i));
  const FILE_POSITION_TYPE save = cp.pos();
  cp.reset();
  base_ = cp.pos();
  const unsigned char *bytes;
  if (cp.has_filename()) {                                        // mapped file: the bytes getnch() hands out (char_io.h:167-169)
    bytes = reinterpret_cast<const unsigned char *>(cp.c_str());
    n_ = cp.length();
  } else {                                                        // BufferedFileChars and friends: read the stream once
    drained_.clear();
    while (!cp.eof()) drained_.push_back(cp.getnch());
    bytes = drained_.empty() ? reinterpret_cast<const unsigned char *>("") : &drained_[0];
    n_ = (FILE_POSITION_TYPE)drained_.size();
  }
  cp.pos(save);
  if (pm_init(h_, bytes, (int64_t)n_, table.empty() ? 0 : reinterpret_cast<const uint8_t *>(table.data()),
              (int32_t)table.size()) != PM_OK)
    fatal("init");
}

bool gpu_pattern_match::find_patterns(CharacterProducer &cp, pattern_hit_vector &pas, long unsigned minka) {
  // The reference's callers loop `while (find_patterns(...) || !l.empty())` and read cp.pos() right
  // after the call as "scanned up to here" (primer_match.cc:1118-1121, pcr_match.cc:952,1057): every
  // hit returned ends at or before cp.pos(), hits arrive in non-decreasing end order.
  // One range of chunk_ stream bytes per pm_scan_view call (1 GiB unless PM_GPU_CHUNK says otherwise: a first hit after
  // ~5 ms of scanning; 0.27 ms of finalize and host round trip per range are 6 % of a 3 Gbp pass at this size and 20 % in
  // 256 MiB ranges; on hit-dense text the library scans a range in pieces by itself, include/pm_gpu.h); the records are
  // pushed straight from the library's buffer, and while they are the GPU already scans the next range (pm_scan_view).
  long unsigned got = 0;
  for (;;) {
    const FILE_POSITION_TYPE begin = cp.pos() - base_;
    if (begin >= n_) return got > 0;
    const FILE_POSITION_TYPE end = begin + chunk_ < n_ ? begin + chunk_ : n_;
    const pm_hit *recs = 0;
    size_t cnt = 0;
    if (pm_scan_view(h_, (int64_t)begin, (int64_t)end, &recs, &cnt) != PM_OK) fatal("find_patterns");
    cp.pos(end + base_);
    for (size_t i = 0; i < cnt; ++i) {
      const pm_hit &r = recs[i];
      pas.push_back((FILE_POSITION_TYPE)r.end + base_, std::make_pair(by_id_[r.pid], (unsigned char)r.k));
    }
    got += cnt;
    report_progress(cp);
    if (got >= minka) return true;
  }
}

void gpu_pattern_match::reset() {
  if (pm_reset(h_) != PM_OK) fatal("reset");
}

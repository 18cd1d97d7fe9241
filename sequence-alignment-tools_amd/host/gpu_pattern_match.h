// gpu_pattern_match.h -- C++ host side above the C ABI (include/pm_gpu.h).
//
// GpuPatternMatch has the shape of the reference's PatternMatch plugin
// (reference pattern_match.h:84-156): add_pattern / init / find_patterns / reset with the same
// argument meaning, hit triples and error convention (message on stderr, exit(1);
// pattern_match.h:122-123, select.cc:88-89).  Inside the reference tree it would derive from
// PatternMatch and take the reference's CharacterProducer; INTEGRATION.md shows that 30-line
// adapter and the select.cc hunk.  Stand-alone (this repo) it uses the minimal CharacterProducer
// view below, which declares exactly the virtuals of char_io.h:18-71 an engine may call.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pm_gpu.h"
#include "pm_ranks.h"

namespace pmgpu {

class CharacterProducer {                       // reference char_io.h:18-71 (subset used by engines)
 public:
  virtual ~CharacterProducer() {}
  virtual unsigned char getnch() = 0;           // next stream byte (normalized code), advances
  virtual char ch(unsigned char nch) = 0;       // code -> character
  virtual int nch(char ch) = 0;                 // character -> code, -1 if absent
  virtual unsigned int size() const = 0;        // alphabet size (256 for raw streams)
  virtual int64_t length() const = 0;
  virtual bool eof() const = 0;
  virtual int64_t pos() const = 0;
  virtual void pos(int64_t p) = 0;
  virtual bool has_filename() const { return false; }
  virtual const char *c_str() const { return nullptr; }   // contiguous bytes when has_filename()
};

// A stream held in memory (what MapFileChars / Normalized<MapFileChars> are to the reference).
class BufferChars : public CharacterProducer {
 public:
  BufferChars(std::vector<unsigned char> bytes, std::string table);   // table empty = raw stream
  // bytes owned by someone else (a file mapping, seq_io.h) who outlives this object
  BufferChars(const unsigned char *data, size_t n, std::string table);
  unsigned char getnch() override { return data_[pos_++]; }
  char ch(unsigned char c) override { return table_.empty() ? (char)c : table_[c]; }
  int nch(char c) override { return inv_[(unsigned char)c]; }
  unsigned int size() const override { return table_.empty() ? 256u : (unsigned)table_.size(); }
  int64_t length() const override { return n_; }
  bool eof() const override { return pos_ >= n_; }
  int64_t pos() const override { return pos_; }
  void pos(int64_t p) override { pos_ = p; }
  bool has_filename() const override { return true; }
  const char *c_str() const override { return reinterpret_cast<const char *>(data_); }
 private:
  void build_inverse();
  std::vector<unsigned char> bytes_;
  const unsigned char *data_ = nullptr;
  int64_t n_ = 0;
  std::string table_;
  int inv_[256];
  int64_t pos_ = 0;
};

struct pattern_hit {                            // one element of pattern_hit_vector (pattern_match.h:82)
  int64_t key;                                  // stream index after the last matched char
  unsigned long id;                             // pattern_list_element::id()
  unsigned char value;                          // errors
};
typedef std::vector<pattern_hit> pattern_hit_vector;

class GpuPatternMatch {
 public:
  // kernel: PM_KERNEL_AUTO / PM_KERNEL_BITPAR (-N 16) / PM_KERNEL_SEED (-N 17); the other
  // arguments are pick_pattern_index's (select.cc:19-30).  semantics forces a reference engine.
  // group: the ranks of a position-sharded run (pm_ranks.h), or null / single.  With N ranks every
  // rank scans its own shard of the stream on its own GPU and rank 0's find_patterns hands out the
  // hits of the WHOLE stream; the other ranks' find_patterns returns nothing.
  GpuPatternMatch(int kernel, unsigned int k, char eos = '\n', bool wc = false, bool tn = false,
                  bool indels = true, bool dna_mut = false, int semantics = PM_SEM_AUTO, int device = 0,
                  RankGroup *group = nullptr);
  ~GpuPatternMatch();
  unsigned long add_pattern(std::string const &pat, unsigned long id = 0, int exact_start_bases = 0,
                            int exact_end_bases = 0);                         // pattern_match.h:116
  void init(CharacterProducer &cp);                                           // pattern_match.h:130
  bool find_patterns(CharacterProducer &cp, pattern_hit_vector &hits, unsigned long minka = 1);  // :131
  void reset();                                                               // :134
  int selected_semantics() const;
  int selected_kernel() const;
  void chunk_bytes(int64_t c) { chunk_ = c; }
  // for the caller's per-hit re-alignment (pm_align_hits_text): the handle that knows the whole stream
  pm_handle *handle() const { return merge_ ? merge_ : h_; }
 private:
  [[noreturn]] void fatal(const char *what) const;
  bool sharded_scan(CharacterProducer &cp, pattern_hit_vector &hits);
  pm_handle *h_ = nullptr;
  pm_handle *merge_ = nullptr;                  // rank 0 of a sharded run: host-stage handle over the whole stream (pm_init_host)
  pm_comm *comm_ = nullptr;                     // RCCL communicator when every rank has its own GPU
  RankGroup *group_ = nullptr;
  int64_t shard_ = 0, lo_ = 0, hi_ = 0, glo_ = 0, ghi_ = 0;   // this rank's slice of the stream and what it holds around it
  bool sharded_done_ = false;
  std::vector<unsigned char> owned_;            // stream drained from a producer without c_str()
  int64_t n_ = 0;
  int64_t chunk_ = (int64_t)1 << 30;
  unsigned long next_id_ = 0;
};

}  // namespace pmgpu

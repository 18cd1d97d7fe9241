// gpu_pattern_match.h -- the MI355X engine as a plugin of the REFERENCE tree.
//
// This is the file a maintainer adds next to the reference's own engines (shift_and.h,
// keyword_tree.h, ...): `class gpu_pattern_match` derives from the reference's PatternMatch
// (reference pattern_match.h:84-156) and is handed the reference's CharacterProducer
// (char_io.h:18-71); everything it computes goes through the C ABI of include/pm_gpu.h.
// It is compiled against the reference headers where they lie (-I$(REF)); nothing of the
// reference is copied here.  oracle/Makefile `plugin` links it with the reference's own
// primer_match.o / pcr_match.o into oracle/_ref/{primer_match,pcr_match}_gpu, and
// tests/test_gpu_ref_plugin.py runs those binaries against the golden CLI output.
#ifndef PM_GPU_PATTERN_MATCH_PLUGIN_H
#define PM_GPU_PATTERN_MATCH_PLUGIN_H

#include <vector>

#include "pattern_match.h"   // reference: PatternMatch, pattern_hit_vector, pattern_list, CharacterProducer

#include "pm_gpu.h"          // this repo: include/pm_gpu.h

class gpu_pattern_match : public PatternMatch {
 public:
  // kernel = PM_KERNEL_BITPAR (-N 16) / PM_KERNEL_AUTO (-N 17: seed kernels where the option set
  // allows them); the other arguments are pick_pattern_index's (select.cc:19-30)
  gpu_pattern_match(int kernel, unsigned int k, char eos, bool wc, bool tn, bool indels, bool dna_mut);
  ~gpu_pattern_match();
  long unsigned int add_pattern(std::string const &pat, unsigned long id = 0,
                                int exact_start_bases = 0, int exact_end_bases = 0);
  void init(CharacterProducer &cp);
  bool find_patterns(CharacterProducer &cp, pattern_hit_vector &pas, long unsigned minka = 1);
  void reset();

 private:
  void fatal(const char *what);
  pm_handle *h_;
  std::vector<pattern_list::const_iterator> by_id_;   // id -> list element, for value.first of a hit
  std::vector<unsigned char> drained_;                // stream bytes of a producer without c_str()
  FILE_POSITION_TYPE n_;                              // stream bytes
  FILE_POSITION_TYPE base_;                           // cp.pos() of stream byte 0 (fasta_io.t:234-235 offset_)
  FILE_POSITION_TYPE chunk_;
};

#endif

// select_gpu.cc -- pick_pattern_index with the two GPU cases.
//
// In the reference tree the maintainer adds two `case` lines to select.cc:197-265 (INTEGRATION.md
// section 1.2).  To link the reference's own main() programs WITHOUT editing or copying select.cc,
// oracle/Makefile compiles the reference's select.cc in place with
// -Dpick_pattern_index=ref_pick_pattern_index; this file then provides pick_pattern_index:
//   -N 16  bit-parallel kernel family      -N 17  seed kernel family where the option set allows
//   -N 0   with PM_GPU_AUTO=1 in the environment: 17 (what "auto-select when a gfx950 device is
//          visible" looks like without touching the reference's command line)
//   anything else: the reference's own choice, unchanged.
#include <cstdlib>

#include "select.h"          // reference: declaration of pick_pattern_index
#include "util.h"

#include "gpu_pattern_match.h"

PatternMatch *ref_pick_pattern_index(CharacterProducer * const &ff, int pmselect, int nmismatch,
                                     std::vector<std::pair<int, int> > *exact_const, std::vector<int> *patlen,
                                     int seedlen, bool wildcard, bool textn, bool indels, bool dna_mut, char eos,
                                     bool verbose);

PatternMatch *pick_pattern_index(CharacterProducer * const &ff, int pmselect, int nmismatch,
                                 std::vector<std::pair<int, int> > *exact_const, std::vector<int> *patlen,
                                 int seedlen, bool wildcard, bool textn, bool indels, bool dna_mut, char eos,
                                 bool verbose) {
  if (pmselect == 0 && !dna_mut) {
    const char *a = getenv("PM_GPU_AUTO");
    if (a && *a == '1') pmselect = 17;
  }
  if (pmselect != 16 && pmselect != 17)
    return ref_pick_pattern_index(ff, pmselect, nmismatch, exact_const, patlen, seedlen, wildcard, textn, indels, dna_mut, eos, verbose);
  if (verbose) timestamp(pmselect == 16 ? "Using MI355X bit-parallel kernels..." : "Using MI355X seed-filter kernels...");
  // the "edits >= inexact bases" fatal check of select.cc:87-90 runs inside the engine's init
  // (pm_pick_semantics) with the patterns' own lengths and exact zones
  return new gpu_pattern_match(pmselect == 16 ? PM_KERNEL_BITPAR : PM_KERNEL_AUTO, (unsigned)nmismatch, eos, wildcard, textn, indels, dna_mut);
}

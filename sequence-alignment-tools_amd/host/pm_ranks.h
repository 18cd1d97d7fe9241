// pm_ranks.h -- one process per GPU for the command lines: launcher, rendezvous
// and the count exchange of a position-sharded scan; the records themselves travel over RCCL
// (include/pm_gpu.h pm_comm_*) when every rank has a GPU of its own, and through the launcher's
// pipes when ranks share one (the two-ranks-on-one-card rehearsal).  The reference has no
// counterpart: primer_match / pcr_match scan the stream in one serial pass
// (primer_match.cc:1118, pcr_match.cc:948).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pm_gpu.h"

namespace pmgpu {

// `--ranks N` / `--ranks=N` anywhere on the command line (removed from argv), else $PM_RANKS, else 1
int take_ranks_option(int *argc, char **argv);

class RankGroup {
 public:
  RankGroup() {}
  // Fork `world` rank processes and return in each of them (rank() tells which); the launching
  // process waits for them and exits with the first non-zero status.  Must run before anything
  // touches the GPU.  world <= 1: returns at once, single rank, no pipes.
  static RankGroup launch(int world);
  int rank() const { return rank_; }
  int world() const { return world_; }
  bool single() const { return world_ <= 1; }
  bool rccl() const { return rccl_; }          // every rank on a GPU of its own: records go over xGMI
  int device() const { return device_; }        // HIP device ordinal of this rank
  // every rank's value, in rank order, on every rank
  void all_gather(uint64_t mine, std::vector<uint64_t> *all);
  // `bytes` bytes from rank 0 to every rank (RCCL's unique id)
  void broadcast(void *buf, size_t bytes);
  // host transport: rank r's n records to rank 0 (all = concatenation in rank order, rank 0 only)
  void gather_host(const pm_hit *mine, size_t n, const std::vector<uint64_t> &counts, std::vector<pm_hit> *all);
  [[noreturn]] void leave(int status);          // rank > 0: done (no output of its own)
 private:
  int rank_ = 0, world_ = 1, device_ = 0;
  bool rccl_ = false;
  std::vector<int> up_, down_;                   // rank 0: read ends of r -> 0 / write ends of 0 -> r; rank r: its own two ends at [r]
};

}  // namespace pmgpu

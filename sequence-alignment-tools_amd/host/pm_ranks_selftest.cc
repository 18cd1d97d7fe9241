// pm_ranks_selftest.cc -- the rank launcher and its pipe transport without a GPU: N forked ranks
// exchange counts (all_gather), a broadcast block and hit records (gather_host), and check what they
// receive.  Exit status 0 = every rank saw what it should.  Run by tests/test_ranks_transport.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>

#include "pm_ranks.h"

using namespace pmgpu;

int main(int argc, char **argv) {
  const int world = take_ranks_option(&argc, argv);
  // --die R: rank R exits with status 3 right away and every other rank waits on something no pipe will ever end (the
  // transfers of a device transport look like that): the launcher has to end them and report the failure
  int die = -1;
  for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], "--die")) die = atoi(argv[i + 1]);
  RankGroup g = RankGroup::launch(world);
  const int r = g.rank();
  if (die >= 0 && !g.single()) {
    if (r == die) g.leave(3);
    sleep(600);
    g.leave(0);
  }
  int bad = 0;
  for (int round = 0; round < 3; ++round) {
    std::vector<uint64_t> all;
    g.all_gather((uint64_t)(1000 * round + 7 * r + 1), &all);
    if ((int)all.size() != g.world()) bad = 1;
    for (int q = 0; q < g.world() && !bad; ++q) if (all[(size_t)q] != (uint64_t)(1000 * round + 7 * q + 1)) bad = 2;
    unsigned char id[128];
    for (int i = 0; i < 128; ++i) id[i] = r == 0 ? (unsigned char)(i * 3 + round) : 0;
    g.broadcast(id, sizeof(id));
    for (int i = 0; i < 128 && !bad; ++i) if (id[i] != (unsigned char)(i * 3 + round)) bad = 3;
    // rank q contributes 100 * q + round records (rank 0 none in round 0)
    std::vector<uint64_t> counts((size_t)g.world());
    for (int q = 0; q < g.world(); ++q) counts[(size_t)q] = (uint64_t)(100 * q + round);
    std::vector<pm_hit> mine(counts[(size_t)r]);
    for (size_t i = 0; i < mine.size(); ++i) { mine[i].end = (int64_t)1e12 + r * 100000 + (int64_t)i; mine[i].pid = (uint32_t)r; mine[i].k = (uint8_t)round; mine[i].aux[0] = mine[i].aux[1] = mine[i].aux[2] = 0; }
    std::vector<pm_hit> got;
    g.gather_host(mine.data(), mine.size(), counts, &got);
    if (r == 0) {
      size_t at = 0;
      for (int q = 0; q < g.world() && !bad; ++q)
        for (size_t i = 0; i < counts[(size_t)q] && !bad; ++i, ++at)
          if (got[at].end != (int64_t)1e12 + q * 100000 + (int64_t)i || got[at].pid != (uint32_t)q || got[at].k != round) bad = 4;
      if (at != got.size()) bad = 5;
    }
  }
  if (bad) fprintf(stderr, "rank %d: check %d failed\n", r, bad);
  if (g.single()) { if (!bad) printf("ok world=1\n"); return bad; }
  if (r == 0 && !bad) printf("ok world=%d device=%d rccl=%d\n", g.world(), g.device(), g.rccl() ? 1 : 0);
  if (r != 0) g.leave(bad);
  return bad;
}

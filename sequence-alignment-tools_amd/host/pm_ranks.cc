// pm_ranks.cc -- see pm_ranks.h.
#include "pm_ranks.h"

#include <dirent.h>
#include <signal.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

namespace pmgpu {

namespace {

[[noreturn]] void die(const char *what) {
  fprintf(stderr, "Fatal error: ranks: %s: %s\n", what, strerror(errno));
  _exit(1);
}

void write_all(int fd, const void *buf, size_t n) {
  const char *p = static_cast<const char *>(buf);
  while (n) {
    const ssize_t w = ::write(fd, p, n);
    if (w < 0) { if (errno == EINTR) continue; die("write"); }
    p += w; n -= (size_t)w;
  }
}

void read_all(int fd, void *buf, size_t n) {
  char *p = static_cast<char *>(buf);
  while (n) {
    const ssize_t r = ::read(fd, p, n);
    if (r < 0) { if (errno == EINTR) continue; die("read"); }
    if (r == 0) { fprintf(stderr, "Fatal error: ranks: a peer rank went away\n"); _exit(1); }
    p += r; n -= (size_t)r;
  }
}

// GPUs this process may use, without initialising the HIP runtime (the launcher forks afterwards):
// entries of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set, else the KFD topology nodes that
// have SIMDs (CPUs are nodes too, with simd_count 0).
int visible_gpus() {
  for (const char *name : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"}) {
    const char *v = getenv(name);
    if (v && *v) {
      int c = 1;
      for (const char *p = v; *p; ++p) c += *p == ',';
      return c;
    }
  }
  int count = 0;
  const std::string base = "/sys/class/kfd/kfd/topology/nodes";
  if (DIR *d = opendir(base.c_str())) {
    while (dirent *e = readdir(d)) {
      if (e->d_name[0] == '.') continue;
      std::ifstream f(base + "/" + e->d_name + "/properties");
      std::string key;
      long long val;
      while (f >> key >> val)
        if (key == "simd_count" && val > 0) { ++count; break; }
    }
    closedir(d);
  }
  return count;
}

}  // namespace

int take_ranks_option(int *argc, char **argv) {
  int ranks = 0;
  int w = 1;
  for (int i = 1; i < *argc; ++i) {
    if (!strcmp(argv[i], "--ranks") && i + 1 < *argc) { ranks = atoi(argv[++i]); continue; }
    if (!strncmp(argv[i], "--ranks=", 8)) { ranks = atoi(argv[i] + 8); continue; }
    argv[w++] = argv[i];
  }
  argv[w] = nullptr;
  *argc = w;
  if (ranks <= 0) { const char *e = getenv("PM_RANKS"); if (e) ranks = atoi(e); }
  return ranks > 0 ? ranks : 1;
}

RankGroup RankGroup::launch(int world) {
  RankGroup g;
  if (world <= 1) return g;
  if (world > 64) { fprintf(stderr, "Fatal error: ranks: at most 64 ranks\n"); exit(1); }
  g.world_ = world;
  // pipes r -> 0 and 0 -> r for every r > 0 (index 0 unused)
  std::vector<int> up_r(world, -1), up_w(world, -1), down_r(world, -1), down_w(world, -1);
  for (int r = 1; r < world; ++r) {
    int a[2], b[2];
    if (pipe(a) || pipe(b)) die("pipe");
    up_r[r] = a[0]; up_w[r] = a[1]; down_r[r] = b[0]; down_w[r] = b[1];
  }
  const int gpus = visible_gpus();
  const char *tr = getenv("PM_RANKS_TRANSPORT");              // "rccl" | "host" | unset = RCCL when every rank gets a GPU
  const bool want_rccl = tr ? !strcmp(tr, "rccl") : gpus >= world;
  fflush(stdout); fflush(stderr);
  std::vector<pid_t> kids(world, 0);
  for (int r = 0; r < world; ++r) {
    const pid_t pid = fork();
    if (pid < 0) die("fork");
    if (pid == 0) {
      g.rank_ = r;
      g.rccl_ = want_rccl;
      g.device_ = gpus > 0 ? r % gpus : 0;
      g.up_.assign(world, -1); g.down_.assign(world, -1);
      for (int q = 1; q < world; ++q) {
        if (r == 0) { g.up_[q] = up_r[q]; g.down_[q] = down_w[q]; close(up_w[q]); close(down_r[q]); }
        else if (q == r) { g.up_[q] = up_w[q]; g.down_[q] = down_r[q]; close(up_r[q]); close(down_w[q]); }
        else { close(up_r[q]); close(up_w[q]); close(down_r[q]); close(down_w[q]); }
      }
      return g;
    }
    kids[r] = pid;
  }
  for (int q = 1; q < world; ++q) { close(up_r[q]); close(up_w[q]); close(down_r[q]); close(down_w[q]); }
  // Wait for whichever rank ends first.  A rank that fails (fatal() after the count exchange, a
  // transport error) leaves its peers blocked in a receive that will never complete -- pipe EOF only
  // covers the host transport -- so the first non-zero status ends the others (SIGTERM to the exact
  // pids forked above) and becomes the launcher's status.
  int status = 0, left = world;
  while (left > 0) {
    int st = 0;
    const pid_t done = waitpid(-1, &st, 0);
    if (done < 0) { if (errno == EINTR) continue; break; }
    int r = -1;
    for (int q = 0; q < world; ++q) if (kids[q] == done) r = q;
    if (r < 0) continue;
    kids[r] = 0; --left;
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (code && !status) {
      status = code;
      for (int q = 0; q < world; ++q) if (kids[q] > 0) kill(kids[q], SIGTERM);
    }
  }
  fflush(stdout);
  _exit(status);
}

void RankGroup::all_gather(uint64_t mine, std::vector<uint64_t> *all) {
  all->assign((size_t)world_, 0);
  (*all)[(size_t)rank_] = mine;
  if (world_ <= 1) return;
  if (rank_ == 0) {
    for (int r = 1; r < world_; ++r) read_all(up_[r], &(*all)[(size_t)r], sizeof(uint64_t));
    for (int r = 1; r < world_; ++r) write_all(down_[r], all->data(), sizeof(uint64_t) * (size_t)world_);
  } else {
    write_all(up_[rank_], &mine, sizeof(mine));
    read_all(down_[rank_], all->data(), sizeof(uint64_t) * (size_t)world_);
  }
}

void RankGroup::broadcast(void *buf, size_t bytes) {
  if (world_ <= 1) return;
  if (rank_ == 0) for (int r = 1; r < world_; ++r) write_all(down_[r], buf, bytes);
  else read_all(down_[rank_], buf, bytes);
}

void RankGroup::gather_host(const pm_hit *mine, size_t n, const std::vector<uint64_t> &counts, std::vector<pm_hit> *all) {
  if (rank_ != 0) { if (n) write_all(up_[rank_], mine, n * sizeof(pm_hit)); return; }
  size_t total = 0;
  for (uint64_t c : counts) total += (size_t)c;
  all->resize(total);
  if (n) memcpy(all->data(), mine, n * sizeof(pm_hit));
  size_t at = n;
  for (int r = 1; r < world_; ++r) {
    if (counts[(size_t)r]) read_all(up_[r], all->data() + at, (size_t)counts[(size_t)r] * sizeof(pm_hit));
    at += (size_t)counts[(size_t)r];
  }
}

void RankGroup::leave(int status) {
  fflush(stdout); fflush(stderr);
  _exit(status);
}

}  // namespace pmgpu

// pm_scan_cli.cc -- minimal driver of GpuPatternMatch with primer_match's scan loop
// (primer_match.cc:1101-1118): prints one line "<end> <id> <errors>" per engine hit.
//   pm_scan_cli [-N 16|17] [-k edits | -K mismatches] [-r] [-n] [-m minka] [-c chunk] -i <db> -P <patterns>
//   -n: <db>.sqn + <db>.tbl (normalized stream), otherwise <db> is a raw byte stream.
//   -B: hide the contiguous buffer (like the reference's -B / BufferedFileChars, char_io.h:172):
//       the engine then drains the stream through getnch() once.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <unistd.h>

#include "gpu_pattern_match.h"

// a producer without c_str(): what BufferedFileChars is to the reference engines
class StreamOnlyChars : public pmgpu::BufferChars {
 public:
  using pmgpu::BufferChars::BufferChars;
  bool has_filename() const override { return false; }
  const char *c_str() const override { return nullptr; }
};

static std::string revcomp(const std::string &s) {           // A,C,G,T only (util.cc:374)
  std::string r(s.rbegin(), s.rend());
  for (char &c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
  return r;
}

static std::vector<unsigned char> slurp(const std::string &path) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) { perror(path.c_str()); exit(1); }
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
  int kernel = PM_KERNEL_AUTO, k = 0, minka = 1000;
  long chunk = 0;
  bool indels = true, rc = false, norm = false, buffered = false;
  std::string db, patfile;
  int c;
  while ((c = getopt(argc, argv, "N:k:K:rnBm:c:i:P:")) != -1) {
    switch (c) {
      case 'N': kernel = atoi(optarg); break;
      case 'k': k = atoi(optarg); indels = true; break;
      case 'K': k = atoi(optarg); indels = false; break;
      case 'r': rc = true; break;
      case 'n': norm = true; break;
      case 'B': buffered = true; break;
      case 'm': minka = atoi(optarg); break;
      case 'c': chunk = atol(optarg); break;
      case 'i': db = optarg; break;
      case 'P': patfile = optarg; break;
      default: return 2;
    }
  }
  if (db.empty() || patfile.empty()) { fprintf(stderr, "need -i and -P\n"); return 2; }
  std::vector<std::string> pats;
  { std::ifstream f(patfile.c_str()); std::string p; while (f >> p) pats.push_back(p); }
  std::string table;
  std::vector<unsigned char> bytes;
  if (norm) { bytes = slurp(db + ".sqn"); auto t = slurp(db + ".tbl"); table.assign(t.begin(), t.end()); }
  else bytes = slurp(db);
  pmgpu::BufferChars mapped(bytes, table);
  StreamOnlyChars streamed(bytes, table);
  pmgpu::CharacterProducer &ff = buffered ? static_cast<pmgpu::CharacterProducer &>(streamed) : mapped;
  pmgpu::GpuPatternMatch kt(kernel, (unsigned)k, '\n', false, false, indels, false);
  if (chunk > 0) kt.chunk_bytes(chunk);
  const size_t n = pats.size();
  for (size_t i = 0; i < n; ++i) kt.add_pattern(pats[i], i + 1, 0, 0);
  if (rc) for (size_t i = 0; i < n; ++i) kt.add_pattern(revcomp(pats[i]), n + i + 1, 0, 0);
  kt.init(ff);
  pmgpu::pattern_hit_vector l;
  bool more;
  while ((more = kt.find_patterns(ff, l, (unsigned long)minka)) || !l.empty()) {
    const int64_t oldpos = ff.pos();
    for (const pmgpu::pattern_hit &h : l) printf("%lld %lu %d\n", (long long)h.key, h.id, (int)h.value);
    l.clear();
    ff.pos(oldpos);
  }
  fprintf(stderr, "semantics=%d kernel=%d\n", kt.selected_semantics(), kt.selected_kernel());
  return 0;
}

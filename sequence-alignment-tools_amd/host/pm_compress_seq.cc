// pm_compress_seq -- writes the database files the scan path reads, byte for byte what the
// reference's compress_seq writes (reference compress_seq.cc:306-1007):
//   <db>.seq   sequence characters, EOS character in front of, between and after the entries
//   <db>.hdr   FASTA headers, one per line
//   <db>.idb   binary index: u64 count, then (int64 position in .seq, int64 offset in .hdr) pairs
//   <db>.tbl   with -n true: characters seen, A,C,G,T first (-D true), then by code
//   <db>.sqn   with -n true: .seq recoded through .tbl; .seq is then removed (-C true)
//
// Options (the subset this build supports; same letters and value syntax as the reference):
//   -i <fasta>  -e [true|false]  -S [true|false]  -E <int>  -u [true|false]  -n [true|false]
//   -D [true|false]  -C [true|false]  -F [true|false] (accepted; files are always rebuilt)
// Not built: -z (4-bit .sqz), -t (suffix tree), -3, -I false (text index), -T, -R, -G, -c, .gz input.
//
// Reference behaviour kept on purpose: the position after the final EOS is counted twice
// (compress_seq.cc:594-609), so the last index key is one too large; characters outside 33..126
// inside sequence lines are dropped (:560-562); CR LF line ends are accepted.
#include <unistd.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "seq_io.h"

namespace {

bool any_of(const char *s, const char *const *v) { for (; **v; ++v) if (!strcmp(*v, s)) return true; return false; }
const char *const kTrue[] = {"true", "True", "TRUE", "T", "t", "1", "Yes", "YES", "yes", ""};      // util.cc:59-64
const char *const kFalse[] = {"false", "False", "FALSE", "F", "f", "0", "No", "NO", "no", ""};
bool is_true(const char *s) { return any_of(s, kTrue); }
bool is_false(const char *s) { return any_of(s, kFalse); }

[[noreturn]] void usage(const char *msg) {
  if (msg && *msg) fprintf(stderr, "%s\n\n", msg);
  fprintf(stderr, "Usage: pm_compress_seq -i <fasta> [-e bool] [-S bool] [-E int] [-u bool] [-n bool] [-z bool] [-D bool] [-C bool] [-F bool]\n");
  exit(1);
}

bool flag(const char *opt, const char *arg) {
  if (is_true(arg)) return true;
  if (is_false(arg)) return false;
  usage((std::string("Invalid value for -") + opt + " option.").c_str());
}

void write_file(const std::string &path, const std::vector<unsigned char> &data) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f || (data.size() && fwrite(data.data(), 1, data.size(), f) != data.size())) {
    fprintf(stderr, "Can't write %s\n", path.c_str());
    exit(1);
  }
  fclose(f);
}

}  // namespace

int main(int argc, char **argv) {
  std::string database;
  bool eos = true, init_eos = true, uc = true, normalize = false, compress = false, dnaopt = true, cleanup = true;
  char eos_char = '\n';
  int c;
  while ((c = getopt(argc, argv, "i:e:S:u:D:E:n:z:F:C:h")) != -1) switch (c) {
      case 'i': database = optarg; break;
      case 'e': eos = flag("e", optarg); break;
      case 'S': init_eos = flag("S", optarg); if (init_eos) eos = true; break;     // compress_seq.cc:152-156
      case 'E': { int ch = 0; sscanf(optarg, "%i", &ch); eos_char = (char)ch; } break;
      case 'u': uc = flag("u", optarg); break;
      case 'D': dnaopt = flag("D", optarg); break;
      case 'n': normalize = flag("n", optarg); break;
      case 'z': compress = flag("z", optarg); break;
      case 'F': (void)flag("F", optarg); break;
      case 'C': cleanup = flag("C", optarg); break;
      default: usage(nullptr);
    }
  if (database.empty()) usage("Arguement -i missing from commandline.");
  std::vector<unsigned char> in;
  if (!pmgpu::read_file(database, &in)) { fprintf(stderr, "Can't open %s\n", database.c_str()); return 1; }
  if (in.empty()) return 0;                              // a file of no bytes: the reference writes nothing (its mapping has no characters to hand out) and exits 0

  std::vector<unsigned char> seq, hdr;
  std::vector<int64_t> idx;                            // (seqpos, headerpos) pairs
  std::vector<bool> obs(256, false);
  if (eos) obs[(unsigned char)eos_char] = true;
  seq.reserve(in.size());
  int64_t seqpos = 0, headerpos = 0;
  if (init_eos) { seq.push_back((unsigned char)eos_char); ++seqpos; }
  idx.push_back(seqpos); idx.push_back(headerpos);
  bool inseq = false, inheader = false, startofline = true;
  const size_t n = in.size();
  for (size_t p = 0; p < n;) {                          // compress_seq.cc:468-576
    char ch = (char)in[p++];
    if (startofline && ch == '>') {
      if (inseq) {
        if (eos) { seq.push_back((unsigned char)eos_char); ++seqpos; }
        idx.push_back(seqpos); idx.push_back(headerpos);
      }
      inheader = true; inseq = false; startofline = false;
      continue;
    }
    if (inheader) {
      if (ch == '\n' || ch == '\r') {
        if (ch == '\r' && p < n) ch = (char)in[p++];
        hdr.push_back((unsigned char)ch); ++headerpos;
        inheader = false; inseq = true; startofline = true;
      } else {
        hdr.push_back((unsigned char)ch); ++headerpos;
        startofline = false;
      }
      continue;
    }
    if (inseq) {
      if (ch == '\n' || ch == '\r') {
        if (ch == '\r' && p < n) ++p;
        startofline = true;
      } else if ((int)ch < 33 || (int)ch > 126) {
        startofline = false;
      } else {
        if (uc) ch = (char)toupper((unsigned char)ch);
        seq.push_back((unsigned char)ch); ++seqpos;
        if (normalize || compress) obs[(unsigned char)ch] = true;      // compress_seq.cc:565
        startofline = false;
      }
    }
  }
  if (inheader) {                                       // file ends inside a header line (:577-592)
    hdr.push_back('\n'); ++headerpos;
    idx.push_back(seqpos); idx.push_back(headerpos);
  } else if (inseq) {
    if (eos) { seq.push_back((unsigned char)eos_char); ++seqpos; ++seqpos; }   // counted twice: :594-609
    idx.push_back(seqpos); idx.push_back(headerpos);
  }

  write_file(database + ".seq", seq);
  write_file(database + ".hdr", hdr);
  {
    std::vector<unsigned char> idb(8 + idx.size() * 8);
    const uint64_t cnt = idx.size() / 2;
    memcpy(idb.data(), &cnt, 8);
    memcpy(idb.data() + 8, idx.data(), idx.size() * 8);
    write_file(database + ".idb", idb);
  }
  if (!normalize && !compress) return 0;

  int order[256];
  for (int i = 0; i < 256; ++i) order[i] = i;
  if (dnaopt) {                                         // :699-704
    order[0] = 'A'; order['A'] = 0; order[1] = 'C'; order['C'] = 1;
    order[2] = 'G'; order['G'] = 2; order[3] = 'T'; order['T'] = 3;
  }
  std::vector<unsigned char> tbl;
  unsigned char inv[256];
  memset(inv, 255, sizeof(inv));
  for (int i = 0; i < 256; ++i)
    if (obs[order[i]]) { inv[order[i]] = (unsigned char)tbl.size(); tbl.push_back((unsigned char)order[i]); }
  if (normalize) write_file(database + ".tbl", tbl);
  if (compress) {
    // <db>.tbz (the same table) + <db>.sqz: the codes at ceil(log2(table size)) bits each, most significant bit
    // first, the last buffer of lcm(bits, 8) bytes filled up with end-of-sequence codes (compress_seq.cc:741-905)
    write_file(database + ".tbz", tbl);
    unsigned bits = 1;
    while ((1u << bits) < tbl.size()) ++bits;
    size_t lcm = bits;
    while (lcm % 8) lcm += bits;
    const size_t bufbits = lcm * 8;                       // bufsize = lcm(bits, 8) / 8 * 8 bytes
    if (inv[(unsigned char)eos_char] == 255 && !seq.empty() && (seq.size() * bits) % bufbits) {
      fprintf(stderr, "The end-of-sequence character is not in the table: cannot fill up %s.sqz\n", database.c_str());
      return 1;
    }
    size_t nchars = seq.size();
    while ((nchars * bits) % bufbits) ++nchars;
    std::vector<unsigned char> sqz(nchars * bits / 8, 0);
    size_t bitat = 0;
    for (size_t i = 0; i < nchars; ++i) {
      const unsigned code = i < seq.size() ? inv[seq[i]] : inv[(unsigned char)eos_char];
      for (int b = (int)bits - 1; b >= 0; --b, ++bitat)
        if ((code >> b) & 1u) sqz[bitat >> 3] |= (unsigned char)(0x80u >> (bitat & 7));
    }
    write_file(database + ".sqz", sqz);
  }
  if (normalize) {
    std::vector<unsigned char> sqn(seq.size());
    for (size_t i = 0; i < seq.size(); ++i) sqn[i] = inv[seq[i]];
    write_file(database + ".sqn", sqn);
  }
  if (cleanup) unlink((database + ".seq").c_str());
  return 0;
}

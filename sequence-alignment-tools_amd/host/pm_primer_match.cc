// pm_primer_match -- the reference's primer_match command line on the MI355X engine
// (reference primer_match.cc:43-1335).
//
// Same options, same output text: patterns are read and expanded as primer_match.cc:866-1084
// does (ids 1..n forward, n+1..2n reverse complement; exact-zone constraints from -s/-e/-5/-3),
// the scan loop is primer_match.cc:1101-1268 with GpuPatternMatch in place of the engine
// pick_pattern_index returns, every hit is re-aligned (pm_align_hits_text = exact_alignment /
// editdist_alignment) and printed through the -A / -C mini-languages (primer_match.cc:355-843).
//
// Not built (refused with a message): -T (translation), DNA-mutation scoring (-k .N),
// the PRIMER3TM escapes %m %G and the peptide-mass escape %M.
#include <unistd.h>
#include <thread>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <algorithm>
#include "seq_io.h"

using namespace pmgpu;

namespace {

struct Options {
  bool both_strands = false, primers_from_file = false, primers_from_fasta = false, primers_from_sts = false;
  std::string primer_arg, db_path;
  bool to_upper = false;
  int zone3 = 0, zone5 = 0, zone_start = 0, zone_end = 0, seed_len = 0;
  char eos = '\n';
  int max_edits = 0, count_cap = 0;
  std::string hit_format = ">%h\\n %T %s %e %d\\n %A\\n %Q %i%R\\n";
  bool print_hits = true;
  std::string tally_format = "%i %r %q %c%+ ( %C )\\n";
  bool print_tallies = false;
  std::string out_path;
  unsigned long progress_every = 1000;
  bool chatty = false, map_db = true;
  int engine_choice = 0, db_variant = 0;
  bool use_db_index = true, iupac = false, text_n_matches = false, with_indels = true, merge_tallies = false, dna_scoring = false, translated = false;
};

[[noreturn]] void usage(const char *msg = nullptr) {
  if (msg && *msg) fprintf(stderr, "%s\n\n", msg);
  fprintf(stderr,
          "Usage: pm_primer_match [options]\n\n"
          "  -i <sequence-database>  database prepared by (pm_)compress_seq. Required.\n"
          "  --ranks <n>             one process per GPU, the database sharded by position (default $PM_RANKS or 1)\n"
          "  -p <sequences> | -P <file> | -F <fasta-file> | -S <unists-file>   primers (\"-\" = stdin)\n"
          "  -o <output-file>        append to file instead of standard out\n"
          "  -k <n> / -K <n>         edits / substitutions permitted (default 0)\n"
          "  -r                      match reverse complements too\n"
          "  -s -e -5 -3 <n|~n>      exact (inexact with ~) zones, as primer_match\n"
          "  -u  -w  -W  -E <int>  -c  -a  -M <max>  -A <format>  -C <format>  -R <int>\n"
          "  -N <int>                engine: 0 auto, 16 bit-parallel kernels, 17 seed kernels,\n"
          "                          1..14 = reproduce that reference engine's hit set\n"
          "  -D (0|1|2|3|4)  -I  -B  -v  -h\n");
  exit(1);
}

int tilde(const char *a) { return a[0] == '~' ? -atoi(a + 1) : atoi(a); }

Options parse(int argc, char **argv) {
  Options o;
  int c;
  while ((c = getopt(argc, argv, "p:i:o:P:F:S:M:k:K:s:e:3:5:x:E:hrucavA:C:R:BN:D:IwWT")) != -1) switch (c) {
      case 'p': o.primer_arg = optarg; o.primers_from_file = false; break;
      case 'P': o.primer_arg = optarg; o.primers_from_file = true; break;
      case 'F': o.primer_arg = optarg; o.primers_from_fasta = true; break;
      case 'S': o.primer_arg = optarg; o.primers_from_sts = true; o.both_strands = true; break;
      case 'i': o.db_path = optarg; break;
      case 'o': o.out_path = optarg; break;
      case '3': o.zone3 = tilde(optarg); break;
      case '5': o.zone5 = tilde(optarg); break;
      case 's': o.zone_start = tilde(optarg); break;
      case 'e': o.zone_end = tilde(optarg); break;
      case 'k':
      case 'K':
        if (optarg[0] == '.') { o.max_edits = atoi(optarg + 1); o.dna_scoring = true; } else o.max_edits = atoi(optarg);
        o.with_indels = c == 'k';
        break;
      case 'r': o.both_strands = true; break;
      case 'c': o.print_tallies = true; o.print_hits = false; break;
      case 'M': o.count_cap = atoi(optarg); break;
      case 'x': o.seed_len = atoi(optarg); break;
      case 'A': if (strlen(optarg) > 0) o.hit_format = optarg; o.print_hits = true; break;
      case 'C': if (strlen(optarg) > 0) o.tally_format = optarg; o.print_tallies = true; break;
      case 'u': o.to_upper = true; break;
      case 'a': o.merge_tallies = true; break;
      case 'T': o.translated = true; break;
      case 'w': o.iupac = true; o.text_n_matches = false; break;
      case 'W': o.iupac = true; o.text_n_matches = true; break;
      case 'R': o.progress_every = (unsigned long)atoi(optarg); break;
      case 'N': o.engine_choice = atoi(optarg); break;
      case 'D': o.db_variant = atoi(optarg); break;
      case 'E': { int ec; if (!sscanf(optarg, "%i", &ec)) usage("Invalid end-of-sequence specification.\n"); o.eos = (char)ec; } break;
      case 'v': o.chatty = true; break;
      case 'I': o.use_db_index = false; break;
      case 'B': o.map_db = false; break;
      default: usage();
    }
  if ((o.primer_arg.empty() || o.db_path.empty()) && !o.chatty) usage("No primers and/or no sequence database supplied.");
  if (o.max_edits < 0) usage("Number of mismatches (-k) must be >= 0.");
  if (o.count_cap > 0 && !o.print_tallies) usage("Can''t use maxcount (-M) without counts (-c or -C).");
  if (o.merge_tallies && !o.print_tallies) usage("Can''t use aggregate (-a) without counts (-c or -C).");
  if (o.db_variant < 0 || o.db_variant > 4) usage("Invalid integer for fasta database indexing (-D).");
  if (o.dna_scoring) usage("DNA mutation scoring (-k .N) is not available on the GPU engine.");
  if (o.translated) usage("Translation (-T) is not available on the GPU engine.");
  return o;
}

void escape(std::ostream &os, const std::string &f, size_t &pos) {      // the backslash half of both mini-languages
  ++pos;
  if (pos < f.size()) {
    switch (f[pos]) {
      case 'n': os << '\n'; break;
      case 't': os << '\t'; break;
      case '\\': os << '\\'; break;
      default: os << f[pos];
    }
  } else {
    os << '\\';
  }
}

void sts_size(std::ostream &os, const StsEntry &sts) {
  if (sts.sizeub != sts.sizelb) os << sts.sizelb << "-" << sts.sizeub; else os << sts.sizelb;
}

struct AlignFields {                                   // the arguments of alignformat() (primer_match.cc:355-378)
  long long s, e, five, three, S, E;
  unsigned long i;
  unsigned int d;
  std::string p, P, q, Q, r, R, t, T, A, h, H;
  unsigned long f;
  const StsEntry *sts;
};

// "%=": text / alignment / pattern rows cut into slices of 50 columns.  After each text slice come
// the offsets of its first and one-past-last text character inside the matched text (a 'v' column
// is a gap in the text and consumes none) and the slice's number of edit columns (every column
// that is neither '|' nor '+'); slices are separated by an empty line.
void wrapped_layout(std::ostream &os, const AlignFields &a) {
  constexpr size_t ROW = 50;
  const size_t total = a.T.size();
  size_t text_at = 0;
  for (size_t from = 0; from < total; from += ROW) {
    const size_t cols = std::min(ROW, total - from);
    const std::string ops = a.A.substr(from, cols);
    const size_t gaps = (size_t)std::count(ops.begin(), ops.end(), 'v');
    const size_t same = (size_t)std::count_if(ops.begin(), ops.end(), [](char c) { return c == '|' || c == '+'; });
    const size_t text_to = text_at + cols - gaps;
    os << ' ' << a.T.substr(from, ROW) << ' ' << text_at << ' ' << text_to << ' ' << cols - same << '\n'
       << ' ' << ops << '\n'
       << ' ' << a.Q.substr(from, ROW) << ' ' << a.i << a.R << '\n';
    if (from + ROW < total) os << std::endl;
    text_at = text_to;
  }
}

void alignformat(std::ostream &os, const std::string &fmt, const AlignFields &a) {
  unsigned ins = 0, del = 0, sub = 0, wcm = 0, mat = 0;
  bool sc = false;
  auto tally = [&]() {
    if (sc) return;
    for (char ch : a.A) switch (ch) { case '|': ++mat; break; case '^': ++del; break; case 'v': ++ins; break; case '*': ++sub; break; case '+': ++wcm; break; }
    sc = true;
  };
  for (size_t pos = 0; pos < fmt.size(); ++pos) {
    if (fmt[pos] == '%') {
      ++pos;
      if (pos >= fmt.size()) { os << "%"; continue; }
      switch (fmt[pos]) {
        case 's': os << a.s; break;
        case 'e': os << a.e; break;
        case 'l': os << a.e - a.s; break;
        case '5': os << a.five; break;
        case '3': os << a.three; break;
        case 'S': os << a.S; break;
        case 'E': os << a.E; break;
        case 'i': os << a.i; break;
        case 'd': os << a.d; break;
        case 'D': os << (unsigned long long)a.p.length() - (unsigned long long)(a.s - a.e); break;
        case 'p': os << a.p; break;
        case 'P': os << a.P; break;
        case 'q': os << a.q; break;
        case 'Q': os << a.Q; break;
        case 'r': os << a.r; break;
        case 'R': os << a.R; break;
        case 't': os << a.t; break;
        case 'T': os << a.T; break;
        case 'U': os << (a.r == "R" ? reverse_comp(a.t) : a.t); break;
        case 'A': os << a.A; break;
        case 'h': os << a.h; break;
        case 'H': os << a.H; break;
        case 'f': os << a.f; break;
        case 'I': os << a.sts->id; break;
        case 'L': sts_size(os, *a.sts); break;
        case 'a': os << a.sts->acc; break;
        case 'O': os << a.sts->species; break;
        case '&': os << a.sts->altacc; break;
        case 'X': os << a.sts->chrom; break;
        case 'F': os << -1; break;                    // frame: only set with -T
        case 'n': break;                              // translated bases: only with -T
        case '%': os << "%"; break;
        case '|': tally(); os << mat; break;
        case '^': tally(); os << del; break;
        case 'v': tally(); os << ins; break;
        case '*': tally(); os << sub; break;
        case '+': tally(); os << wcm; break;
        case '=': wrapped_layout(os, a); break;        // the default layout, 50 alignment columns per row (:593-628)
        default: os << fmt[pos];
      }
    } else if (fmt[pos] == '\\') {
      escape(os, fmt, pos);
    } else {
      os << fmt[pos];
    }
  }
}

void countformat(std::ostream &os, const std::string &fmt, unsigned long i, const std::string &p, const std::string &P,
                 const std::string &q, const std::string &r, const std::string &R, unsigned long c,
                 const std::vector<unsigned long> &C, unsigned k, bool gtmax, const StsEntry &sts) {
  for (size_t pos = 0; pos < fmt.size(); ++pos) {
    if (fmt[pos] == '%') {
      ++pos;
      if (pos >= fmt.size()) { os << "%"; continue; }
      switch (fmt[pos]) {
        case 'i': os << i; break;
        case 'p': os << p; break;
        case 'P': os << P; break;
        case 'q': os << q; break;
        case 'r': os << r; break;
        case 'R': os << R; break;
        case 'c': os << c; break;
        case 'C': for (unsigned j = 0; j < k; ++j) os << C[j] << " "; os << C[k]; break;
        case '+': if (gtmax) os << "+"; break;
        case '%': os << "%"; break;
        case 'I': os << sts.id; break;
        case 'L': sts_size(os, sts); break;
        case 'a': os << sts.acc; break;
        case 'O': os << sts.species; break;
        case '&': os << sts.altacc; break;
        case 'X': os << sts.chrom; break;
        default: os << fmt[pos];
      }
    } else if (fmt[pos] == '\\') {
      escape(os, fmt, pos);
    } else {
      os << fmt[pos];
    }
  }
}

std::string with_gaps(const std::string &src, const std::string &ops, char gap_op) {
  // pattern_alignment::alignment_text ('v' = deletion: gap in the text) and alignment_pattern
  // ('^' = insertion: gap in the pattern), pattern_alignment.h:166-196
  std::string r;
  size_t p = 0;
  for (char op : ops) {
    if (op != gap_op) { r += p < src.size() ? src[p] : ' '; ++p; } else r += "-";
  }
  return r;
}

}  // namespace

int main(int argc, char **argv) {
  const int nranks = take_ranks_option(&argc, argv);                  // --ranks N: one process per GPU, the stream sharded by position
  Options opt = parse(argc, argv);
  // the HIP runtime takes 0.06 - 0.15 s to come up: let it, on a thread of its own, while the primers and the database are read
  if (nranks <= 1) std::thread([]() { (void)pm_prepare_device(getenv("PM_GPU_DEVICE") ? atoi(getenv("PM_GPU_DEVICE")) : 0); }).detach();
  Phases ph; ph.on = opt.chatty;
  std::ofstream fout;
  if (!opt.out_path.empty()) fout.open(opt.out_path.c_str(), std::ios::out | std::ios::app | std::ios::ate);
  std::ostream &out = opt.out_path.empty() ? std::cout : fout;

  // ---- primers (primer_match.cc:866-934) ----------------------------------------------------
  std::vector<std::string> patterns, patdeflines;
  std::vector<StsEntry> sts;
  {
    std::ifstream file;
    std::istream *ifs = &std::cin;
    if ((opt.primers_from_file || opt.primers_from_fasta || opt.primers_from_sts) && opt.primer_arg != "-") {
      file.open(opt.primer_arg.c_str());
      ifs = &file;
    }
    if (opt.primers_from_file) {
      std::string p;
      while ((*ifs) >> p) patterns.push_back(p);
    } else if (opt.primers_from_fasta) {
      FastaEntry f;
      while (read_fasta_entry(*ifs, &f)) {
        if (f.sequence.empty()) break;
        patterns.push_back(f.sequence);
        patdeflines.push_back(f.defline);
      }
    } else if (opt.primers_from_sts) {
      StsEntry s;
      for (;;) {
        read_sts_entry(*ifs, &s);
        if (!(*ifs)) break;
        if (s.forward_primer.empty()) break;
        patterns.push_back(s.forward_primer);
        patterns.push_back(s.reverse_primer);
        sts.push_back(s);
      }
    } else {
      std::istringstream sis(opt.primer_arg);
      std::string p;
      while (sis >> p) patterns.push_back(p);
    }
  }
  if (patterns.empty()) return 0;
  ph.mark("Read primers");
  // the ranks are forked only now: the primers may come from stdin ("-"), which the rank processes would otherwise
  // share -- one would drain it, or each would read a different slice and build different tables.  Nothing above touches a GPU.
  RankGroup ranks = RankGroup::launch(nranks);                      // returns in every rank process
  if (opt.to_upper) for (std::string &p : patterns) uppercase(p);

  const unsigned long n = patterns.size();
  const unsigned long N1 = (opt.both_strands ? 2 : 1) * n;
  std::vector<std::string> patarray(N1 + 1);
  std::vector<std::pair<int, int>> patconst(N1 + 1);
  std::vector<std::string> patdefarray(opt.primers_from_fasta ? n + 1 : 0);
  std::vector<StsEntry> stsarray(opt.primers_from_sts ? n / 2 + 1 : 0);
  const unsigned K1 = (unsigned)opt.max_edits + 1;
  std::vector<unsigned long> patcount(opt.print_tallies ? N1 * K1 : 0, 0);
  std::vector<bool> maxpatcount(opt.print_tallies && opt.count_cap > 0 ? N1 + 1 : 0, false);
  auto cidx = [&](unsigned long i, unsigned k) { return (i - 1) * K1 + k; };
  for (unsigned long i = 1; i <= n; ++i) {              // primer_match.cc:966-1076
    const std::string &pat = patterns[i - 1];
    const int L = (int)pat.length();
    patarray[i] = pat;
    if (opt.primers_from_fasta) patdefarray[i] = patdeflines[i - 1];
    if (opt.primers_from_sts && i % 2 == 1) stsarray[(i + 1) / 2] = sts[(i - 1) / 2];
    int &f1 = patconst[i].first, &s1 = patconst[i].second;
    f1 = opt.zone_start > 0 ? opt.zone_start : 0;
    if (opt.zone5 > f1) f1 = opt.zone5;
    if (opt.zone_end < 0 && L + opt.zone_end > f1) f1 = L + opt.zone_end;
    if (opt.zone3 < 0 && L + opt.zone3 > f1) f1 = L + opt.zone3;
    s1 = opt.zone_end > 0 ? opt.zone_end : 0;
    if (opt.zone3 > s1) s1 = opt.zone3;
    if (opt.zone_start < 0 && L + opt.zone_start > s1) s1 = L + opt.zone_start;
    if (opt.zone5 < 0 && L + opt.zone5 > s1) s1 = L + opt.zone5;
    if (opt.both_strands) {
      patarray[i + n] = reverse_comp(pat);
      int &f2 = patconst[i + n].first, &s2 = patconst[i + n].second;
      f2 = opt.zone_start > 0 ? opt.zone_start : 0;
      if (opt.zone3 > f2) f2 = opt.zone3;
      if (opt.zone_end < 0 && L + opt.zone_end > f2) f2 = L + opt.zone_end;
      if (opt.zone5 < 0 && L + opt.zone5 > f2) f2 = L + opt.zone5;
      s2 = opt.zone_end > 0 ? opt.zone_end : 0;
      if (opt.zone5 > s2) s2 = opt.zone5;
      if (opt.zone_start < 0 && L + opt.zone_start > s2) s2 = L + opt.zone_start;
      if (opt.zone3 < 0 && L + opt.zone3 > s2) s2 = L + opt.zone3;
    }
  }

  // ---- database and engine (primer_match.cc:1086-1112) --------------------------------------
  SeqDb db(opt.db_path, opt.db_variant, opt.print_hits && opt.use_db_index, opt.use_db_index, opt.to_upper, opt.eos, opt.map_db);
  ph.mark("Loaded sequence database");
  int kernel = PM_KERNEL_AUTO, semantics = PM_SEM_AUTO;
  if (opt.engine_choice == 16) kernel = PM_KERNEL_BITPAR;
  else if (opt.engine_choice == 17 || opt.engine_choice == 0) kernel = PM_KERNEL_AUTO;
  else semantics = opt.engine_choice;                            // reproduce that reference engine's hit set
  GpuPatternMatch kt(kernel, (unsigned)opt.max_edits, opt.eos, opt.iupac, opt.text_n_matches, opt.with_indels, false, semantics, 0, &ranks);
  size_t maxlen = 0;
  for (unsigned long i = 1; i <= N1; ++i) {
    kt.add_pattern(patarray[i], i, patconst[i].first, patconst[i].second);
    maxlen = std::max(maxlen, patarray[i].size());
  }
  BufferChars &ff = db.chars();
  kt.init(ff);
  if (!ranks.single() && ranks.rank() != 0) {                        // this rank scans its shard, hands its records to rank 0 and is done
    pattern_hit_vector none;
    kt.find_patterns(ff, none, 1);
    ranks.leave(0);
  }
  ph.mark("Primer index built, stream resident on the GPU");
  double t_scan = 0, t_report = 0;
  unsigned long nhits = 0;

  // ---- scan loop (primer_match.cc:1114-1268) ------------------------------------------------
  const size_t stride = maxlen + (size_t)opt.max_edits + 2;
  pattern_hit_vector l;
  std::vector<pm_hit> hv;
  std::vector<pm_alignment> al;
  std::vector<char> opsbuf, textbuf;
  StsEntry null_sts;
  for (;;) {
    const auto ts0 = std::chrono::steady_clock::now();
    const bool more = kt.find_patterns(ff, l, opt.progress_every);
    const auto ts1 = std::chrono::steady_clock::now();
    t_scan += std::chrono::duration<double>(ts1 - ts0).count();
    if (!more && l.empty()) break;
    nhits += l.size();
    const int64_t oldcharspos = ff.pos();
    hv.resize(l.size()); al.resize(l.size());
    if (opt.print_hits) { opsbuf.assign(l.size() * stride, 0); textbuf.assign(l.size() * stride, 0); }
    for (size_t j = 0; j < l.size(); ++j) { hv[j].end = l[j].key; hv[j].pid = (uint32_t)l[j].id; hv[j].k = l[j].value; hv[j].aux[0] = hv[j].aux[1] = hv[j].aux[2] = 0; }
    // counts only (-c): distances are enough, no alignment strings
    if (!l.empty() && (opt.print_hits ? pm_align_hits_text(kt.handle(), hv.data(), hv.size(), al.data(), opsbuf.data(), textbuf.data(), stride)
                                      : pm_align_hits(kt.handle(), hv.data(), hv.size(), al.data())) != PM_OK) {
      fprintf(stderr, "Fatal error: alignment: %s\n", pm_last_error(kt.handle()));
      return 1;
    }
    for (size_t j = 0; j < l.size(); ++j) {
      const unsigned long pid = l[j].id;
      if (!pid || (opt.count_cap > 0 && maxpatcount[pid])) continue;
      const pm_alignment &pa = al[j];
      if (pa.editdist < 0 || pa.editdist > opt.max_edits) {      // "Bogus hit" (primer_match.cc:1249-1263)
        fprintf(stderr, "Bogus hit returned to primer_match main()\n");
        if (opt.print_hits) fprintf(stderr, "Problem sequence is near:\n>%s\n", db.get_header_data(l[j].key).header.c_str());
        else fprintf(stderr, "Approximate absolute sequence position:\n %lld\n", (long long)l[j].key);
        fprintf(stderr, "Problem primer:\n %s\n", patarray[pid].c_str());
        return 1;
      }
      if (opt.print_hits) {
        const long long length = pa.end - pa.start + 1;           // pattern_alignment::length (pattern_alignment.h:96-99)
        const long long p = pa.end;
        const long long spe = db.get_seq_pos(p), sps = spe - length + 1, pe = pa.end, ps = pe - length + 1;
        const bool rc = pid > n;
        const unsigned long ind = pid - (rc ? n : 0);
        const HeaderData &h = db.get_header_data(p);
        const std::string ops(opsbuf.data() + j * stride), mt(textbuf.data() + j * stride);
        AlignFields a;
        a.s = sps; a.e = spe; a.five = rc ? spe : sps; a.three = rc ? sps : spe; a.S = ps; a.E = pe;
        a.i = ind; a.d = (unsigned)pa.editdist; a.p = patarray[ind];
        a.P = opt.primers_from_fasta ? patdefarray[ind] : std::string();
        a.q = patarray[pid]; a.Q = with_gaps(patarray[pid], ops, '^');
        a.r = rc ? "R" : "F"; a.R = rc ? " REVCOMP" : "";
        a.t = mt; a.T = with_gaps(mt, ops, 'v'); a.A = ops;
        a.h = h.header; a.H = h.short_header; a.f = h.index;
        a.sts = opt.primers_from_sts ? &stsarray[(ind + 1) / 2] : &null_sts;
        alignformat(out, opt.hit_format, a);
      }
      if (opt.print_tallies) {
        patcount[cidx(pid, (unsigned)pa.editdist)]++;
        if (opt.count_cap > 0) {
          unsigned long count = 0;
          for (unsigned k = 0; k < K1; ++k) count += patcount[cidx(pid, k)];
          if (count >= (unsigned)opt.count_cap) maxpatcount[pid] = true;
        }
      }
    }
    l.clear();
    ff.pos(oldcharspos);
    t_report += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count();
  }
  if (opt.chatty) fprintf(stderr, "scan (find_patterns) %.3f s, re-align + report %.3f s, %lu hits\n", t_scan, t_report, nhits);
  ph.mark("Scanned sequence database");

  // ---- counts (primer_match.cc:1270-1328) ---------------------------------------------------
  if (opt.print_tallies) {
    std::vector<unsigned long> counts(K1);
    for (unsigned long i = 1; i <= n; ++i) {
      unsigned long total = 0;
      for (unsigned k = 0; k < K1; ++k) { counts[k] = patcount[cidx(i, k)]; total += counts[k]; }
      bool gtmax = opt.count_cap > 0 ? (bool)maxpatcount[i] : false;
      const std::string patdef = opt.primers_from_fasta ? patdefarray[i] : std::string();
      const StsEntry &stsref = opt.primers_from_sts ? stsarray[(i + 1) / 2] : null_sts;
      if (!opt.merge_tallies) countformat(out, opt.tally_format, i, patarray[i], patdef, patarray[i], "F", "", total, counts, (unsigned)opt.max_edits, gtmax, stsref);
      if (opt.both_strands) {
        if (!opt.merge_tallies) { total = 0; std::fill(counts.begin(), counts.end(), 0ul); gtmax = false; }
        for (unsigned k = 0; k < K1; ++k) { counts[k] += patcount[cidx(i + n, k)]; total += patcount[cidx(i + n, k)]; }
        if (opt.count_cap > 0) gtmax = gtmax || maxpatcount[i + n];
        if (!opt.merge_tallies) countformat(out, opt.tally_format, i, patarray[i], patdef, patarray[i + n], "R", " REVCOMP", total, counts, (unsigned)opt.max_edits, gtmax, stsref);
      }
      if (opt.merge_tallies) countformat(out, opt.tally_format, i, patarray[i], patdef, "", "", "", total, counts, (unsigned)opt.max_edits, gtmax, stsref);
    }
  }
  out.flush();
  // The reference leaks its engine at exit (primer_match.cc: no delete of kt); tearing down the HIP runtime, the pinned
  // buffers and a 3 GB mapping took 0.3 s of a 0.9 s run.  Everything is flushed: leave without the destructors.
  fflush(stdout);
  fflush(stderr);
  if (!opt.out_path.empty()) fout.close();
  _exit(0);
}

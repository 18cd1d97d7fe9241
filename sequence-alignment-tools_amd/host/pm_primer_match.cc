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
// raw FASTA / .sqz databases (-D 1, -D 4; run pm_compress_seq first),
// the PRIMER3TM escapes %m %G and the peptide-mass escape %M.
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "seq_io.h"

using namespace pmgpu;

namespace {

struct Options {
  bool rev_comp = false, pattern_file = false, fasta_pattern_file = false, sts_pattern_file = false;
  std::string patterns, database;
  bool ucdict = false;
  int tplen = 0, fplen = 0, stlen = 0, edlen = 0, seedlen = 0;
  char eos_char = '\n';
  int nmismatch = 0, maxcount = 0;
  std::string alignformat = ">%h\\n %T %s %e %d\\n %A\\n %Q %i%R\\n";
  bool alignments = true;
  std::string countformat = "%i %r %q %c%+ ( %C )\\n";
  bool counts = false;
  std::string outfile;
  unsigned long report_interval = 1000;
  bool verbose = false, memmap = true;
  int node = 0, dbind = 0;
  bool dbindex = true, wc = false, tn = false, indels = true, aggregate = false, dna_mutations = false, translate = false;
};

[[noreturn]] void usage(const char *msg = nullptr) {
  if (msg && *msg) fprintf(stderr, "%s\n\n", msg);
  fprintf(stderr,
          "Usage: pm_primer_match [options]\n\n"
          "  -i <sequence-database>  database prepared by (pm_)compress_seq. Required.\n"
          "  -p <sequences> | -P <file> | -F <fasta-file> | -S <unists-file>   primers (\"-\" = stdin)\n"
          "  -o <output-file>        append to file instead of standard out\n"
          "  -k <n> / -K <n>         edits / substitutions permitted (default 0)\n"
          "  -r                      match reverse complements too\n"
          "  -s -e -5 -3 <n|~n>      exact (inexact with ~) zones, as primer_match\n"
          "  -u  -w  -W  -E <int>  -c  -a  -M <max>  -A <format>  -C <format>  -R <int>\n"
          "  -N <int>                engine: 0 auto, 16 bit-parallel kernels, 17 seed kernels,\n"
          "                          1..14 = reproduce that reference engine's hit set\n"
          "  -D (0|2|3)  -I  -B  -v  -h\n");
  exit(1);
}

int tilde(const char *a) { return a[0] == '~' ? -atoi(a + 1) : atoi(a); }

Options parse(int argc, char **argv) {
  Options o;
  int c;
  while ((c = getopt(argc, argv, "p:i:o:P:F:S:M:k:K:s:e:3:5:x:E:hrucavA:C:R:BN:D:IwWT")) != -1) switch (c) {
      case 'p': o.patterns = optarg; o.pattern_file = false; break;
      case 'P': o.patterns = optarg; o.pattern_file = true; break;
      case 'F': o.patterns = optarg; o.fasta_pattern_file = true; break;
      case 'S': o.patterns = optarg; o.sts_pattern_file = true; o.rev_comp = true; break;
      case 'i': o.database = optarg; break;
      case 'o': o.outfile = optarg; break;
      case '3': o.tplen = tilde(optarg); break;
      case '5': o.fplen = tilde(optarg); break;
      case 's': o.stlen = tilde(optarg); break;
      case 'e': o.edlen = tilde(optarg); break;
      case 'k':
      case 'K':
        if (optarg[0] == '.') { o.nmismatch = atoi(optarg + 1); o.dna_mutations = true; } else o.nmismatch = atoi(optarg);
        o.indels = c == 'k';
        break;
      case 'r': o.rev_comp = true; break;
      case 'c': o.counts = true; o.alignments = false; break;
      case 'M': o.maxcount = atoi(optarg); break;
      case 'x': o.seedlen = atoi(optarg); break;
      case 'A': if (strlen(optarg) > 0) o.alignformat = optarg; o.alignments = true; break;
      case 'C': if (strlen(optarg) > 0) o.countformat = optarg; o.counts = true; break;
      case 'u': o.ucdict = true; break;
      case 'a': o.aggregate = true; break;
      case 'T': o.translate = true; break;
      case 'w': o.wc = true; o.tn = false; break;
      case 'W': o.wc = true; o.tn = true; break;
      case 'R': o.report_interval = (unsigned long)atoi(optarg); break;
      case 'N': o.node = atoi(optarg); break;
      case 'D': o.dbind = atoi(optarg); break;
      case 'E': { int ec; if (!sscanf(optarg, "%i", &ec)) usage("Invalid end-of-sequence specification.\n"); o.eos_char = (char)ec; } break;
      case 'v': o.verbose = true; break;
      case 'I': o.dbindex = false; break;
      case 'B': o.memmap = false; break;
      default: usage();
    }
  if ((o.patterns.empty() || o.database.empty()) && !o.verbose) usage("No primers and/or no sequence database supplied.");
  if (o.nmismatch < 0) usage("Number of mismatches (-k) must be >= 0.");
  if (o.maxcount > 0 && !o.counts) usage("Can''t use maxcount (-M) without counts (-c or -C).");
  if (o.aggregate && !o.counts) usage("Can''t use aggregate (-a) without counts (-c or -C).");
  if (o.dbind < 0 || o.dbind > 4) usage("Invalid integer for fasta database indexing (-D).");
  if (o.dna_mutations) usage("DNA mutation scoring (-k .N) is not available on the GPU engine.");
  if (o.translate) usage("Translation (-T) is not available on the GPU engine.");
  if (o.dbind == 1 || o.dbind == 4) usage("Only indexed (-D 2) and normalized (-D 3) databases are supported; run pm_compress_seq first.");
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
        case '=': {                                   // the default layout wrapped at 50 columns (:593-628)
          const unsigned len0 = (unsigned)a.T.length(), width0 = 50;
          unsigned textchars_start = 0;
          for (unsigned i0 = 0; i0 < len0; i0 += width0) {
            unsigned nchars = width0;
            if (i0 + nchars > len0) nchars = len0 - i0;
            unsigned textchars_end = textchars_start + nchars, editcount0 = nchars;
            for (unsigned j0 = 0; j0 < nchars; ++j0) {
              if (a.A[i0 + j0] == '|' || a.A[i0 + j0] == '+') --editcount0;
              if (a.A[i0 + j0] == 'v') --textchars_end;
            }
            os << " " << a.T.substr(i0, width0) << " " << textchars_start << " " << textchars_end << " " << editcount0 << "\n"
               << " " << a.A.substr(i0, width0) << "\n"
               << " " << a.Q.substr(i0, width0) << " " << a.i << a.R << "\n";
            if (len0 - i0 > width0) os << std::endl;
            textchars_start = textchars_end;
          }
        } break;
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
  Options opt = parse(argc, argv);
  Phases ph; ph.on = opt.verbose;
  std::ofstream fout;
  if (!opt.outfile.empty()) fout.open(opt.outfile.c_str(), std::ios::out | std::ios::app | std::ios::ate);
  std::ostream &out = opt.outfile.empty() ? std::cout : fout;

  // ---- primers (primer_match.cc:866-934) ----------------------------------------------------
  std::vector<std::string> patterns, patdeflines;
  std::vector<StsEntry> sts;
  {
    std::ifstream file;
    std::istream *ifs = &std::cin;
    if ((opt.pattern_file || opt.fasta_pattern_file || opt.sts_pattern_file) && opt.patterns != "-") {
      file.open(opt.patterns.c_str());
      ifs = &file;
    }
    if (opt.pattern_file) {
      std::string p;
      while ((*ifs) >> p) patterns.push_back(p);
    } else if (opt.fasta_pattern_file) {
      FastaEntry f;
      while (read_fasta_entry(*ifs, &f)) {
        if (f.sequence.empty()) break;
        patterns.push_back(f.sequence);
        patdeflines.push_back(f.defline);
      }
    } else if (opt.sts_pattern_file) {
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
      std::istringstream sis(opt.patterns);
      std::string p;
      while (sis >> p) patterns.push_back(p);
    }
  }
  if (patterns.empty()) return 0;
  ph.mark("Read primers");
  if (opt.ucdict) for (std::string &p : patterns) uppercase(p);

  const unsigned long n = patterns.size();
  const unsigned long N1 = (opt.rev_comp ? 2 : 1) * n;
  std::vector<std::string> patarray(N1 + 1);
  std::vector<std::pair<int, int>> patconst(N1 + 1);
  std::vector<std::string> patdefarray(opt.fasta_pattern_file ? n + 1 : 0);
  std::vector<StsEntry> stsarray(opt.sts_pattern_file ? n / 2 + 1 : 0);
  const unsigned K1 = (unsigned)opt.nmismatch + 1;
  std::vector<unsigned long> patcount(opt.counts ? N1 * K1 : 0, 0);
  std::vector<bool> maxpatcount(opt.counts && opt.maxcount > 0 ? N1 + 1 : 0, false);
  auto cidx = [&](unsigned long i, unsigned k) { return (i - 1) * K1 + k; };
  for (unsigned long i = 1; i <= n; ++i) {              // primer_match.cc:966-1076
    const std::string &pat = patterns[i - 1];
    const int L = (int)pat.length();
    patarray[i] = pat;
    if (opt.fasta_pattern_file) patdefarray[i] = patdeflines[i - 1];
    if (opt.sts_pattern_file && i % 2 == 1) stsarray[(i + 1) / 2] = sts[(i - 1) / 2];
    int &f1 = patconst[i].first, &s1 = patconst[i].second;
    f1 = opt.stlen > 0 ? opt.stlen : 0;
    if (opt.fplen > f1) f1 = opt.fplen;
    if (opt.edlen < 0 && L + opt.edlen > f1) f1 = L + opt.edlen;
    if (opt.tplen < 0 && L + opt.tplen > f1) f1 = L + opt.tplen;
    s1 = opt.edlen > 0 ? opt.edlen : 0;
    if (opt.tplen > s1) s1 = opt.tplen;
    if (opt.stlen < 0 && L + opt.stlen > s1) s1 = L + opt.stlen;
    if (opt.fplen < 0 && L + opt.fplen > s1) s1 = L + opt.fplen;
    if (opt.rev_comp) {
      patarray[i + n] = reverse_comp(pat);
      int &f2 = patconst[i + n].first, &s2 = patconst[i + n].second;
      f2 = opt.stlen > 0 ? opt.stlen : 0;
      if (opt.tplen > f2) f2 = opt.tplen;
      if (opt.edlen < 0 && L + opt.edlen > f2) f2 = L + opt.edlen;
      if (opt.fplen < 0 && L + opt.fplen > f2) f2 = L + opt.fplen;
      s2 = opt.edlen > 0 ? opt.edlen : 0;
      if (opt.fplen > s2) s2 = opt.fplen;
      if (opt.stlen < 0 && L + opt.stlen > s2) s2 = L + opt.stlen;
      if (opt.tplen < 0 && L + opt.tplen > s2) s2 = L + opt.tplen;
    }
  }

  // ---- database and engine (primer_match.cc:1086-1112) --------------------------------------
  SeqDb db(opt.database, opt.dbind, opt.alignments && opt.dbindex, opt.dbindex, opt.ucdict, opt.eos_char, opt.memmap);
  ph.mark("Loaded sequence database");
  int kernel = PM_KERNEL_AUTO, semantics = PM_SEM_AUTO;
  if (opt.node == 16) kernel = PM_KERNEL_BITPAR;
  else if (opt.node == 17 || opt.node == 0) kernel = PM_KERNEL_AUTO;
  else semantics = opt.node;                            // reproduce that reference engine's hit set
  GpuPatternMatch kt(kernel, (unsigned)opt.nmismatch, opt.eos_char, opt.wc, opt.tn, opt.indels, false, semantics);
  size_t maxlen = 0;
  for (unsigned long i = 1; i <= N1; ++i) {
    kt.add_pattern(patarray[i], i, patconst[i].first, patconst[i].second);
    maxlen = std::max(maxlen, patarray[i].size());
  }
  BufferChars &ff = db.chars();
  kt.init(ff);
  ph.mark("Primer index built, stream resident on the GPU");
  double t_scan = 0, t_report = 0;
  unsigned long nhits = 0;

  // ---- scan loop (primer_match.cc:1114-1268) ------------------------------------------------
  const size_t stride = maxlen + (size_t)opt.nmismatch + 2;
  pattern_hit_vector l;
  std::vector<pm_hit> hv;
  std::vector<pm_alignment> al;
  std::vector<char> opsbuf, textbuf;
  StsEntry null_sts;
  for (;;) {
    const auto ts0 = std::chrono::steady_clock::now();
    const bool more = kt.find_patterns(ff, l, opt.report_interval);
    const auto ts1 = std::chrono::steady_clock::now();
    t_scan += std::chrono::duration<double>(ts1 - ts0).count();
    if (!more && l.empty()) break;
    nhits += l.size();
    const int64_t oldcharspos = ff.pos();
    hv.resize(l.size()); al.resize(l.size());
    if (opt.alignments) { opsbuf.assign(l.size() * stride, 0); textbuf.assign(l.size() * stride, 0); }
    for (size_t j = 0; j < l.size(); ++j) { hv[j].end = l[j].key; hv[j].pid = (uint32_t)l[j].id; hv[j].k = l[j].value; hv[j].aux[0] = hv[j].aux[1] = hv[j].aux[2] = 0; }
    // counts only (-c): distances are enough, no alignment strings
    if (!l.empty() && (opt.alignments ? pm_align_hits_text(kt.handle(), hv.data(), hv.size(), al.data(), opsbuf.data(), textbuf.data(), stride)
                                      : pm_align_hits(kt.handle(), hv.data(), hv.size(), al.data())) != PM_OK) {
      fprintf(stderr, "Fatal error: alignment: %s\n", pm_last_error(kt.handle()));
      return 1;
    }
    for (size_t j = 0; j < l.size(); ++j) {
      const unsigned long pid = l[j].id;
      if (!pid || (opt.maxcount > 0 && maxpatcount[pid])) continue;
      const pm_alignment &pa = al[j];
      if (pa.editdist < 0 || pa.editdist > opt.nmismatch) {      // "Bogus hit" (primer_match.cc:1249-1263)
        fprintf(stderr, "Bogus hit returned to primer_match main()\n");
        if (opt.alignments) fprintf(stderr, "Problem sequence is near:\n>%s\n", db.get_header_data(l[j].key).header.c_str());
        else fprintf(stderr, "Approximate absolute sequence position:\n %lld\n", (long long)l[j].key);
        fprintf(stderr, "Problem primer:\n %s\n", patarray[pid].c_str());
        return 1;
      }
      if (opt.alignments) {
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
        a.P = opt.fasta_pattern_file ? patdefarray[ind] : std::string();
        a.q = patarray[pid]; a.Q = with_gaps(patarray[pid], ops, '^');
        a.r = rc ? "R" : "F"; a.R = rc ? " REVCOMP" : "";
        a.t = mt; a.T = with_gaps(mt, ops, 'v'); a.A = ops;
        a.h = h.header; a.H = h.short_header; a.f = h.index;
        a.sts = opt.sts_pattern_file ? &stsarray[(ind + 1) / 2] : &null_sts;
        alignformat(out, opt.alignformat, a);
      }
      if (opt.counts) {
        patcount[cidx(pid, (unsigned)pa.editdist)]++;
        if (opt.maxcount > 0) {
          unsigned long count = 0;
          for (unsigned k = 0; k < K1; ++k) count += patcount[cidx(pid, k)];
          if (count >= (unsigned)opt.maxcount) maxpatcount[pid] = true;
        }
      }
    }
    l.clear();
    ff.pos(oldcharspos);
    t_report += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count();
  }
  if (opt.verbose) fprintf(stderr, "scan (find_patterns) %.3f s, re-align + report %.3f s, %lu hits\n", t_scan, t_report, nhits);
  ph.mark("Scanned sequence database");

  // ---- counts (primer_match.cc:1270-1328) ---------------------------------------------------
  if (opt.counts) {
    std::vector<unsigned long> counts(K1);
    for (unsigned long i = 1; i <= n; ++i) {
      unsigned long total = 0;
      for (unsigned k = 0; k < K1; ++k) { counts[k] = patcount[cidx(i, k)]; total += counts[k]; }
      bool gtmax = opt.maxcount > 0 ? (bool)maxpatcount[i] : false;
      const std::string patdef = opt.fasta_pattern_file ? patdefarray[i] : std::string();
      const StsEntry &stsref = opt.sts_pattern_file ? stsarray[(i + 1) / 2] : null_sts;
      if (!opt.aggregate) countformat(out, opt.countformat, i, patarray[i], patdef, patarray[i], "F", "", total, counts, (unsigned)opt.nmismatch, gtmax, stsref);
      if (opt.rev_comp) {
        if (!opt.aggregate) { total = 0; std::fill(counts.begin(), counts.end(), 0ul); gtmax = false; }
        for (unsigned k = 0; k < K1; ++k) { counts[k] += patcount[cidx(i + n, k)]; total += patcount[cidx(i + n, k)]; }
        if (opt.maxcount > 0) gtmax = gtmax || maxpatcount[i + n];
        if (!opt.aggregate) countformat(out, opt.countformat, i, patarray[i], patdef, patarray[i + n], "R", " REVCOMP", total, counts, (unsigned)opt.nmismatch, gtmax, stsref);
      }
      if (opt.aggregate) countformat(out, opt.countformat, i, patarray[i], patdef, "", "", "", total, counts, (unsigned)opt.nmismatch, gtmax, stsref);
    }
  }
  out.flush();
  return 0;
}

// pm_pcr_match -- the reference's pcr_match command line on the MI355X engine
// (BASELINE.json configs[4]; reference pcr_match.cc:40-1265).
//
// Primer pairs (ids 2j-1 forward, 2j reverse; n+1..2n their reverse complements,
// pcr_match.cc:776-906) are searched like primer_match's primers; the pairing stage
// (pcr_match.cc:948-1259) then joins, for every hit, the partner primer's hits that lie within the
// amplicon-length window on the same strand arrangement, re-aligns both ends, keeps pairs inside
// one FASTA entry with length in [-m, -M] (and within -d of the UniSTS size) and prints them
// through the -A format language (pcr_match.cc:339-686).
#include <unistd.h>
#include <thread>

#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "seq_io.h"

using namespace pmgpu;

namespace {

struct Options {
  bool primers_from_file = false, primers_from_sts = false, primers_from_fasta = false;
  std::string primer_arg, db_path, out_path;
  int max_edits = 0;
  char eos = '\n';
  std::string hit_format = ">%h\\n %>T %>s ... %l ... %<e %<T\\n %>A  %!>s    %!l    %!<e  %<A\\n %>Q %>r%!>s    %!l    %!<e%<r %<Q %a%R\\n";
  bool chatty = false, map_db = true;
  int db_variant = 0, engine_choice = 0;
  bool iupac = false, text_n_matches = false, to_upper = false, any_orientation = false, both_strands = false;
  int amplicon_max = 2000, amplicon_min = 0, size_slack = -1, zone_start = 0, zone_end = 0, zone5 = 0, zone3 = 0, seed_len = 0;
  bool with_indels = true, length_between = false;
  unsigned long progress_every = 1000;
};

[[noreturn]] void usage(const char *msg = nullptr) {
  if (msg && *msg) fprintf(stderr, "%s\n\n", msg);
  fprintf(stderr,
          "Usage: pm_pcr_match [options]\n\n"
          "  -i <sequence-database>  database prepared by (pm_)compress_seq. Required.\n"
          "  --ranks <n>             one process per GPU, the database sharded by position (default $PM_RANKS or 1)\n"
          "  -p <sequences> | -P <file> | -S <unists-file> | -F <fasta-file>   primer pairs (\"-\" = stdin)\n"
          "  -o <output-file>  -k <edits> | -K <mismatches>  -r  -a  -s -e -5 -3 <n|~n>  -u  -w  -W  -E <int>\n"
          "  -m <min amplicon>  -M <max amplicon, default 2000>  -d <deviation from UniSTS size>  -b\n"
          "  -A <format>  -R <int>  -N <int>  -D (0|1|2|3|4)  -B  -v  -h\n");
  exit(1);
}

int tilde(const char *a) { return a[0] == '~' ? -atoi(a + 1) : atoi(a); }

Options parse(int argc, char **argv) {
  Options o;
  int c;
  while ((c = getopt(argc, argv, "p:i:o:P:S:F:E:R:k:K:s:e:5:3:x:hrvVubaA:BD:wWN:M:m:d:")) != -1) switch (c) {
      case 'p': o.primer_arg = optarg; o.primers_from_file = false; break;
      case 'P': o.primer_arg = optarg; o.primers_from_file = true; break;
      case 'S': o.primer_arg = optarg; o.primers_from_sts = true; break;
      case 'F': o.primer_arg = optarg; o.primers_from_fasta = true; break;
      case 'i': o.db_path = optarg; break;
      case 'o': o.out_path = optarg; break;
      case 'k': o.max_edits = atoi(optarg); o.with_indels = true; break;
      case 'K': o.max_edits = atoi(optarg); o.with_indels = false; break;
      case '3': o.zone3 = tilde(optarg); break;
      case '5': o.zone5 = tilde(optarg); break;
      case 's': o.zone_start = tilde(optarg); break;
      case 'e': o.zone_end = tilde(optarg); break;
      case 'x': o.seed_len = atoi(optarg); break;
      case 'R': o.progress_every = (unsigned long)atoi(optarg); break;
      case 'A': o.hit_format = optarg; break;
      case 'w': o.iupac = true; o.text_n_matches = false; break;
      case 'W': o.iupac = true; o.text_n_matches = true; break;
      case 'u': o.to_upper = true; break;
      case 'D': o.db_variant = atoi(optarg); break;
      case 'N': o.engine_choice = atoi(optarg); break;
      case 'M': o.amplicon_max = atoi(optarg); break;
      case 'd': o.size_slack = atoi(optarg); break;
      case 'm': o.amplicon_min = atoi(optarg); break;
      case 'E': { int e0; if (!sscanf(optarg, "%i", &e0)) usage("Invalid end-of-sequence specification.\n"); o.eos = (char)e0; } break;
      case 'v': case 'V': o.chatty = true; break;
      case 'b': o.length_between = true; break;
      case 'r': o.both_strands = true; break;
      case 'a': o.any_orientation = true; break;
      case 'B': o.map_db = false; break;
      default: usage();
    }
  if ((o.primer_arg.empty() || o.db_path.empty()) && !o.chatty) usage();
  if (o.max_edits < 0) usage("Number of mismatches (-k) must be at least 0");
  if (o.db_variant < 0 || o.db_variant > 4) usage("Invalid integer for fasta database indexing (-D).");
  return o;
}

std::string spaces(long long fp) {                     // pcr_match.cc:243-247
  std::string ret = " ";
  while (fp /= 10) ret += ' ';
  return ret;
}
std::string spaces(const std::string &s) { return std::string(s.size(), ' '); }

struct End {                                           // one primer end of a pair as alignformat() sees it
  long long s, e, five, three, S, E;
  unsigned d;
  std::string p, patdef, q, Q, r, R, t, T, A;
};

struct PairFields {
  End a, b;                                            // '>' and '<' ends
  unsigned long i;
  const StsEntry *sts;
  bool ppo;
  std::string h, H;
  unsigned long f;
  std::string amplicon;
  unsigned long ncount;
};

void alignformat(std::ostream &os, const std::string &fmt, const PairFields &x) {
  const StsEntry &sts = *x.sts;
  for (size_t pos = 0; pos < fmt.size(); ++pos) {
    if (fmt[pos] == '%') {
      ++pos;
      if (pos >= fmt.size()) { os << "%"; continue; }
      bool widthonly = false;
      if (fmt[pos] == '!') { widthonly = true; ++pos; }
      int dirn = 0;
      if (pos < fmt.size() && fmt[pos] == '>') dirn = 1;
      if (pos < fmt.size() && fmt[pos] == '<') dirn = -1;
      if (dirn != 0) ++pos;
      const char code = pos < fmt.size() ? fmt[pos] : '\0';
      const End *en = dirn > 0 ? &x.a : (dirn < 0 ? &x.b : nullptr);
      switch (code) {
        case 's': if (en) { if (!widthonly) os << en->s; else os << spaces(en->s); } break;
        case 'e': if (en) { if (!widthonly) os << en->e; else os << spaces(en->e); } break;
        case 'l':
          if (en) os << en->e - en->s;
          else if (!widthonly) os << x.b.e - x.a.s;
          else os << spaces(x.b.e - x.a.s);
          break;
        case 'S': if (en) os << en->S; break;
        case 'E': if (en) os << en->E; break;
        case 'i': os << x.i; break;
        case 'd': if (en) os << en->d; break;
        case 'p': if (en) os << en->p; break;
        case 'P': if (en) os << en->patdef; break;
        case 'I': os << sts.id; break;
        case 'L':
          if (sts.sizeub != sts.sizelb) {
            if (dirn > 0) os << sts.sizelb; else if (dirn < 0) os << sts.sizeub; else os << sts.sizelb << "-" << sts.sizeub;
          } else {
            os << sts.sizelb;
          }
          break;
        case 'D': {                                    // int vs unsigned long comparisons as in the reference
          const int amplen = (int)(x.b.e - x.a.s);
          int deviance = 0;
          if ((unsigned long)(long)amplen > sts.sizeub) deviance = (int)((unsigned long)(long)amplen - sts.sizeub);
          else if ((unsigned long)(long)amplen < sts.sizelb) deviance = (int)(sts.sizelb - (unsigned long)(long)amplen);
          os << deviance;
        } break;
        case 'a': os << sts.acc; break;
        case 'O': os << sts.species; break;
        case '&': os << sts.altacc; break;
        case 'X': os << sts.chrom; break;
        case 'q': if (en) os << en->q; break;
        case 'Q': if (en) { if (!widthonly) os << en->Q; else os << spaces(en->Q); } break;
        case 'r': if (en) os << en->r; else os << (x.ppo ? "F" : "R"); break;
        case 'R': if (en) os << en->R; else os << (x.ppo ? "" : " REVERSE-STRAND"); break;
        case 't': if (en) os << en->t; break;
        case 'T': if (en) os << en->T; break;
        case 'A': if (en) { if (!widthonly) os << en->A; else os << spaces(en->A); } break;
        case 'h': os << x.h; break;
        case 'H': os << x.H; break;
        case 'f': os << x.f; break;
        case '@': os << x.amplicon; break;
        case '*': os << (x.ppo ? x.amplicon : reverse_comp(x.amplicon)); break;
        case 'N': os << x.ncount; break;
        case '%': os << "%"; break;
        case '0':
          os << x.H << " " << x.a.s + 1 << ".." << x.b.e << '\t' << sts.id << '\t';
          if (!sts.acc.empty()) {
            os << '\t' << sts.acc;
            if (!sts.chrom.empty()) {
              os << '\t' << sts.chrom;
              if (!sts.altacc.empty()) {
                os << '\t' << sts.altacc;
                if (!sts.species.empty()) os << '\t' << sts.species;
              }
            }
          }
          break;
        default: if (code) os << code;
      }
    } else if (fmt[pos] == '\\') {
      ++pos;
      if (pos < fmt.size()) {
        switch (fmt[pos]) {
          case 'n': os << std::endl; break;
          case 't': os << '\t'; break;
          case '\\': os << '\\'; break;
          default: os << fmt[pos];
        }
      } else {
        os << '\\';
      }
    } else {
      os << fmt[pos];
    }
  }
}

std::string with_gaps(const std::string &src, const std::string &ops, char gap_op) {
  std::string r;
  size_t p = 0;
  for (char op : ops) {
    if (op != gap_op) { r += p < src.size() ? src[p] : ' '; ++p; } else r += "-";
  }
  return r;
}

struct Hit { int64_t key; unsigned long id; unsigned char value; };

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

  // ---- primer pairs (pcr_match.cc:712-790) --------------------------------------------------
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
    } else if (opt.primers_from_sts) {
      StsEntry s;
      for (;;) {
        read_sts_entry(*ifs, &s);
        if (!(*ifs)) break;
        if (s.forward_primer.empty()) break;
        sts.push_back(s);
        patterns.push_back(s.forward_primer);
        patterns.push_back(s.reverse_primer);
      }
    } else if (opt.primers_from_fasta) {
      FastaEntry f;
      while (read_fasta_entry(*ifs, &f)) {
        if (f.sequence.empty()) break;
        patdeflines.push_back(f.defline);
        patterns.push_back(f.sequence);
      }
    } else {
      std::istringstream sis(opt.primer_arg);
      std::string p;
      while (sis >> p) patterns.push_back(p);
    }
  }
  if (patterns.empty()) return 0;
  if (patterns.size() % 2 != 0) usage("Odd number of primers!");
  // the ranks are forked only now: the primers may come from stdin ("-"), which the rank processes would otherwise
  // share -- one would drain it, or each would read a different slice and build different tables.  Nothing above touches a GPU.
  RankGroup ranks = RankGroup::launch(nranks);                      // returns in every rank process
  if (opt.to_upper) for (std::string &p : patterns) uppercase(p);
  if (opt.both_strands || opt.primers_from_sts) {
    opt.both_strands = true;
    for (size_t i = 1; i < patterns.size(); i += 2) patterns[i] = reverse_comp(patterns[i]);
  }

  const unsigned long n = patterns.size(), N1 = 2 * n;
  std::vector<std::string> patarray(N1 + 1);
  std::vector<std::pair<int, int>> patconst(N1 + 1);
  std::vector<int> patlen(N1 + 1);
  std::vector<StsEntry> stsarray(sts.size() + 1);
  std::vector<std::string> patdefarray(patdeflines.size() + 1);
  for (unsigned long i = 1; i <= n; ++i) {              // pcr_match.cc:806-906
    const std::string &pat = patterns[i - 1];
    const int L = (int)pat.length();
    int fplen = opt.zone5, tplen = opt.zone3;
    if (i % 2 == 0) { fplen = opt.zone3; tplen = opt.zone5; }
    patarray[i] = pat; patlen[i] = L;
    if (opt.primers_from_sts && i % 2 == 1) stsarray[(i + 1) / 2] = sts[(i - 1) / 2];
    if (opt.primers_from_fasta) patdefarray[i] = patdeflines[i - 1];
    int &f1 = patconst[i].first, &s1 = patconst[i].second;
    f1 = opt.zone_start > 0 ? opt.zone_start : 0;
    if (fplen > f1) f1 = fplen;
    if (opt.zone_end < 0 && L + opt.zone_end > f1) f1 = L + opt.zone_end;
    if (tplen < 0 && L + tplen > f1) f1 = L + tplen;
    s1 = opt.zone_end > 0 ? opt.zone_end : 0;
    if (tplen > s1) s1 = tplen;
    if (opt.zone_start < 0 && L + opt.zone_start > s1) s1 = L + opt.zone_start;
    if (fplen < 0 && L + fplen > s1) s1 = L + fplen;
    patarray[i + n] = reverse_comp(pat); patlen[i + n] = L;
    int &f2 = patconst[i + n].first, &s2 = patconst[i + n].second;
    f2 = opt.zone_start > 0 ? opt.zone_start : 0;
    if (tplen > f2) f2 = tplen;
    if (opt.zone_end < 0 && L + opt.zone_end > f2) f2 = L + opt.zone_end;
    if (fplen < 0 && L + fplen > f2) f2 = L + fplen;
    s2 = opt.zone_end > 0 ? opt.zone_end : 0;
    if (fplen > s2) s2 = fplen;
    if (opt.zone_start < 0 && L + opt.zone_start > s2) s2 = L + opt.zone_start;
    if (tplen < 0 && L + tplen > s2) s2 = L + tplen;
  }

  // ---- database and engine (pcr_match.cc:911-931) -------------------------------------------
  ph.mark("Read primer pairs");
  SeqDb db(opt.db_path, opt.db_variant, /*load_headers=*/true, /*check=*/true, /*upper_case=*/false, opt.eos, opt.map_db);
  ph.mark("Loaded sequence database");
  int kernel = PM_KERNEL_AUTO, semantics = PM_SEM_AUTO;
  if (opt.engine_choice == 16) kernel = PM_KERNEL_BITPAR;
  else if (opt.engine_choice != 17 && opt.engine_choice != 0) semantics = opt.engine_choice;
  GpuPatternMatch pm(kernel, (unsigned)opt.max_edits, opt.eos, opt.iupac, opt.text_n_matches, opt.with_indels, false, semantics, 0, &ranks);
  size_t maxlen = 0;
  for (unsigned long i = 1; i <= N1; ++i) {
    pm.add_pattern(patarray[i], i, patconst[i].first, patconst[i].second);
    maxlen = std::max(maxlen, patarray[i].size());
  }
  BufferChars &ff = db.chars();
  pm.init(ff);
  if (!ranks.single() && ranks.rank() != 0) {                        // this rank scans its shard, hands its records to rank 0 and is done
    pattern_hit_vector none;
    pm.find_patterns(ff, none, 1);
    ranks.leave(0);
  }
  ph.mark("Primer index built, stream resident on the GPU");
  double t_scan = 0, t_pair = 0;
  unsigned long nhits = 0, npairs = 0;

  // ---- scan + pairing (pcr_match.cc:937-1259) ------------------------------------------------
  const size_t stride = maxlen + (size_t)opt.max_edits + 2;
  const int slack = opt.with_indels ? opt.max_edits : 1;
  pattern_hit_vector l;
  StsEntry null_sts;
  for (;;) {
    const auto ts0 = std::chrono::steady_clock::now();
    const size_t before = l.size();
    const bool more = pm.find_patterns(ff, l, opt.progress_every);
    const auto ts1 = std::chrono::steady_clock::now();
    t_scan += std::chrono::duration<double>(ts1 - ts0).count();
    if (!more && l.empty()) break;
    nhits += l.size() - before;
    const int64_t oldcharspos = ff.pos();
    std::sort(l.begin(), l.end(), [](const pattern_hit &a, const pattern_hit &b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });
    // per pattern id: hit indices in position order
    std::map<unsigned long, std::vector<size_t>> m;
    for (size_t j = 0; j < l.size(); ++j) m[l[j].id].push_back(j);
    std::vector<int64_t> live(l.size());                // the key field the reference zeroes (pcr_match.cc:1221)
    for (size_t j = 0; j < l.size(); ++j) live[j] = l[j].key;

    // pass A: who pairs with whom (depends only on positions and the order of processing)
    std::vector<std::pair<size_t, size_t>> todo;        // (hit, partner)
    for (size_t it = 0; it < l.size(); ++it) {
      const unsigned long pid = l[it].id;
      const int64_t pos = l[it].key;
      unsigned long pid1 = 0, pid2 = 0;
      if (pid <= n && pid % 2 == 1) pid1 = pid + 1;
      else if (pid > n && (pid - n) % 2 == 0) pid1 = pid - 1;
      if (opt.any_orientation) {
        if (pid <= n) {
          if (pid % 2 == 1) pid2 = pid + n + 1;
          else { pid1 = pid - 1; pid2 = pid + n - 1; }
        } else {
          if (pid % 2 == 0) pid2 = pid - n - 1;
          else { pid1 = pid + 1; pid2 = pid - n + 1; }
        }
      }
      const unsigned long pair = (pid - (pid > n ? n : 0) + 1) / 2;
      int64_t stretch_max = opt.amplicon_max, stretch_min = opt.amplicon_min;
      if (opt.length_between) {
        int64_t plen = 0;
        if (pid1 != 0) plen = patlen[pid1];
        if (pid2 != 0 && patlen[pid2] > plen) plen = patlen[pid2];
        stretch_max += plen + patlen[pid];
      }
      if (opt.primers_from_sts && opt.size_slack >= 0) {  // unsigned comparisons as in the reference (:1040-1047)
        const uint64_t ub = (uint64_t)stsarray[pair].sizeub + (uint64_t)(int64_t)opt.size_slack;
        const uint64_t lb = (uint64_t)stsarray[pair].sizelb - (uint64_t)(int64_t)opt.size_slack;
        if ((uint64_t)stretch_max > ub) stretch_max = (int64_t)ub;
        if ((uint64_t)stretch_min < lb) stretch_min = (int64_t)lb;
      }
      stretch_max += pos - patlen[pid] + slack;
      stretch_min += pos - patlen[pid] - slack;
      if (oldcharspos < stretch_max && more) continue;   // window not scanned yet: keep the hit for the next round
      for (unsigned long partner : {pid1, pid2}) {
        if (partner == 0) continue;
        auto mit = m.find(partner);
        if (mit == m.end()) continue;
        const std::vector<size_t> &pq = mit->second;
        size_t q = std::lower_bound(pq.begin(), pq.end(), stretch_min, [&](size_t idx, int64_t v) { return l[idx].key < v; }) - pq.begin();
        for (; q < pq.size() && live[pq[q]] <= stretch_max; ++q)
          if (live[pq[q]]) todo.push_back(std::make_pair(it, pq[q]));
      }
      live[it] = 0;
    }

    // pass B: re-align every hit that takes part in a candidate pair (editdist_alignment, :1097-1115)
    std::vector<long> slot(l.size(), -1);
    std::vector<pm_hit> hv;
    for (const auto &pr : todo)
      for (size_t j : {pr.first, pr.second})
        if (slot[j] < 0) {
          slot[j] = (long)hv.size();
          pm_hit h; h.end = l[j].key; h.pid = (uint32_t)l[j].id; h.k = l[j].value; h.aux[0] = h.aux[1] = h.aux[2] = 0;
          hv.push_back(h);
        }
    std::vector<pm_alignment> al(hv.size());
    std::vector<char> opsbuf(hv.size() * stride, 0), textbuf(hv.size() * stride, 0);
    if (!hv.empty() && pm_align_hits_text(pm.handle(), hv.data(), hv.size(), al.data(), opsbuf.data(), textbuf.data(), stride) != PM_OK) {
      fprintf(stderr, "Fatal error: alignment: %s\n", pm_last_error(pm.handle()));
      return 1;
    }

    // pass C: filter and print (:1116-1217)
    for (const auto &pr : todo) {
      const size_t ja = pr.first, jb = pr.second;
      const pm_alignment &pa = al[slot[ja]], &pa1 = al[slot[jb]];
      if (pa.editdist < 0 || pa.editdist > opt.max_edits || pa1.editdist < 0 || pa1.editdist > opt.max_edits) continue;
      const unsigned long pid = l[ja].id, pid1 = l[jb].id;
      const long long len = pa.end - pa.start + 1, len1 = pa1.end - pa1.start + 1;
      const long long spe = db.get_seq_pos(pa.end), spe1 = db.get_seq_pos(pa1.end);
      const long long sps = spe - len + 1, sps1 = spe1 - len1 + 1;
      const long long pe = pa.end, pe1 = pa1.end, ps = pe - len + 1, ps1 = pe1 - len1 + 1;
      bool rc = pid > n, rc1 = pid1 > n;
      const unsigned long ind = pid - (rc ? n : 0), ind1 = pid1 - (rc1 ? n : 0);
      const unsigned long pind = ind < ind1 ? ind / 2 + 1 : ind1 / 2 + 1;
      const StsEntry &stsref = opt.primers_from_sts ? stsarray[pind] : null_sts;
      if (opt.both_strands) {
        if (ind % 2 == 0) rc = !rc;
        else if (ind1 % 2 == 0) rc1 = !rc1;
      }
      const long amplicon_len = !opt.length_between ? (long)(pe1 - ps) : (long)(ps1 - pe);
      if (!(db.is_subseq(ps, pe1) && amplicon_len <= opt.amplicon_max && amplicon_len >= opt.amplicon_min &&
            (!opt.primers_from_sts || opt.size_slack < 0 ||
             (((unsigned long)(amplicon_len + opt.size_slack) >= stsref.sizelb) && (amplicon_len <= (long)((int)stsref.sizeub) + opt.size_slack)))))
        continue;
      const HeaderData &h = db.get_header_data(pa.end);
      PairFields x;
      x.amplicon.resize((size_t)amplicon_len);
      x.ncount = 0;
      for (long i = 0; i < amplicon_len; ++i) {
        const int64_t q = ps + i;
        const char ch = q >= 0 && q < db.length() ? ff.ch((unsigned char)ff.c_str()[q]) : '\0';
        x.amplicon[(size_t)i] = ch;
        if (ch == 'N' || ch == 'n') ++x.ncount;
      }
      const std::string ops(opsbuf.data() + slot[ja] * stride), mt(textbuf.data() + slot[ja] * stride);
      const std::string ops1(opsbuf.data() + slot[jb] * stride), mt1(textbuf.data() + slot[jb] * stride);
      x.a = End{sps, spe, rc ? spe : sps, rc ? sps : spe, ps, pe, (unsigned)pa.editdist, patarray[ind],
                opt.primers_from_fasta ? patdefarray[ind] : std::string(), patarray[pid], with_gaps(patarray[pid], ops, '^'),
                rc ? "R" : "F", rc ? " REVCOMP" : "", mt, with_gaps(mt, ops, 'v'), ops};
      x.b = End{sps1, spe1, rc1 ? spe1 : sps1, rc1 ? sps1 : spe1, ps1, pe1, (unsigned)pa1.editdist, patarray[ind1],
                opt.primers_from_fasta ? patdefarray[ind1] : std::string(), patarray[pid1], with_gaps(patarray[pid1], ops1, '^'),
                rc1 ? "R" : "F", rc1 ? " REVCOMP" : "", mt1, with_gaps(mt1, ops1, 'v'), ops1};
      x.i = pind; x.sts = &stsref; x.ppo = ind < ind1;
      x.h = h.header; x.H = h.short_header; x.f = h.index;
      alignformat(out, opt.hit_format, x);
      ++npairs;
    }

    // keep only the hits whose window is not complete yet (:1222-1252)
    pattern_hit_vector keep;
    for (size_t j = 0; j < l.size(); ++j)
      if (live[j] != 0) keep.push_back(l[j]);
    l.swap(keep);
    ff.pos(oldcharspos);
    t_pair += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count();
  }
  if (opt.chatty) fprintf(stderr, "scan (find_patterns) %.3f s, pairing + re-align + report %.3f s, %lu primer hits, %lu amplicons\n", t_scan, t_pair, nhits, npairs);
  ph.mark("Scanned sequence database");
  out.flush();
  // The reference leaks its engine at exit (primer_match.cc: no delete of kt); tearing down the HIP runtime, the pinned
  // buffers and a 3 GB mapping took 0.3 s of a 0.9 s run.  Everything is flushed: leave without the destructors.
  fflush(stdout);
  fflush(stderr);
  if (!opt.out_path.empty()) fout.close();
  _exit(0);
}

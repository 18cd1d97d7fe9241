// oracle/ref_harness.cc -- TEST INFRASTRUCTURE ONLY.
//
// A small driver (our own code) that links against the *reference's* objects, compiled in place
// from /root/reference by oracle/Makefile, and dumps what the reference's PatternMatch engines
// return at the find_patterns() boundary (pattern_match.h:131) -- i.e. before primer_match's
// per-hit re-alignment (primer_match.cc:1135-1151).  It exists only in this container: it is
// used to generate tests/golden/* and to fuzz oracle/pm_oracle.c; nothing in the product path
// or on the GPU box depends on it.
//
// usage: ref_harness [-N sel] [-k edits | -K mismatches] [-r] [-m minka] [-w] [-W] [-s esb] [-e eeb]
//                    [-n] -i <db> -P <patterns.txt>
//   -i <db>   without -n: <db> is a raw byte stream file (e.g. a compress_seq .seq), read with
//             MapFileChars (alphabet size 256).  With -n: <db>.sqn + <db>.tbl, read with
//             Normalized<MapFileChars> (char_io.t:216-278).
//   -N sel    1..14 as pick_pattern_index (select.cc:197-265), 0 = auto, 100 = bare
//             shift_and_inexact (the candidate generator inside filter_bitvec).
//   -r        also add reverse complements as ids n+1..2n (primer_match.cc:1030).
// output: one line per engine hit, "<end> <id> <value>", in engine emission order, then
//         "#calls <number of find_patterns calls that returned true>".

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include <unistd.h>

#include "char_io.h"
#include "char_io.t"
#include "pattern_match.h"
#include "select.h"
#include "shift_and_inexact.h"
#include "util.h"

int main(int argc, char **argv) {
  int sel = 0, k = 0, minka = 1000, esb = 0, eeb = 0;
  bool indels = true, rc = false, wc = false, tn = false, norm = false;
  std::string db, patfile;
  int c;
  while ((c = getopt(argc, argv, "N:k:K:rm:wWs:e:ni:P:")) != -1) {
    switch (c) {
      case 'N': sel = atoi(optarg); break;
      case 'k': k = atoi(optarg); indels = true; break;
      case 'K': k = atoi(optarg); indels = false; break;
      case 'r': rc = true; break;
      case 'm': minka = atoi(optarg); break;
      case 'w': wc = true; break;
      case 'W': wc = true; tn = true; break;
      case 's': esb = atoi(optarg); break;
      case 'e': eeb = atoi(optarg); break;
      case 'n': norm = true; break;
      case 'i': db = optarg; break;
      case 'P': patfile = optarg; break;
      default: fprintf(stderr, "bad option\n"); return 2;
    }
  }
  if (db.empty() || patfile.empty()) { fprintf(stderr, "need -i and -P\n"); return 2; }

  std::vector<std::string> pats;
  { std::ifstream ifs(patfile.c_str()); std::string p; while (ifs >> p) pats.push_back(p); }
  size_t n = pats.size();
  size_t N1 = rc ? 2 * n : n;
  std::vector<std::string> patarray(N1 + 1);
  std::vector<std::pair<int,int> > patconst(N1 + 1);
  std::vector<int> patlen(N1 + 1);
  for (size_t i = 1; i <= n; i++) {
    patarray[i] = pats[i - 1];
    if (rc) patarray[i + n] = reverse_comp(pats[i - 1]);
  }
  for (size_t i = 1; i <= N1; i++) {
    patlen[i] = patarray[i].length();
    // forward: (esb,eeb); reverse complement mirrors them (primer_match.cc:1033-1060 with -s/-e)
    patconst[i] = (i <= n) ? std::make_pair(esb, eeb) : std::make_pair(esb, eeb);
  }

  CharacterProducer *cp;
  if (norm) cp = new Normalized<MapFileChars>(db, '\n');
  else      cp = new MapFileChars(db, '\n');

  PatternMatch *pm;
  if (sel == 100) pm = new shift_and_inexact(k, '\n', wc, tn, indels, false);
  else pm = pick_pattern_index(cp, sel, k, &patconst, &patlen, 0, wc, tn, indels, false, '\n', false);

  for (size_t i = 1; i <= N1; i++) pm->add_pattern(patarray[i], i, patconst[i].first, patconst[i].second);
  pm->init(*cp);

  pattern_hit_vector l(minka * 2);
  pattern_hit_vector::iterator it;
  bool more;
  long calls = 0;
  while ((more = pm->find_patterns(*cp, l, minka)) || !l.empty()) {
    FILE_POSITION_TYPE oldpos = cp->pos();
    if (more) calls++;
    for (it = l.begin(); it != l.end(); ++it) {
      printf("%lld %lu %d\n", (long long)it->key(), it->value().first->id(), (int)it->value().second);
    }
    l.clear();
    cp->pos(oldpos);
  }
  printf("#calls %ld\n", calls);
  return 0;
}

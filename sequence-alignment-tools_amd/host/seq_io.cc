// seq_io.cc -- see seq_io.h.
#include "seq_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pmgpu {

namespace {

[[noreturn]] void die(const char *a, const char *b = nullptr) {
  fprintf(stderr, "%s\n", a);
  if (b) fprintf(stderr, "%s\n", b);
  exit(1);
}

char revcomp_char(char c) {                           // util.cc:319-346
  switch (c) {
    case 'a': return 't'; case 'A': return 'T'; case 'c': return 'g'; case 'C': return 'G';
    case 'g': return 'c'; case 'G': return 'C'; case 't': return 'a'; case 'T': return 'A';
    case 'u': return 'a'; case 'U': return 'A';
    case 'm': return 'k'; case 'M': return 'K'; case 'r': return 'y'; case 'R': return 'Y';
    case 'y': return 'r'; case 'Y': return 'R'; case 'k': return 'm'; case 'K': return 'M';
    case 'v': return 'b'; case 'V': return 'B'; case 'h': return 'd'; case 'H': return 'D';
    case 'd': return 'h'; case 'D': return 'H'; case 'b': return 'v'; case 'B': return 'V';
    default: return c;                                // w, s, n and everything else map to themselves
  }
}

}  // namespace

std::string reverse_comp(const std::string &s) {
  std::string r(s);
  const size_t n = s.size();
  for (size_t i = 0; i < n; ++i) r[i] = revcomp_char(s[n - 1 - i]);
  return r;
}

std::string reverse(const std::string &s) { return std::string(s.rbegin(), s.rend()); }

void uppercase(std::string &s) {
  for (char &c : s) c = (char)toupper((unsigned char)c);
}

bool file_exists(const std::string &path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0;
}

bool read_file(const std::string &path, std::vector<unsigned char> *out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(out->data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == out->size();
}

MappedFile::~MappedFile() {
  if (data_) munmap(const_cast<unsigned char *>(data_), size_);
}

bool MappedFile::open(const std::string &path) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return false; }
  size_ = (size_t)st.st_size;
  if (size_ == 0) { close(fd); data_ = nullptr; return true; }
  void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { size_ = 0; return false; }
  data_ = static_cast<const unsigned char *>(p);
  return true;
}

SeqDb::SeqDb(const std::string &database, int format, bool load_headers, bool check, bool upper_case, char eos_char, bool memmap) {
  std::vector<unsigned char> bytes;
  bool mapped = false, raw = false;
  auto load = [&](const std::string &path) -> bool {
    if (memmap && map_.open(path)) { mapped = true; return true; }
    return read_file(path, &bytes);
  };
  if ((format == 0 && file_exists(database + ".sqn")) || format == 3) {          // select.t:30
    normalized_ = true;
    std::vector<unsigned char> tb;
    if (!load(database + ".sqn") || !read_file(database + ".tbl", &tb))
      die(("Can't open normalized sequence database " + database + ".sqn/.tbl").c_str());
    table_.assign(tb.begin(), tb.end());
  } else if ((format == 0 && file_exists(database + ".sqz")) || format == 4) {   // select.t:74
    // compressed database (char_io.t:18-214): codes of ceil(log2(table size)) bits, most significant bit first;
    // unpacked here into one code per byte -- the stream the engines see is the .sqn one plus the end-of-sequence
    // codes that fill up the last buffer
    normalized_ = true;
    std::vector<unsigned char> packed, tb;
    if (!read_file(database + ".sqz", &packed) || !read_file(database + ".tbz", &tb) || tb.empty())
      die(("Can't open compressed sequence database " + database + ".sqz/.tbz").c_str());
    table_.assign(tb.begin(), tb.end());
    unsigned bits = 1;
    while ((1u << bits) < tb.size()) ++bits;
    const size_t nchars = packed.size() * 8 / bits;
    bytes.resize(nchars);
    size_t bitat = 0;
    for (size_t i = 0; i < nchars; ++i) {
      unsigned code = 0;
      for (unsigned b = 0; b < bits; ++b, ++bitat) code = (code << 1) | ((packed[bitat >> 3] >> (7 - (bitat & 7))) & 1u);
      bytes[i] = (unsigned char)code;
    }
  } else if ((format == 0 && file_exists(database + ".seq")) || format == 2) {   // select.t:118
    if (!load(database + ".seq")) die(("Can't open indexed sequence database " + database + ".seq").c_str());
  } else {                                               // none of the files, or -D 1: the FASTA file itself (select.t:152-186)
    // StreamedFastaFile (fasta_io.t:448-751) as one pass over the file: newline, carriage return and blank are
    // skipped, '>' opens a header line, every other character is a stream character; an end-of-sequence character
    // in front of the first entry (eos_start), between entries and behind the last.  The index (entry start ->
    // header) comes out of the same pass.  For FASTA files with lines of one length -- the ones the reference
    // does not warn about -- this is the stream of <db>.seq without compress_seq's character filter.
    std::vector<unsigned char> in;
    if (!read_file(database, &in)) die(("Can't open sequence database " + database).c_str());
    raw = true;
    bytes.reserve(in.size());
    bytes.push_back((unsigned char)eos_char);
    const size_t n = in.size();
    size_t i = 0;
    bool ended = false;
    while (!ended) {
      if (i >= n) { bytes.push_back((unsigned char)eos_char); break; }
      unsigned char ch = in[i++];
      if (ch != '\n' && ch != '\r' && ch != ' ' && ch != '>') {
        bytes.push_back(upper_case ? (unsigned char)toupper(ch) : ch);
        continue;
      }
      while (ch == '\n' || ch == '\r' || ch == ' ') {
        if (i >= n) { bytes.push_back((unsigned char)eos_char); ended = true; break; }
        ch = in[i++];
      }
      if (ended) break;
      if (ch == '>') {
        const size_t hs = i;
        while (i < n && in[i] != '\n' && in[i] != '\r') ++i;
        const size_t he = i;
        if (i < n) { if (in[i] == '\r' && i + 1 < n && in[i + 1] == '\n') ++i; ++i; }
        const bool first = bytes.size() == 1;
        if (!first) bytes.push_back((unsigned char)eos_char);
        if (load_headers && (keys_.empty() || keys_.back() < (int64_t)bytes.size())) {   // _store_headers (fasta_io.t:468-474)
          keys_.push_back((int64_t)bytes.size());
          hdr_off_.push_back((int64_t)hdr_.size());
          hdr_len_.push_back((int64_t)(he - hs));
          hdr_.insert(hdr_.end(), in.begin() + hs, in.begin() + he);
        }
        if (first && i < n) { const unsigned char c2 = in[i++]; bytes.push_back(upper_case ? (unsigned char)toupper(c2) : c2); }   // (:576-585: taken as it is)
      } else {
        bytes.push_back(upper_case ? (unsigned char)toupper(ch) : ch);
      }
    }
    cache_.resize(keys_.size());
    cached_.assign(keys_.size(), false);
  }
  if (mapped) {
    length_ = (int64_t)map_.size();
    chars_ = new BufferChars(map_.data(), map_.size(), table_);
  } else {
    length_ = (int64_t)bytes.size();
    chars_ = new BufferChars(std::move(bytes), table_);
  }

  std::vector<int64_t> ikeys, ivals;
  const bool need_index = (check || load_headers) && !raw;
  if (raw) check = false;                               // check_fasta_file_params is IndexedFastaFile's (fasta_io.t:267-313)
  if (need_index) {
    std::vector<unsigned char> idb;
    if (!read_file(database + ".idb", &idb) || idb.size() < 8)
      die(("Can't open sequence index " + database + ".idb (text .idx indexes are not supported)").c_str());
    uint64_t cnt = 0;
    memcpy(&cnt, idb.data(), 8);                       // sortedvector::bread (sortedvector.t:774-781)
    if (idb.size() < 8 + cnt * 16) die("Bad format for indexed sequence database.", "Truncated index.");
    std::vector<std::pair<int64_t, int64_t>> el(cnt);
    for (uint64_t i = 0; i < cnt; ++i) {
      memcpy(&el[i].first, idb.data() + 8 + 16 * i, 8);
      memcpy(&el[i].second, idb.data() + 16 + 16 * i, 8);
    }
    std::stable_sort(el.begin(), el.end(), [](const std::pair<int64_t, int64_t> &a, const std::pair<int64_t, int64_t> &b) { return a.first < b.first; });
    for (auto &e : el) { ikeys.push_back(e.first); ivals.push_back(e.second); }
  }
  if (check && !ikeys.empty()) {                        // check_fasta_file_params (fasta_io.t:267-313), eos_start = true
    if (ikeys[0] == 0)
      die("Bad format for indexed sequence database.", "Parameter indicates EOS as first character, but first sequence starts at 0.");
    if (ikeys[0] > 1) die("Bad format for indexed sequence database.", "First sequence starts at position > 1.");
    const char c0 = length_ > 0 ? chars_->ch((unsigned char)chars_->c_str()[0]) : 0;
    if (c0 != eos_char) {
      fprintf(stderr, "Bad format for indexed sequence database.\nEOS character mismatch.\n");
      fprintf(stderr, "From indexed sequence database: %c\nFrom primer_match config: %c\n", c0, eos_char);
      exit(1);
    }
    if (upper_case && chars_->nch('a') >= 0)
      die("Bad format for indexed sequence database.", "Parameter indicates uppercase, but lowercase characters permitted.");
  }
  if (load_headers && ikeys.size() > 1) {               // index_headers (fasta_io.t:372-395)
    if (!read_file(database + ".hdr", &hdr_)) die(("Can't open header file " + database + ".hdr").c_str());
    for (size_t j = 0; j + 1 < ikeys.size(); ++j) {
      keys_.push_back(ikeys[j]);
      hdr_off_.push_back(ivals[j]);
      hdr_len_.push_back(ivals[j + 1] - ivals[j] - 1);
    }
    cache_.resize(keys_.size());
    cached_.assign(keys_.size(), false);
  }
}

bool SeqDb::locate(int64_t pos, size_t *idx) const {
  // locate_last_at_most(pos - 1): KeyOutOfRange when no key qualifies (fasta_io.t:155-170)
  auto it = std::upper_bound(keys_.begin(), keys_.end(), pos - 1);
  if (it == keys_.begin()) return false;
  *idx = (size_t)(it - keys_.begin()) - 1;
  return true;
}

int64_t SeqDb::get_seq_pos(int64_t pos) {
  size_t i;
  if (!locate(pos, &i)) return 0;
  return pos - keys_[i];
}

const HeaderData &SeqDb::get_header_data(int64_t pos) {
  size_t i;
  if (!locate(pos, &i)) return null_;
  if (!cached_[i]) {                                    // Lazy_Header_SI::read_header (fasta_io.t:113-131)
    HeaderData &h = cache_[i];
    h.index = (unsigned long)(i + 1);
    const int64_t off = hdr_off_[i], len = std::max<int64_t>(0, hdr_len_[i]);
    if (off >= 0 && off + len <= (int64_t)hdr_.size()) h.header.assign(reinterpret_cast<const char *>(hdr_.data()) + off, (size_t)len);
    const size_t p = h.header.find_first_of(" \t");
    h.short_header = p == std::string::npos ? h.header : h.header.substr(0, p);
    cached_[i] = true;
  }
  return cache_[i];
}

bool SeqDb::is_subseq(int64_t start, int64_t end) {
  size_t a, b;
  if (!locate(start + 1, &a) || !locate(end, &b)) return false;
  return a == b;
}

bool read_fasta_entry(std::istream &is, FastaEntry *e) {     // fasta_io.cc:11-58
  std::string line;
  int peek = is.peek();
  while (peek != EOF && (peek == '#' || peek == '\n')) { std::getline(is, line); peek = is.peek(); }
  e->defline.clear(); e->sequence.clear();
  if (peek == EOF) { is.get(); return false; }
  std::getline(is, line);
  e->defline = line.empty() ? std::string() : line.substr(1);
  peek = is.peek();
  while (peek != EOF && peek != '>' && peek != '#' && peek != '\n') {
    std::getline(is, line);
    e->sequence += line;
    peek = is.peek();
  }
  while (peek != EOF && (peek == '#' || peek == '\n')) { std::getline(is, line); peek = is.peek(); }
  return true;
}

void read_sts_entry(std::istream &is, StsEntry *e) {         // sts_io.cc:11-47
  std::string line;
  std::getline(is, line);
  std::istringstream iss(line);
  std::string size;
  iss >> e->id >> e->forward_primer >> e->reverse_primer >> size >> e->acc >> e->chrom >> e->altacc;
  const std::string::size_type p = size.find('-');
  if (p != std::string::npos) {
    e->sizelb = (unsigned long)atoi(size.substr(0, p).c_str());
    e->sizeub = (unsigned long)atoi(size.substr(p + 1).c_str());
  } else {
    e->sizelb = e->sizeub = (unsigned long)atoi(size.c_str());
  }
  std::string rest;
  if (iss.good()) std::getline(iss, rest); else rest.clear();
  e->species = rest;
}

}  // namespace pmgpu

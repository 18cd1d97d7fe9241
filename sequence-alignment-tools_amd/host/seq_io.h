// seq_io.h -- the data formats either side of the scan path, host C++.
//
// * SeqDb: the indexed sequence database written by compress_seq (ours or the reference's):
//   <db>.seq (characters), <db>.sqn + <db>.tbl (normalized codes + table) or <db>.sqz + <db>.tbz (bit-packed codes), <db>.idb (binary
//   index) and <db>.hdr (FASTA headers).  Mirrors what primer_match/pcr_match ask of
//   IndexedFastaFile<...,Lazy_Header_SI> (reference fasta_io.t:142-260,262-435): get_seq_pos,
//   get_header_data, is_subseq, and the parameter checks of check_fasta_file_params.
// * FASTA / UniSTS pattern readers (reference fasta_io.cc:11-58, sts_io.cc:11-47).
// * IUPAC reverse complement (reference util.cc:319-381).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <istream>
#include <string>
#include <vector>

#include "gpu_pattern_match.h"

namespace pmgpu {

std::string reverse_comp(const std::string &s);      // util.cc:374-381 (iupac_revcomp per character)
std::string reverse(const std::string &s);           // util.cc:383-390
void uppercase(std::string &s);

bool file_exists(const std::string &path);
bool read_file(const std::string &path, std::vector<unsigned char> *out);

// Read-only mapping of a whole file (what MapFileChars is to the reference, char_io.h:122-175);
// the pages come straight from the page cache, nothing is copied.
class MappedFile {
 public:
  MappedFile() {}
  ~MappedFile();
  MappedFile(const MappedFile &) = delete;
  MappedFile &operator=(const MappedFile &) = delete;
  bool open(const std::string &path);
  const unsigned char *data() const { return data_; }
  size_t size() const { return size_; }
 private:
  const unsigned char *data_ = nullptr;
  size_t size_ = 0;
};

struct HeaderData {                                   // Lazy_Header_SI (fasta_io.t:93-140)
  unsigned long index = 0;
  std::string header, short_header;
};

class SeqDb {
 public:
  // format: 0 auto (.sqn, then .sqz, then .seq, else the FASTA file itself; select.t:30,74,118,152), 1 raw FASTA,
  // 2 indexed (.seq), 3 normalized (.sqn+.tbl), 4 compressed (.sqz+.tbz).
  // load_headers = the `alignments && dbindex` argument of pick_fasta_file (primer_match.cc:1093).
  // check = ffp.check_params; upper_case / eos_char: ffp fields (fasta_io.t:18-30).
  // Errors follow the reference: message on stderr, exit(1).
  // memmap: map the sequence file (the reference's default) instead of reading it (-B, BufferedFileChars).
  SeqDb(const std::string &database, int format, bool load_headers, bool check, bool upper_case, char eos_char, bool memmap = true);
  BufferChars &chars() { return *chars_; }
  bool normalized() const { return normalized_; }
  const std::string &table() const { return table_; }
  int64_t length() const { return length_; }
  int64_t get_seq_pos(int64_t pos);                  // fasta_io.t:195-203
  const HeaderData &get_header_data(int64_t pos);    // fasta_io.t:186-194
  bool is_subseq(int64_t start, int64_t end);        // fasta_io.t:204-214
  size_t entries() const { return keys_.size(); }
 private:
  bool locate(int64_t pos, size_t *idx) const;       // last entry with key <= pos-1
  BufferChars *chars_ = nullptr;
  MappedFile map_;
  bool normalized_ = false;
  std::string table_;
  int64_t length_ = 0;
  std::vector<int64_t> keys_, hdr_off_, hdr_len_;
  std::vector<HeaderData> cache_;
  std::vector<bool> cached_;
  std::vector<unsigned char> hdr_;
  HeaderData null_;
};

struct FastaEntry { std::string defline, sequence; };
bool read_fasta_entry(std::istream &is, FastaEntry *e);   // false at end of input

struct StsEntry {                                     // sts_io.h:11-94
  std::string id, forward_primer, reverse_primer, acc, chrom, altacc, species;
  unsigned long sizelb = 0, sizeub = 0;
};
void read_sts_entry(std::istream &is, StsEntry *e);   // sts_io.cc:11-47 (fields of a short line keep their old values)

// -v of the command lines: wall-clock seconds per phase on stderr (the reference prints timestamp() lines there)
struct Phases {
  bool on = false;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  void mark(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[%8.3f s, +%7.3f] %s\n", std::chrono::duration<double>(now - t0).count(),
            std::chrono::duration<double>(now - last).count(), what);
    last = now;
  }
};

}  // namespace pmgpu

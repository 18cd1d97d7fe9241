// pm_seqdb_dump.cc -- what SeqDb (seq_io.cc) makes of a database, without a GPU: the stream as the engines see it
// (one byte per position, hex), the alphabet table and, per entry, start position and header.  Used by
// tests/test_seq_io_formats.py to hold the four database forms (select.t:22-188) against each other and against files
// the reference's compress_seq wrote.
// Usage: pm_seqdb_dump <database> <format 0..4> [upper_case 0|1]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "seq_io.h"

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "Usage: pm_seqdb_dump <database> <format> [upper_case]\n"); return 2; }
  const bool uc = argc > 3 && atoi(argv[3]) != 0;
  pmgpu::SeqDb db(argv[1], atoi(argv[2]), /*load_headers=*/true, /*check=*/false, uc, '\n', /*memmap=*/false);
  printf("normalized %d\n", db.normalized() ? 1 : 0);
  printf("table ");
  for (unsigned char c : db.table()) printf("%02x", c);
  printf("\nstream ");
  const char *p = db.chars().c_str();
  for (int64_t i = 0; i < db.length(); ++i) printf("%02x", (unsigned char)p[i]);
  printf("\nentries %zu\n", db.entries());
  unsigned long last = 0;
  for (int64_t i = 1; i <= db.length(); ++i) {                     // entry of every position (an end position as the engines report it): index + start
    const pmgpu::HeaderData &h = db.get_header_data(i);
    if (h.index == 0 || h.index == last) continue;                 // (position 1 ends on the leading end-of-sequence character: no entry)
    printf("entry %lu at %lld [%s|%s]\n", h.index, (long long)(i - db.get_seq_pos(i)), h.header.c_str(), h.short_header.c_str());
    last = h.index;
  }
  return 0;
}

/* include/pm_gpu.h -- C ABI of the MI355X (gfx950) multi-pattern matcher.
 *
 * This is the drop-in boundary for the reference's PatternMatch plugin surface
 * (reference: pattern_match.h:84-156) on the primer_match / pcr_match scan path.  One handle
 * replaces one PatternMatch engine instance; every entry point names the reference interface it
 * stands in for.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All functions return PM_OK (0) or a negative pm_status; pm_last_error() has the text.
 * A handle is owned by one host thread (the reference is single-threaded).
 *
 * Data flow:  pm_create -> pm_add_pattern xN -> pm_init[_device] -> pm_scan ... -> pm_destroy
 * Multi-GPU:  each rank pm_scan_candidates() on its shard of the stream; filter_bitvec option sets
 *             then pm_finalize_device_owned() on the same rank and the final hits are gathered
 *             (RCCL) in rank order; for the others the candidate records are gathered and one
 *             rank pm_finalize()s them in stream order.
 */
#ifndef PM_GPU_H
#define PM_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 1

typedef struct pm_handle pm_handle;

/* One hit, as the reference pushes it into pattern_hit_vector (pattern_match.h:82):
 *   end = key  = stream index after the last matched char (cp.pos() at emission),
 *   pid        = pattern_list_element::id() of value.first,
 *   k          = value.second (0 for exact engines, min level for shift_and_inexact, DP value
 *                for the seed+verify wrappers).
 * In *candidate* records (pm_scan_candidates) aux[] carries engine-private data; it is zero in
 * final hits. */
typedef struct {
  int64_t  end;
  uint32_t pid;
  uint8_t  k;
  uint8_t  aux[3];
} pm_hit;

typedef enum {
  PM_OK = 0,
  PM_E_INVALID = -1,       /* bad argument / call order */
  PM_E_UNSUPPORTED = -2,   /* valid in the reference, not implemented by this engine (message says what) */
  PM_E_NOMEM = -3,
  PM_E_HIP = -4,           /* HIP runtime error (no device, launch failure, ...) */
  PM_E_OVERFLOW = -5,      /* candidate buffer too small; *n_out holds the required count (> capacity) */
  PM_E_FATAL = -6          /* the reference would timestamp()+exit(1) here (e.g. select.cc:87-90) */
} pm_status;

/* Which reference engine's hit set to reproduce: the reference's own -N numbers
 * (select.cc:197-265).  PM_SEM_AUTO applies pick_pattern_index's automatic choice
 * (select.cc:101-141, NOPRIMEGEN build) to the patterns added. */
typedef enum {
  PM_SEM_AUTO = 0,
  PM_SEM_KEYWORD_TREE = 2,      /* keyword_tree<...>::find_patterns (keyword_tree.t:427); -N 1,2,3 */
  PM_SEM_SHIFT_AND = 4,         /* shift_and::find_patterns (shift_and.cc:208) */
  PM_SEM_FILTER_BITVEC = 5,     /* filter_bitvec::find_patterns (filter_bitvec.cc:73) */
  PM_SEM_EXACT_BASES = 8,       /* exact_bases::find_patterns (exact_bases.cc:69); -N 7..10 */
  PM_SEM_EXACT_HALVES = 12,     /* exact_halves::find_patterns (exact_halves.cc:120); -N 11..14 */
  PM_SEM_SHIFT_AND_INEXACT = 100 /* bare shift_and_inexact::find_patterns (shift_and_inexact.cc:249) */
} pm_semantics;

/* Which kernel family computes it (the "-N 16 / -N 17" of INTEGRATION.md). */
typedef enum {
  PM_KERNEL_AUTO = 0,
  PM_KERNEL_BITPAR = 16,   /* bit-parallel Shift-And / Wu-Manber rows, any alphabet, <= 6 accepted stream codes */
  PM_KERNEL_SEED = 17      /* 2-bit packed k-mer seeds (LDS filter) + verify; A,C,G,T patterns <= 32 nt */
} pm_kernel;

/* Replaces the constructor arguments of the reference engines as pick_pattern_index passes them
 * (select.cc:19-30): nmismatch, indels, wildcard, textn, eos.  dna_mut is not supported. */
typedef struct {
  int32_t abi_version;     /* PM_ABI_VERSION */
  int32_t semantics;       /* pm_semantics */
  int32_t kernel;          /* pm_kernel */
  int32_t k;               /* -k / -K value */
  int32_t indels;          /* 1 = -k (edits), 0 = -K (substitutions only) */
  int32_t wildcards;       /* -w/-W: IUPAC pattern classes (bit-parallel kernel family) */
  int32_t text_n;          /* -W: a text N also matches (shift_and.cc:112) */
  int32_t eos;             /* raw end-of-sequence char, '\n' in the CLIs */
  int32_t device;          /* HIP device ordinal */
  int32_t reserved[7];
} pm_config;

/* new engine (reference: `new shift_and(...)` etc. in select.cc:197-265). */
int pm_create(const pm_config *cfg, pm_handle **out);

/* PatternMatch::add_pattern (pattern_match.h:116): pattern text, caller's id (CLIs use 1..N1,
 * primer_match.cc:1105-1107), exact_start_bases / exact_end_bases. */
int pm_add_pattern(pm_handle *h, const char *pat, size_t len, uint64_t id, int32_t esb, int32_t eeb);

/* PatternMatch::init(CharacterProducer&) (pattern_match.h:130) for a host-resident stream:
 * `text` = the bytes getnch() would return (c_str() of the mmap, char_io.h:167-169), n bytes;
 * `table` = cp.ch(0..size-1) (the .tbl of a Normalized<> stream, char_io.t:216-246) or NULL for
 * a raw stream (size 256, identity).  The bytes are copied to HBM; `text` must stay valid until
 * pm_destroy (the verify stage reads windows from it). */
int pm_init(pm_handle *h, const uint8_t *text, int64_t n, const uint8_t *table, int32_t table_len);

/* Same, for a stream that is already resident in HBM (borrowed; 4-byte aligned).  `hip_stream`
 * is a hipStream_t (NULL = default stream) on which all work of this handle is enqueued. */
int pm_init_device(pm_handle *h, const void *d_text, int64_t n, const uint8_t *table, int32_t table_len,
                   void *hip_stream);

/* PatternMatch::find_patterns (pattern_match.h:131) over the stream range [begin,end):
 * appends to out[0..cap) every final hit with begin < hit.end <= end whose cluster/dedup fate is
 * decided (filter_bitvec.cc:118-121 defers the rest to the next call), sorted by (end,pid).
 * Ranges must be consecutive and increasing between pm_reset()s; end == n flushes everything.
 * *more = 1 when out was too small: call again with begin == end to drain.
 * On hit-dense text a range whose internal record lists would outgrow 2^30 records is scanned in
 * pieces by the library (pm_scan_stats out[6] counts the halvings of the piece length); the hits are those of the
 * whole range, in the same order. */
int pm_scan(pm_handle *h, int64_t begin, int64_t end, pm_hit *out, size_t cap, size_t *n_out, int *more);

/* pm_scan without the copy: *hits points at the final hits of the range -- plus any an earlier pm_scan call left
 * undrained -- in the handle's own (pinned) buffer, *n says how many; the span stays valid until the next call on the
 * handle.  This is the form the PatternMatch plugin uses (host/plugin/gpu_pattern_match.cc): the reference's
 * find_patterns appends to the caller's pattern_hit_vector (pattern_match.h:131), so the plugin pushes the records
 * straight from this span.  Both forms share one pipeline: when the finalize stage of the option set runs on the GPU, the
 * scan of the range expected next (the same number of stream bytes, as primer_match.cc:1118's loop walks the stream) is
 * enqueued before the call returns, and the copy out of HBM, the caller's work on the hits and its next call overlap it.
 * A different next range, pm_reset or pm_destroy simply let that scan finish unused. */
int pm_scan_view(pm_handle *h, int64_t begin, int64_t end, const pm_hit **hits, size_t *n);

/* Device stage only, position-independent (what shards across GPUs): candidate records for
 * begin < end_pos <= end.  Records stay in HBM (pm_candidates_device) and are copied to `out`
 * when it is not NULL.  PM_E_OVERFLOW if more than the internal capacity (pm_set_capacity). */
int pm_scan_candidates(pm_handle *h, int64_t begin, int64_t end, pm_hit *out, size_t cap, size_t *n_out);

/* Asynchronous form for benchmarking and overlap: enqueue the scan on the handle's stream and
 * return; pm_scan_wait() synchronises and reports the count. */
int pm_scan_candidates_async(pm_handle *h, int64_t begin, int64_t end);
int pm_scan_wait(pm_handle *h, size_t *n_out);

/* HBM address of the records of the last pm_scan_candidates (for an RCCL gather) */
int pm_candidates_device(pm_handle *h, void **d_records, size_t *n);
int pm_set_capacity(pm_handle *h, size_t max_candidates);

/* Host stage: cluster / dedup / verify candidate records that arrive in any order within a
 * batch but batch-wise in increasing stream order (filter_bitvec.cc:88-177,
 * exact_halves.cc:140-190).  flags: PM_FINALIZE_LAST flushes deferred clusters (no more input),
 * PM_FINALIZE_SORTED orders the output by (end,pid) (otherwise unspecified, like the emission
 * order of the reference engines).  Text, where the verify needs it, comes from the host pointer
 * given to pm_init or, for pm_init_device, from windows fetched out of HBM. */
#define PM_FINALIZE_LAST 1
#define PM_FINALIZE_SORTED 2
int pm_finalize(pm_handle *h, const pm_hit *cands, size_t n, int64_t scanned_to, int flags,
                pm_hit *out, size_t cap, size_t *n_out);

/* Device form of pm_finalize: the records are in HBM (d_cands, or NULL = those of the last
 * pm_scan_candidates), the sort + clustering runs on the GPU and only final hits cross PCIe into
 * `out`.  Available for: exact engines and bare shift_and_inexact (pass-through); filter_bitvec
 * with -K and no exact-base constraints (sort + segmented pass), and with -k on the seed family
 * (A,C,G,T patterns of 20..32 characters, k <= 2: clusters and their DPs on the GPU against the
 * text in HBM); exact_halves on the seed family (exact_halves.cc:142,163,178: its per-pattern
 * "end beyond the last kept end" rule as a sort + one walk per pattern -- stateless, so only with
 * PM_FINALIZE_LAST on an engine state that is fresh since pm_init / pm_reset).
 * PM_E_UNSUPPORTED otherwise: use pm_finalize. */
int pm_finalize_device(pm_handle *h, const void *d_cands, size_t n, int64_t scanned_to, int flags,
                       pm_hit *out, size_t cap, size_t *n_out);

/* One shard of a position-sharded scan (the reference has no such call, its scan is
 * one serial pass -- filter_bitvec.cc:88-177 over the whole stream).  The records (d_cands, or
 * NULL = the last pm_scan_candidates) must hold every candidate with guard_lo < end <= guard_hi;
 * the call reports the final hits with own_lo < end <= own_hi.  Because a cluster's hit ends between
 * its first and last candidate, the shards' outputs concatenate to exactly the single-scan result
 * as long as no same-pattern chain reaches from a guard edge into the owned range (a tandem repeat
 * longer than the guard band): that case returns PM_E_UNSUPPORTED.  guard_lo <= 0 and guard_hi ==
 * INT64_MAX declare the true start / end of the stream (nothing can be hidden beyond them).  Only
 * for the option sets pm_finalize_device supports; implies PM_FINALIZE_LAST. */
int pm_finalize_device_owned(pm_handle *h, const void *d_cands, size_t n, int64_t own_lo, int64_t own_hi,
                             int64_t guard_lo, int64_t guard_hi, int flags, pm_hit *out, size_t cap, size_t *n_out);

/* pm_finalize_device / pm_finalize_device_owned with out == NULL leave the final hits in HBM (the
 * exchange step of a position-sharded scan gathers them from there: no trip through host memory);
 * this returns their address and count.  Valid until the next scan or finalize call of the handle.
 * PM_FINALIZE_SORTED is not available in this form (PM_E_INVALID). */
int pm_final_hits_device(pm_handle *h, void **d_hits, size_t *n);

/* Copy n 16-byte records from HBM (a pointer obtained from pm_candidates_device /
 * pm_final_hits_device, or a gather buffer) to host memory on the handle's stream, synchronously. */
int pm_copy_records(pm_handle *h, const void *d_src, size_t n, pm_hit *out);

/* The caller's per-hit re-alignment (primer_match.cc:1135-1151, pcr_match.cc:1108-1127):
 * exact_alignment::align (pattern_alignment.cc:29-43) for k == 0, otherwise
 * editdist_alignment(key,key,k,eos,wc,tn,indels,dm,esb,eeb,false)::align with traceback
 * (pattern_alignment.cc:117-705).  start/end are stream indices (pa->start(), pa->end());
 * editdist is pa->editdist(), INT32_MAX for a constraint violation ("Bogus hit"). */
typedef struct { int64_t start, end; int32_t editdist, value; } pm_alignment;
int pm_align_hits(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out);

/* Same, plus what the caller's formatter prints per hit (primer_match.cc:1196-1202): the alignment
 * string pa->alignment_string() ('|' equal, '*' substitution, '^' insertion, 'v' deletion, '!'
 * constraint violation; pattern_alignment.h:122-165) and the matching text pa->matching_text(),
 * both NUL-terminated at ops + i*stride and text + i*stride.  stride > longest pattern + k. */
int pm_align_hits_text(pm_handle *h, const pm_hit *hits, size_t n, pm_alignment *out, char *ops, char *text, size_t stride);

/* ---- Multi-GPU exchange (the reference has no counterpart: its scan is one serial
 * pass, primer_match.cc:1118).  One rank = one process = one GPU.  The ranks scan position shards
 * (pm_scan_candidates on local stream indices), exchange their record counts by whatever means the
 * launcher has, then gather the 16-byte records out of HBM into rank 0 over xGMI (RCCL send/recv;
 * librccl.so is loaded on first use).  The unique id (PM_COMM_ID_BYTES) is made on rank 0 and handed
 * to the other ranks by the launcher.  Errors: pm_comm_last_error (NULL = the last failed create). */
#define PM_COMM_ID_BYTES 128
typedef struct pm_comm pm_comm;
int pm_comm_unique_id(void *id_out);
int pm_comm_create(int device, int rank, int world, const void *id, pm_comm **out);
/* counts[0..world): records every rank contributes (the same array on every rank).  Rank r sends
 * d_send[0..n_send) (n_send == counts[r]); rank 0 receives all of them in rank order and copies the
 * list to host_out (sum of counts records).  Collective: every rank of the communicator calls it. */
int pm_comm_gather(pm_comm *c, const void *d_send, size_t n_send, const uint64_t *counts, pm_hit *host_out);
void pm_comm_destroy(pm_comm *c);
const char *pm_comm_last_error(const pm_comm *c);

/* PatternMatch::init for a handle that only runs the HOST stage -- pm_finalize, pm_align_hits[_text]
 * -- over the whole stream: the merge rank of a position-sharded scan, whose own GPU holds one shard
 * only.  `text` (borrowed, n bytes, stays on the host) and `table` as for pm_init; the pattern tables
 * are built as for pm_init (they decide what the shards' records mean), nothing of the stream is
 * uploaded, and every scan entry point returns PM_E_INVALID. */
int pm_init_host(pm_handle *h, const uint8_t *text, int64_t n, const uint8_t *table, int32_t table_len);

/* Bring the HIP runtime up on `device` (no reference counterpart).  A command line calls this on a thread of its own at
 * process start, so that the runtime's start-up (0.06 - 0.15 s) runs beside reading the primers and mapping the database;
 * pm_create / pm_init work without it. */
int pm_prepare_device(int device);

/* PatternMatch::reset (pattern_match.h:134): forget scan state, keep patterns and text. */
int pm_reset(pm_handle *h);
void pm_destroy(pm_handle *h);

/* Introspection */
const char *pm_last_error(const pm_handle *h);        /* NULL handle: error of the last pm_create */
int pm_selected_semantics(const pm_handle *h);        /* resolved pm_semantics after pm_init */
int pm_selected_kernel(const pm_handle *h);
/* name of the dominant scan kernel and its launch geometry, for profiles */
int pm_describe(const pm_handle *h, char *buf, size_t buflen);

/* Timing of the scan kernels of the last pm_scan_candidates[_async], measured with HIP events on
 * the handle's stream: milliseconds and number of launches. */
int pm_last_kernel_time(pm_handle *h, float *ms, int *launches);

/* Counters of the last scan, for measurement (bench.py --stream-style; no reference counterpart): out[0] candidate
 * records; out[1] records handed from the first-stage kernel to the verify kernel (suspects of the pair plan, seed records
 * of the edit / halves plans; the pattern tile with most); out[2] scans the library repeated on its own since pm_init
 * because an internal buffer was too small (pm_last_kernel_time then covers every attempt); with PM_SEED_DEBUG bit 5 set
 * on the pair plan also out[3] blocks of 1024 positions, out[4] rounds of its second pass, out[5] key hits, summed over
 * waves and field pairs; out[6] times pm_scan halved its piece length since pm_init because a range's record lists would have
 * outgrown its bound (hit-dense text).  n <= 8 values are written. */
int pm_scan_stats(pm_handle *h, uint64_t *out, int n);

/* Measurement helper (no reference counterpart; DESIGN.md 4.6 "pair geometry for edits"): on a -K 2 handle of the pair plan,
 * time the 14-test pair geometry as the first stage of an edit-distance plan over the whole stream.  mode 1: substitution
 * compare (lower bound), mode 2: five-shift edit test on two patterns per slot.  No hits are produced. */
int pm_measure_pair_edit_floor(pm_handle *h, int mode, float *ms, uint64_t *suspects);

/* Duration of the one-off re-encoding of the stream to 2 bits per base that pm_init[_device] runs
 * for the seed kernel family (0 for the bit-parallel family): not part of a scan, reported so that a
 * reader can add it to a single cold pass. */
int pm_pack_time(pm_handle *h, float *ms);

/* pick_pattern_index's automatic choice (select.cc:101-141) without a handle. */
int pm_pick_semantics(int32_t alphabet_size, int32_t acgt_normalized, int32_t k, int32_t wildcards,
                      int32_t npat, const int32_t *patlen, const int32_t *esb, const int32_t *eeb);

/* Measurement helper (no reference counterpart; SURVEY.md 8(d) "measured ceiling from the same
 * box"): streams `bytes` of HBM at d_buf (16-byte aligned) through 16-byte loads `reps` times on
 * `stream` and reports the read rate in GB/s. */
int pm_measure_stream_read(const void *d_buf, size_t bytes, int reps, void *stream, float *gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif

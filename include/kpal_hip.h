/*
 * kpal_hip.h -- C-ABI of libkpal_hip.so: the MI355X (gfx950) k-mer counting and
 * profile-distance hot path of kPAL.
 *
 * The reference (kPAL 2.1.2.dev, pure Python) has no FFI for this path; its boundary is
 * the Python API of kpal/klib.py, kpal/metrics.py and kpal/kdistlib.py.  Each entry point
 * below names the reference interface it replaces (file:line into the reference tree).
 * kpal_amd/_native.py is the ctypes binding; INTEGRATION.md shows the stub a kPAL
 * maintainer would add.
 *
 * Conventions
 *  - every function returns an int status: 0 = ok, <0 = error (KPAL_E_*);
 *    kpal_last_error() returns a thread-local, NUL-terminated description.
 *  - the caller owns every host pointer; the library owns device memory behind kpal_ctx
 *    (or borrows caller-provided device pointers in the *_device variants).
 *  - calls are blocking unless stated otherwise; a ctx is bound to one GPU and one HIP
 *    stream and must not be used from two threads at once.
 *  - count vectors are int64, length 4^k, index = big-endian 2-bit packing of the k-mer
 *    with A/a=0 C/c=1 G/g=2 T/t=3 (klib.py:43-48, doc/method.rst:23-45).
 *  - 1 <= k <= KPAL_MAX_K = 16.  The reference takes any length and is bounded by memory alone (klib.py:149-151: a Python list
 *    of 4^k integers, then an int64 array); here the bound is the encoder's window -- a lane sees its own 16 bases and the 16
 *    before them -- and 4^16 int64 counts are 32 GiB of the 288 GB of one MI355X; k = 17 (128 GiB per profile, two of them
 *    for any distance) is refused with KPAL_E_INVALID -- ValueError in the Python face -- not miscounted.
 */
#ifndef KPAL_HIP_H
#define KPAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPAL_MAX_K 16

#define KPAL_OK 0
#define KPAL_E_INVALID (-1) /* bad argument (-> ValueError) */
#define KPAL_E_NOMEM (-2)   /* host or device allocation failed (-> MemoryError) */
#define KPAL_E_HIP (-3)     /* HIP runtime error (-> RuntimeError) */
#define KPAL_E_STATE (-4)   /* call sequence error, e.g. feed before begin (-> RuntimeError) */
#define KPAL_E_IO (-5)      /* a file could not be opened or read (-> OSError) */

/* metric selectors */
#define KPAL_PAIRWISE_PROD 0 /* metrics.pairwise['prod'], metrics.py:160 */
#define KPAL_PAIRWISE_SUM 1  /* metrics.pairwise['sum'],  metrics.py:161 */
#define KPAL_EUCLIDEAN 2     /* metrics.euclidean,        metrics.py:126-135 */

#define KPAL_COSINE 3        /* metrics.cosine_similarity, metrics.py:138-147 (kpal_profile_distance only) */

/* summary functions of dynamic smoothing: the keys of metrics.summary, metrics.py:165-170 */
#define KPAL_SUMMARY_MIN 0
#define KPAL_SUMMARY_AVERAGE 1
#define KPAL_SUMMARY_MEDIAN 2

/* counting strategies (kpal_count_set_strategy) */
#define KPAL_STRATEGY_AUTO 0
#define KPAL_STRATEGY_GLOBAL_ATOMIC 1 /* one 64-bit global atomic per k-mer; any k (AUTO only for feeds <= 256 KiB; cross-check of the other pipelines) */
#define KPAL_STRATEGY_LDS_DIRECT 2    /* whole 4^k table privatised in LDS; k <= 7 */
#define KPAL_STRATEGY_PARTITION 3     /* exact-offset radix partition (count, scan, scatter, histogram); 8 <= k <= 12 */
#define KPAL_STRATEGY_PARTITION_CHUNKED 5 /* one-pass partition into chunked bucket lists (no counting pass); 8 <= k <= 12; AUTO */
#define KPAL_STRATEGY_PARTITION2 4    /* two-level radix partition; 13 <= k <= 16 */
#define KPAL_STRATEGY_PARTITION2_QUADS 7 /* two-level partition of 4-k-mer items (quad_kernels.hpp); 13 <= k <= 16; AUTO for feeds >= 64 MiB that hold 1/8 byte per table entry (first piece of a count, a whole device buffer) or 3 bytes (any other) */
#define KPAL_STRATEGY_PARTITION_QUADS 6 /* partition of 4-k-mer items into aligned records (quad_kernels.hpp); 8 <= k <= 12; AUTO for feeds >= 32 MiB */

typedef struct kpal_ctx kpal_ctx;

const char *kpal_last_error(void);
const char *kpal_version(void);

int kpal_device_count(int *n);
int kpal_ctx_create(int device, kpal_ctx **out);
void kpal_ctx_destroy(kpal_ctx *ctx);
int kpal_sync(kpal_ctx *ctx);

/* ---- device memory helpers (bench / tests keep inputs resident in HBM without torch) ---- */
int kpal_dev_alloc(kpal_ctx *ctx, size_t nbytes, void **dev_out);
int kpal_dev_free(kpal_ctx *ctx, void *dev);
int kpal_memcpy_h2d(kpal_ctx *ctx, void *dev_dst, const void *host_src, size_t nbytes);
int kpal_memcpy_d2h(kpal_ctx *ctx, void *host_dst, const void *dev_src, size_t nbytes);
int kpal_memcpy_d2d(kpal_ctx *ctx, void *dev_dst, const void *dev_src, size_t nbytes);   /* asynchronous, ordered on the context's stream */

/* ---- counting: replaces Profile.from_sequences / from_fasta inner loops, klib.py:149-170 ----
 * Input is a flat byte stream; every byte outside AaCcGgTt separates sequences (klib.py:152,
 * the regex '[^AaCcGgTt]'); windows never span a separator or two feeds.  The host side joins
 * the sequences of one from_sequences() call with a single '\n'. */
int kpal_count_begin(kpal_ctx *ctx, int k);            /* klib.py:149-151: zeroed 4^k table */
int kpal_count_set_strategy(kpal_ctx *ctx, int strategy);
int kpal_count_feed(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes);        /* klib.py:154-168 */
/* Page-locked host memory for the caller's own staging (the host side of Profile.from_sequences gathers a list of reads into
 * it): kpal_count_feed_pinned lets the DMA engine read it in place -- no staging copy inside the library -- and returns when the
 * buffer may be refilled (the counting kernels may still run). */
int kpal_host_alloc(kpal_ctx *ctx, size_t nbytes, void **host_out);
int kpal_host_free(kpal_ctx *ctx, void *host);
int kpal_count_feed_pinned(kpal_ctx *ctx, const uint8_t *pinned_buf, size_t nbytes);   /* klib.py:154-168, as kpal_count_feed */
/* The same with the input already in HBM.  Stream-ordered: work queued on the context afterwards (kpal_memcpy_*,
 * kpal_synth_reads_device, the next feed) may reuse dev_buf; anything outside the context's stream waits for kpal_sync first.
 * The call does not wait for the GPU, with one exception: the FIRST whole-buffer feed of a k >= 13 count (FRESH mode of the
 * two-level quad pipeline) reads one word back before it returns -- whether a bypass list overflowed, in which case the piece is
 * counted again the classic way from the same buffer -- so that the buffer is the caller's again when the call is back. */
int kpal_count_feed_device(kpal_ctx *ctx, const void *dev_buf, size_t nbytes);
/* FASTA text of whole records (Profile.from_fasta, klib.py:97-112; tokenising the reference
 * delegates to Bio.SeqIO.parse, klib.py:111): header lines dropped, the lines of a record joined
 * with all ASCII whitespace removed, records separated; anything before the first header is
 * ignored.  The text goes to the device in 64 MiB chunks cut anywhere (pinned staging), is flattened there and counted; chunk
 * i + 1 is read, copied and flattened while chunk i is counted; k-mer windows span the chunk seams (never a record boundary).
 * Windows never span two CALLS: a call holds whole records -- or a record's tail / middle / head when the caller cuts one
 * giant record into ranges and hands every range but the first the k - 1 bases before it as `prefix` (below). */
int kpal_count_feed_fasta(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes);
/* The same for the bytes [begin, end) of a FILE (end = 0: to its end), read by the library itself: parallel preads straight
 * into the pinned staging buffers (KPAL_READ_THREADS, default 16) -- the path of `kpal count` on a FASTA file (kmer.py:112-146
 * -> klib.py:97-112) and of one rank's shard of the input (byte ranges cut at record boundaries; for one giant
 * record, ranges with a (k-1)-base read-only halo).  `prefix` (may be NULL): text that logically precedes the range -- for a
 * range that begins inside a record: ">\n" + the last k - 1 sequence bytes before `begin`; the range itself must begin at a
 * line start.  Without a prefix, text before the first header of the range is ignored, as at the start of a file. */
int kpal_count_feed_fasta_file(kpal_ctx *ctx, const char *path, uint64_t begin, uint64_t end, const uint8_t *prefix, size_t prefix_len);
/* The flattening alone (tests): host_out needs nbytes bytes; records are each preceded by '\n'. */
int kpal_fasta_flatten(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes, uint8_t *host_out, uint64_t *n_out);
/* Profile.from_fasta_by_record, klib.py:114-133, batched: host_flat holds n_records records,
 * record r = bytes [starts[r], starts[r+1]) (starts ascending, starts[n_records] = nbytes; put a
 * separator byte such as '\n' between records so that no window spans two of them).  host_out
 * receives n_records tables of 4^k int64, record-major.  Independent of the begin/feed/finish state. */
int kpal_count_records(kpal_ctx *ctx, int k, const uint8_t *host_flat, size_t nbytes, const uint64_t *host_starts,
                       size_t n_records, int64_t *host_out);
/* Profile.from_fasta_by_record, klib.py:114-133, with the records found on the device (what the reference leaves to
 * Bio.SeqIO.parse, klib.py:131).  kpal_fasta_records_begin takes FASTA text that holds whole records (text before its first
 * header line is ignored, as at the start of a file), flattens it and indexes its records: *n_records, *flat_bytes.
 * kpal_fasta_records_index copies out, per record, the offset of its header line ('>') in host_text (n_records values: the
 * caller reads the name there, klib.py:132) and its start in the flattened stream (n_records + 1 values, the last = flat_bytes);
 * either pointer may be NULL.  kpal_fasta_records_count counts records [first, first + n) of that text into host_out: n tables of
 * 4^k int64, record-major -- as many records per call as the caller has room for.  The index lives until the next
 * kpal_fasta_records_begin of the context.  Independent of the begin / feed / finish state. */
int kpal_fasta_records_begin(kpal_ctx *ctx, const uint8_t *host_text, size_t nbytes, uint64_t *n_records, uint64_t *flat_bytes);
int kpal_fasta_records_index(kpal_ctx *ctx, uint64_t *header_off, uint64_t *flat_start);
int kpal_fasta_records_count(kpal_ctx *ctx, int k, uint64_t first, uint64_t n, int64_t *host_out);
int kpal_fasta_records_count_device(kpal_ctx *ctx, int k, uint64_t first, uint64_t n, int64_t *dev_out);   /* the same into n x 4^k int64 of the caller's device memory (kpal_dev_alloc): profiles that stay in HBM */
/* The same over a FILE the library reads itself (bytes [begin, end) of path; end = 0: to its end): after _open, every _next
 * indexes the next piece of whole records (as kpal_fasta_records_begin does for text) -- *text_offset = the file offset of the
 * piece's first byte (header offsets of kpal_fasta_records_index are relative to it), *done = 1 (and no records) after the last
 * piece.  Between two _next calls kpal_fasta_records_index / _count serve the current piece. */
int kpal_fasta_records_file_open(kpal_ctx *ctx, const char *path, uint64_t begin, uint64_t end);
int kpal_fasta_records_file_next(kpal_ctx *ctx, uint64_t *n_records, uint64_t *flat_bytes, uint64_t *text_offset, int *done);
int kpal_fasta_records_file_tell(kpal_ctx *ctx, uint64_t *offset);   /* file offset of the first byte no piece has covered yet: where a closed scan is opened again */
int kpal_fasta_records_file_close(kpal_ctx *ctx);
int kpal_count_finish(kpal_ctx *ctx, int64_t *host_out /* 4^k, or NULL to keep the result on the device */); /* klib.py:170 */
/* Profile.balance (klib.py:285-298) on the count table in place, on the device: count + balance is the unit the
 * north-star metric is quoted on.  Call after the last feed, before kpal_count_finish (which then returns the balanced
 * counts).  For k >= 13 on the two-level quad pipeline the balance is fused into the pass that finalises the table
 * (one read and one write of the 4^k entries instead of two each); otherwise it is kpal_balance_device on the table. */
int kpal_count_balance(kpal_ctx *ctx);
/* Diagnostics (tests, A/B timing): the pipeline the last piece of the last feed actually took (a KPAL_STRATEGY_* value: AUTO
 * resolves per feed size and input composition) and the tile sizes of the quad scatters (wave-steps per wave and tile of level 1 /
 * level 2; 0 where not applicable).  The environment variables KPAL_QUAD_STEPS / KPAL_QUAD_STEPS2, read when the context is
 * created, force those tile sizes. */
int kpal_count_last_plan(kpal_ctx *ctx, int *strategy, int *steps1, int *steps2);
/* Diagnostics: how often the slow paths of the quad pipelines ran since the context was created (cumulative; tests assert that
 * skewed inputs -- poly-A, (AC)n, adapter prefixes: every k-mer still counted once, klib.py:157-168 -- really exercised them).
 * out[0] entries of the per-workgroup hot-item tables in use at kernel ends, out[1] items that rode in a spill list,
 * out[2] items a spill list could not hold (counted on the spot), out[3] pieces finalised FRESH (k >= 13: table written, not
 * added to), out[4] FRESH pieces run again the classic way (a bypass list overflowed), out[5] pieces through a quad pipeline,
 * out[6] pieces through the chunked / round-1 pipelines, out[7] pieces halved (pool or offset limits), out[8] quad pieces whose
 * scatter ran as the REPEAT instantiation (hot rows in the sample: repeat lanes straight to the hot-item table).  n <= KPAL_COUNT_STATS
 * values are written.  Synchronises the context's stream. */
#define KPAL_COUNT_STATS 9
int kpal_count_stats(kpal_ctx *ctx, uint64_t *out, int n);
int kpal_count_table(kpal_ctx *ctx, void **dev_table, uint64_t *n_bins); /* device pointer of the int64 table (for the RCCL reduce) */

/* Deterministic synthetic reads (same bytes as oracle/kpal_oracle.c
 * kpal_oracle_synth_reads): n_reads*(read_len+1) bytes, each read followed by '\n'. */
int kpal_synth_reads_device(kpal_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                            int read_len, int noisy, void *dev_out);

/* ---- multi-GPU: one process per GPU, the per-rank count tables merged by ONE RCCL reduce over xGMI ----
 * Replaces nothing the reference has (it is single-threaded); mirrors Profile.merge with the 'sum' merger, klib.py:269-283 /
 * metrics.py:175: every rank counts its shard of the reads into its own table (begin / feed), the tables add -- int64 sums,
 * bit-exact for any reduction order.  RCCL is bound at run time: rccl_library names the librccl.so to load (NULL: the
 * KPAL_RCCL_LIBRARY environment variable, then the loader's search path); inside a PyTorch process pass PyTorch's copy.
 *   rank 0:      kpal_comm_unique_id(lib, id)          -> hand the 128 bytes to the other ranks (MPI, a file, torch.distributed ...)
 *   other ranks: kpal_comm_probe(lib)                  -> can RCCL be bound here?  (no id, no bootstrap listener; agree on the answer
 *                                                         BEFORE the collective kpal_comm_init: a rank that cannot join leaves the others waiting)
 *   every rank:  kpal_comm_init(ctx, lib, rank, world, id)
 *   per job:     kpal_count_begin / feed ...; kpal_comm_reduce_table(ctx, root, balance); kpal_count_finish(root's host buffer)
 * kpal_comm_reduce_table: ncclReduce(int64, sum) of the count table onto `root` and, if balance != 0, Profile.balance there -- all
 * queued on the context's stream, no host synchronisation.  kpal_comm_reduce_table_async: the same on a copy of the table and on a
 * second stream, so that the next kpal_count_begin / feed overlaps it (throughput pipelines; two extra tables of HBM); the merged table
 * is then read with kpal_comm_merged_table (valid until the reduce after next, or until a kpal_count_begin that changes k or -- serial
 * form, where the merged table IS the count table -- starts the next count) after kpal_sync. */
#define KPAL_COMM_ID_BYTES 128
int kpal_comm_probe(const char *rccl_library);
int kpal_comm_unique_id(const char *rccl_library, uint8_t *id_out /* KPAL_COMM_ID_BYTES */);
int kpal_comm_init(kpal_ctx *ctx, const char *rccl_library, int rank, int world, const uint8_t *id);
int kpal_comm_destroy(kpal_ctx *ctx);
int kpal_comm_reduce_table(kpal_ctx *ctx, int root, int balance);
int kpal_comm_reduce_table_async(kpal_ctx *ctx, int root, int balance);
int kpal_comm_merged_table(kpal_ctx *ctx, void **dev_table, uint64_t *n_bins);
/* The bin-range merge (k >= 13, where one rank cannot usefully hold and balance the whole merged vector): ONE ncclReduceScatter
 * (int64 sum) leaves rank r the merged bins [r * 4^k / W, (r + 1) * 4^k / W) in place in its count table; with balance,
 * Profile.balance (klib.py:285-298) of that range -- the entries rc(i) it needs lie 1 / W on every rank: one all-to-all of
 * 4^k / W^2 entries per pair of ranks between two permutation kernels (csrc/range_index.hpp).  W must be a power of two with
 * W^2 <= 4^k.  kpal_comm_merged_range tells where a rank's part lies (first_bin, n_bins; after the whole-table reduces: 0, 4^k);
 * kpal_comm_gather_table (collective, after the range merge) completes every rank's table by one ncclAllGather. */
int kpal_comm_reduce_scatter_table(kpal_ctx *ctx, int balance);
int kpal_comm_gather_table(kpal_ctx *ctx);
int kpal_comm_merged_range(kpal_ctx *ctx, void **dev_table, uint64_t *first_bin, uint64_t *n_bins);
/* The two permutation kernels of that exchange for callers that move the blocks themselves (kpal_amd.dist over torch.distributed):
 * dev_send / dev_recv hold 4^k / W entries, block q = what goes to / came from rank q; dev_table is the whole table, of which only
 * the rank's range is read (pack) or balanced in place (unpack). */
int kpal_range_pack_device(kpal_ctx *ctx, int k, int rank, int world, const int64_t *dev_table, int64_t *dev_send);
int kpal_range_unpack_device(kpal_ctx *ctx, int k, int rank, int world, int64_t *dev_table, const int64_t *dev_recv);
/* kdistlib.distance_matrix (kdistlib.py:164-186) over several GPUs, sharded by BIN RANGE: every rank holds the same bins
 * [first, first + bin_count) of all P profiles (dev_slices: int64[P][bin_count] on the device, profile-major; ranges of different
 * ranks tile 0 .. 4^k; the LDS-staged and matrix-core kernels run when EVERY rank's range is a multiple of 64 bins and >= 4096 -- the
 * ranks agree on that with one 4-byte all-reduce before anything else, a ragged range anywhere puts all of them on the plain kernels), computes every pair's partial sum / term count / dot
 * product over its bins, and ONE all-reduce (two calls: fp64 sums, 64-bit counts -- 32 KB at P = 64) gives every rank the finished lower
 * triangle.  metric: prod / sum / euclidean; balancing needs whole profiles (kpal_balance_device before slicing). */
int kpal_comm_distance_matrix_device(kpal_ctx *ctx, int P, uint64_t bin_count, const int64_t *dev_slices, int metric,
                                     double *out_lower /* P(P-1)/2, kdistlib.py:179-186 order */);
int kpal_comm_max_f64(kpal_ctx *ctx, double *inout);   /* max of a host scalar over the ranks (timing: the slowest rank) */

/* ---- vector operations on 4^k int64 count vectors ---- */
/* Profile.balance, klib.py:285-298: c[i] += c[rc(i)] (palindromes doubled), in place. */
int kpal_balance(kpal_ctx *ctx, int k, int64_t *host_inout);
int kpal_balance_device(kpal_ctx *ctx, int k, int64_t *dev_inout);
/* Profile.reverse_complement, klib.py:394-412 (host helper, no GPU). */
uint64_t kpal_reverse_complement(uint64_t number, int k);
/* Profile.split, klib.py:300-327: forward/reverse each need (4^k + 4^(k/2))/2 entries when k is
 * even, 4^k/2 when odd; *n_out receives that length. */
int kpal_split(kpal_ctx *ctx, int k, const int64_t *host_counts, int64_t *host_forward,
               int64_t *host_reverse, uint64_t *n_out);
/* kmer.get_balance score, kmer.py:243-245: multiset(*split(), pairwise) fused in one pass. */
int kpal_strand_balance(kpal_ctx *ctx, int k, const int64_t *host_counts, int pairwise, double *out);

/* metrics.multiset (metrics.py:101-123) with pairwise prod/sum, and metrics.euclidean
 * (metrics.py:126-135), on two int64 vectors of n entries.  If do_balance != 0 both vectors are
 * balanced copies first (kdistlib.py:136-141; n must equal 4^k); inputs are never modified.
 * aux_out (optional): multiset -> number of bins with l!=0 or r!=0 (len(distances));
 * euclidean -> the exact int64 dot product (np.dot wraps like int64). */
int kpal_pair_distance(kpal_ctx *ctx, size_t n, const int64_t *host_left, const int64_t *host_right,
                       int metric, int do_balance, int k, double *out, int64_t *aux_out);
int kpal_pair_distance_device(kpal_ctx *ctx, size_t n, const int64_t *dev_left, const int64_t *dev_right,
                              int metric, int do_balance, int k, double *out, int64_t *aux_out);
/* float64 inputs (profiles after do_scale, kdistlib.py:149-157); multiset only. */
int kpal_pair_distance_f64(kpal_ctx *ctx, size_t n, const double *host_left, const double *host_right,
                           int pairwise, double *out, int64_t *aux_out);

/* kdistlib.distance_matrix values, kdistlib.py:179-186: out_lower[i*(i-1)/2 + j] =
 * distance(profiles[i], profiles[j]) for 0 <= j < i < P (row = left, column = right). */
int kpal_distance_matrix(kpal_ctx *ctx, int P, int k, const int64_t *const *host_profiles, int metric,
                         int do_balance, double *out_lower);
int kpal_distance_matrix_device(kpal_ctx *ctx, int P, int k, const int64_t *dev_profiles /* P x 4^k */,
                                int metric, int do_balance, double *out_lower);

/* ---- ProfileDistance with its full option set, kdistlib.py:25-51,126-161 ----
 * The constructor arguments of kdistlib.ProfileDistance that select built-in behaviour; a
 * user-supplied summary / pairwise / distance callable cannot enter a kernel and stays in Python. */
typedef struct kpal_distance_options {
    int do_balance;   /* kdistlib.py:139-141: balance copies of both profiles */
    int do_positive;  /* kdistlib.py:143-145: zero every bin that is zero in either profile */
    int do_smooth;    /* kdistlib.py:147-148: dynamic smoothing */
    int summary;      /* KPAL_SUMMARY_*: the `summary` function of the smoothing test */
    double threshold; /* collapse iff min(summary(left quarters), summary(right quarters)) <= threshold */
    int do_scale;     /* kdistlib.py:149-157: scale by the totals (metrics.get_scale) -> float64 profiles */
    int down;         /* metrics.scale_down: both factors <= 1 */
    int metric;       /* KPAL_PAIRWISE_PROD / _SUM (multiset), KPAL_EUCLIDEAN, KPAL_COSINE */
} kpal_distance_options;

/* ProfileDistance.distance(left, right), kdistlib.py:126-161; inputs are never modified. */
int kpal_profile_distance(kpal_ctx *ctx, int k, const int64_t *host_left, const int64_t *host_right,
                          const kpal_distance_options *opt, double *out);
int kpal_profile_distance_device(kpal_ctx *ctx, int k, const int64_t *dev_left, const int64_t *dev_right,
                                 const kpal_distance_options *opt, double *out);
/* ProfileDistance.dynamic_smooth(left, right), kdistlib.py:112-124: both vectors smoothed in place. */
int kpal_dynamic_smooth(kpal_ctx *ctx, int k, int64_t *host_left_inout, int64_t *host_right_inout,
                        int summary, double threshold);
/* kdistlib.distance_matrix values (kdistlib.py:179-186) for any option set: profiles are uploaded
 * (and balanced) once, every pair runs the option pipeline on the device. */
int kpal_profile_distance_matrix(kpal_ctx *ctx, int P, int k, const int64_t *const *host_profiles,
                                 const kpal_distance_options *opt, double *out_lower);

/* ---- profile summaries, merge, shrink ---- */
/* Profile.total / non_zero / mean / median / std (kpal/klib.py:193-225) of one int64 vector in two
 * streaming passes plus a radix select for the median.  total wraps like ndarray.sum(); mean and std
 * follow NumPy's float64 formulation (an exact 128-bit sum, then sum((x - mean)^2) / n), <= 1e-9
 * relative; median, min and max are exact. */
typedef struct kpal_profile_stats {
    int64_t total;     /* counts.sum(), int64 wrap */
    int64_t non_zero;  /* np.count_nonzero(counts) */
    int64_t min, max;
    double mean;       /* counts.mean() */
    double median;     /* np.median(counts): mean of the two middle elements for even n */
    double std;        /* counts.std(), population standard deviation */
} kpal_profile_stats;
int kpal_stats(kpal_ctx *ctx, size_t n, const int64_t *host_counts, kpal_profile_stats *out);
int kpal_stats_device(kpal_ctx *ctx, size_t n, const int64_t *dev_counts, kpal_profile_stats *out);

/* Profile.merge(profile, merger) for the built-in metrics.mergers (kpal/metrics.py:174-179,
 * kpal/klib.py:269-283): out = merger(left, right); out may be left or right. */
#define KPAL_MERGE_SUM 0   /* x + y */
#define KPAL_MERGE_XOR 1   /* (x + y) * logical_xor(x, y) */
#define KPAL_MERGE_INT 2   /* x * bool(y) */
#define KPAL_MERGE_NINT 3  /* x * logical_not(y) */
int kpal_merge(kpal_ctx *ctx, size_t n, const int64_t *host_left, const int64_t *host_right, int merger,
               int64_t *host_out);
int kpal_merge_device(kpal_ctx *ctx, size_t n, const int64_t *dev_left, const int64_t *dev_right, int merger,
                      int64_t *dev_out);

/* Profile.shrink(factor), kpal/klib.py:329-352: out[j] = sum of the 4^factor counts that share the
 * (k - factor)-mer prefix j; 1 <= factor < k (else KPAL_E_INVALID, the reference's ValueError);
 * out has 4^(k - factor) entries and must not overlap the input. */
int kpal_shrink(kpal_ctx *ctx, int k, int factor, const int64_t *host_counts, int64_t *host_out);
int kpal_shrink_device(kpal_ctx *ctx, int k, int factor, const int64_t *dev_counts, int64_t *dev_out);

/* ---- per-kernel timing with HIP events on the ctx stream (bench.py roofline leg) ---- */
int kpal_prof_enable(kpal_ctx *ctx, int on);
int kpal_prof_reset(kpal_ctx *ctx);
int kpal_prof_count(kpal_ctx *ctx, int *n_kernels);
int kpal_prof_get(kpal_ctx *ctx, int index, char *name_out, size_t name_cap, double *total_ms,
                  uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* KPAL_HIP_H */

// kpal_host.hpp -- host side shared by the translation units of libkpal_hip.so: errors, the context,
// workspace buffers, per-kernel HIP-event timing, the launch / dispatch macros and the few host functions
// one unit calls in another.  (kpal_ctx.hip: context + profiling API; kpal_count.hip: counting front end and the
// round-1 pipelines; kpal_quads.hip / kpal_quads2.hip: the quad record pipelines; kpal_vec.hip: balance, split,
// distances, matrices, options, summaries; kpal_multi.hip: multi-GPU entry points over RCCL.)
#pragma once
#include "../../include/kpal_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "kpal_device.hpp"
#include "host_pool.hpp"

#define KPAL_API extern "C" __attribute__((visibility("default")))

using namespace kpal;

// ----------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------
int set_err(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(e_ == hipErrorOutOfMemory ? KPAL_E_NOMEM : KPAL_E_HIP, "%s failed: %s (%s:%d)", \
                           #expr, hipGetErrorString(e_), __FILE__, __LINE__);                      \
    } while (0)

#define CHK(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != KPAL_OK) return rc_; \
    } while (0)


// ----------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    int name;
    hipEvent_t a, b;
};

struct kpal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;
    int num_cu = 256;
    int numa_node = -1;                      // NUMA node of the host the GPU is attached to (-1: unknown): staging buffers and copy threads go there
    // counting state
    int k = 0;
    int strategy = KPAL_STRATEGY_AUTO;
    bool counting = false;
    DevBuf table;  // int64[4^k]
    uint64_t bins = 0;
    size_t batch_bytes = (size_t)1 << 30;
    bool batch_bytes_set = false;            // KPAL_BATCH_BYTES given (else the chunked path uses its own maximum)
    uint64_t split_above = 0xFFFFFFFFull;   // two-level path: largest coarse bucket one batch may hold (32-bit offsets)
    size_t quad_pool_max = (size_t)30 << 30; // quad pipelines: largest record pool of one piece (a record store is a scalar base
                                             // + a 32-bit per-thread offset that spans 1/8 of the pool); larger pieces are halved
    // partition workspace
    DevBuf keys, cntmat, offs, bucket_start, slice_start;
    DevBuf chunk_meta, chunk_table, chunk_ovf, chunk_sorted;   // chunked one-level path
    DevBuf quad_meta, quad_meta2;            // quad path: rounds per workgroup, error word; level-2 rounds (k >= 13)
    uint32_t *quad_error_word = nullptr;
    bool chunk_error_armed = false;
    int quad_steps_forced = 0, quad_steps2_forced = 0;   // KPAL_QUAD_STEPS / KPAL_QUAD_STEPS2 at context creation (tests, A/B): tile sizes of the quad scatters
    bool quad_verbose = false;               // KPAL_QUAD_VERBOSE
    int quad_repeat_forced = -1;             // KPAL_QUAD_REPEAT=0 / 1: the scatter instantiation without / with the repeat lanes' shortcut (-1: by the sample)
    bool sample_hot_rows = false;            // the last row-load sample showed hot rows (quad_choose_steps)
    // what the last piece of the last feed took (kpal_count_last_plan): strategy, wave-steps per wave and tile of level 1 / level 2
    int plan_strategy = 0, plan_steps1 = 0, plan_steps2 = 0;
    // kpal_count_stats: pieces by pipeline and the FRESH pieces / re-runs, since the context was created
    uint64_t stat_fresh_pieces = 0, stat_fresh_reruns = 0, stat_quad_pieces = 0, stat_chunked_pieces = 0, stat_split_pieces = 0, stat_repeat_pieces = 0;
    // tile sizes chosen from the sample of an earlier feed of this count (kpal_count_begin clears them): a file streamed in
    // many feeds is sampled once per 16 feeds, not once per feed (the sample costs a D2H copy + a host synchronisation)
    int cached_steps1 = 0, cached_steps2 = 0;
    uint32_t cached_uses = 0;
    size_t cached_bytes = 0;
    // two-level quad pipeline: the staged forms of the last piece have not been added to the table yet (quad2_finalize)
    bool finalize_pending = false;
    const void *finalize_stage = nullptr;
    // FRESH mode of that pipeline (kpal_quads2.hip): kpal_count_begin leaves the table of a k >= 13 count UNZEROED
    // (table_zero_pending) -- the first piece, if it is a whole device feed on the two-level quad pipeline, lets its
    // finalisation WRITE the table instead of adding to it, and keeps the few counts that bypass the records in lists until
    // then (finalize_fresh).  Every other consumer of the table materialises the zeros first (table_ready).
    bool table_zero_pending = false;
    bool finalize_fresh = false;
    bool fresh_resolved = false;             // the overflow word of the pending FRESH piece has been read (quad2_resolve_fresh)
    bool fresh_feed = false;                 // set by kpal_count_feed_device around count_device_range: a whole device feed
    Span fresh_span = {};                    // the piece of a FRESH finalisation (re-run classically if a list overflowed)
    DevBuf direct_list, direct_meta;         // TableSink segments ((index << 32) | count entries); per-segment counts + overflow word
    uint32_t direct_seg = 16384;             // entries per segment (KPAL_DIRECT_SEG: tests force the overflow path)
    uint32_t direct_seg_used = 16384;        // ... of the current lists (sixteen times that for a feed with hot rows)
    bool quad_hist_unpacked = false;         // KPAL_HIST_PACKED=0: the 128 KiB histogram at every k (A/B, tests)
    bool direct_seg_forced = false;          // ... then the histogram stage's segment is as small
    uint32_t direct_seg_hist = 16384;        // entries of the last segment (histogram stage, shared) of the current lists
    uint32_t direct_nseg = 0;                // segments in use by the pending finalisation
    int level2_mode = 2;                     // level 2 of the two-level path (KPAL_LEVEL2): 0 count + exact offsets, 1 chunked per-tile runs, 2 chunked aligned lines (default)
    alignas(16) unsigned char chunk_pool_sent[96] = {};   // (ChunkPool) what the device copy of the pool descriptor holds
    void *chunk_pool_dev = nullptr;
    uint32_t chunk_meta_y = 0;               // coarse-bucket count the meta layout was cleared for
    uint32_t *chunk_error_word = nullptr;
    DevBuf residuals, cnt1, offs1, start1;  // two-level path (k = 13..16)
    // FASTA ingest (kpal_count.hip): raw text and flattened stream of two chunks in flight, scan metadata, the flattened tail of
    // the previous chunk (the k-1 bytes the next one's first windows begin in), the chunks' flattened sizes in pinned host memory
    DevBuf fa_raw[2], fa_flat[2], fa_meta[2], fa_tail;
    // record index of the text of the last kpal_fasta_records_begin (from_fasta_by_record): raw text, flattened stream, scan metadata,
    // record starts in the flattened stream (R + 1) and header offsets in the raw text (R) on the device and on the host
    DevBuf rec_raw, rec_flat, rec_meta, rec_starts, rec_hdr;
    std::vector<uint64_t> rec_starts_host, rec_hdr_host;
    uint64_t rec_n = 0, rec_nf = 0;
    // ... over a file the library reads itself (kpal_fasta_records_file_*): the open range, the unfinished record carried between pieces
    int rec_fd = -1;
    uint64_t rec_pos = 0, rec_end = 0, rec_piece_at = 0;
    std::vector<uint8_t> rec_carry;
    uint64_t *fa_nflat_host = nullptr;
    std::vector<void *> host_allocs;         // kpal_host_alloc buffers still owned by callers (released with the context at the latest)
    size_t fa_chunk = kStage;                // text bytes per chunk (KPAL_FASTA_CHUNK: tests put the seams everywhere)
    // host-feed staging
    static constexpr size_t kStage = (size_t)64 << 20;
    static constexpr size_t kStagePad = 64;
    void *pinned[2] = {nullptr, nullptr};
    DevBuf dstage[2];
    hipEvent_t ev_copied[2] = {nullptr, nullptr};
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    // scratch for vector ops
    DevBuf scratch[4];
    DevBuf partials, result;
    DevBuf opt_l, opt_r, opt_levels, opt_profiles;   // ProfileDistance option pipeline
    // the canonical tile pairs of the LDS-tiled balance kernels for the k they were last built for (kpal_vec.hip: canon_tiles)
    DevBuf canon;
    std::vector<uint32_t> canon_host;
    int canon_k = 0;
    // multi-GPU (kpal_multi.hip): RCCL communicator, the stream the pipelined reduce runs on, two side buffers for it
    void *comm = nullptr;                    // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_table_copied = nullptr;
    hipEvent_t ev_side_free[2] = {nullptr, nullptr};
    bool side_used[2] = {false, false};
    DevBuf side[2];
    int side_turn = 0;
    void *merged = nullptr;                  // where the last merged table lies (the count table or a side buffer)
    uint64_t merged_bins = 0;                // ... and how many bins it has (kpal_count_begin with another k discards it)
    uint64_t merged_first = 0;               // ... and which bin its first one is (bin-range merge: a rank holds its range only)
    DevBuf xsend, xrecv;                     // the packed blocks of the mirror exchange (kpal_comm_reduce_scatter_table)
    // profiling
    bool prof = false;
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<uint64_t> prof_launches;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    uint64_t prof_dropped = 0;               // launches whose timing events could not be recorded
};


int ensure(kpal_ctx *ctx, DevBuf &b, size_t bytes);
int prof_name_id(kpal_ctx *ctx, const char *name);
hipEvent_t prof_event(kpal_ctx *ctx);
int prof_collect(kpal_ctx *ctx);

struct ProfScope {
    kpal_ctx *ctx;
    ProfRec rec;
    bool on;
    ProfScope(kpal_ctx *c, const char *name) : ctx(c), on(c->prof)
    {
        if (on) {
            rec.name = prof_name_id(c, name);
            rec.a = prof_event(c);
            rec.b = prof_event(c);
            // a launch that cannot be timed is still launched: the pair is dropped, the failure is counted
            if (!rec.a || !rec.b || hipEventRecord(rec.a, c->stream) != hipSuccess) drop();
        }
    }
    void drop()
    {
        on = false;
        ++ctx->prof_dropped;
        if (rec.a) ctx->ev_pool.push_back(rec.a);
        if (rec.b) ctx->ev_pool.push_back(rec.b);
    }
    ~ProfScope()
    {
        if (on) {
            if (hipEventRecord(rec.b, ctx->stream) == hipSuccess) ctx->prof_pending.push_back(rec);
            else drop();
        }
    }
};

#define LAUNCH(ctx, name, kernel, grid, block, ...)                                       \
    do {                                                                                  \
        {                                                                                 \
            ProfScope ps_(ctx, name);                                                     \
            hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);       \
        }                                                                                 \
        HIPCHK(hipGetLastError());                                                        \
    } while (0)

#define CASE_K(N, ...)           \
    case N: {                    \
        constexpr int K = N;     \
        __VA_ARGS__;             \
    } break;

#define DISPATCH_K_1_16(k, ...)                                                                            \
    switch (k) {                                                                                            \
        CASE_K(1, __VA_ARGS__) CASE_K(2, __VA_ARGS__) CASE_K(3, __VA_ARGS__) CASE_K(4, __VA_ARGS__) CASE_K(5, __VA_ARGS__) CASE_K(6, __VA_ARGS__)     \
        CASE_K(7, __VA_ARGS__) CASE_K(8, __VA_ARGS__) CASE_K(9, __VA_ARGS__) CASE_K(10, __VA_ARGS__) CASE_K(11, __VA_ARGS__) CASE_K(12, __VA_ARGS__) \
        CASE_K(13, __VA_ARGS__) CASE_K(14, __VA_ARGS__) CASE_K(15, __VA_ARGS__) CASE_K(16, __VA_ARGS__)                                 \
    default:                                                                                                \
        return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);                           \
    }

#define DISPATCH_K_1_7(k, ...)                                                                         \
    switch (k) {                                                                                        \
        CASE_K(1, __VA_ARGS__) CASE_K(2, __VA_ARGS__) CASE_K(3, __VA_ARGS__) CASE_K(4, __VA_ARGS__) CASE_K(5, __VA_ARGS__) CASE_K(6, __VA_ARGS__) \
        CASE_K(7, __VA_ARGS__)                                                                                 \
    default:                                                                                            \
        return set_err(KPAL_E_INVALID, "LDS-direct strategy needs k <= 7 (k=%d)", k);                   \
    }

#define DISPATCH_K_13_16(k, ...)                                                                   \
    switch (k) {                                                                                   \
        CASE_K(13, __VA_ARGS__) CASE_K(14, __VA_ARGS__) CASE_K(15, __VA_ARGS__) CASE_K(16, __VA_ARGS__) \
    default:                                                                                       \
        return set_err(KPAL_E_INVALID, "two-level partition strategy needs 13 <= k <= 16 (k=%d)", k); \
    }

#define DISPATCH_K_12_16(k, ...)                                                                   \
    switch (k) {                                                                                   \
        CASE_K(12, __VA_ARGS__) CASE_K(13, __VA_ARGS__) CASE_K(14, __VA_ARGS__) CASE_K(15, __VA_ARGS__) CASE_K(16, __VA_ARGS__) \
    default:                                                                                       \
        return set_err(KPAL_E_INVALID, "staged quad histograms need 12 <= k <= 16 (k=%d)", k);     \
    }

#define DISPATCH_K_8_16(k, ...)                                                                   \
    switch (k) {                                                                                   \
        CASE_K(8, __VA_ARGS__) CASE_K(9, __VA_ARGS__) CASE_K(10, __VA_ARGS__) CASE_K(11, __VA_ARGS__) CASE_K(12, __VA_ARGS__)         \
        CASE_K(13, __VA_ARGS__) CASE_K(14, __VA_ARGS__) CASE_K(15, __VA_ARGS__) CASE_K(16, __VA_ARGS__)                               \
    default:                                                                                       \
        return set_err(KPAL_E_INVALID, "quad partition needs 8 <= k <= 16 (k=%d)", k);             \
    }

#define DISPATCH_K_8_12(k, ...)                                                                   \
    switch (k) {                                                                                   \
        CASE_K(8, __VA_ARGS__) CASE_K(9, __VA_ARGS__) CASE_K(10, __VA_ARGS__) CASE_K(11, __VA_ARGS__) CASE_K(12, __VA_ARGS__)         \
    default:                                                                                       \
        return set_err(KPAL_E_INVALID, "partition strategy needs 8 <= k <= 12 (k=%d)", k);         \
    }

#define CTX_ENTER(ctx)                                            \
    if (!(ctx)) return set_err(KPAL_E_INVALID, "ctx is NULL");    \
    HIPCHK(hipSetDevice((ctx)->device))


inline unsigned stream_grid(kpal_ctx *ctx, uint64_t n_items, unsigned block = 256)
{
    const uint64_t want = (n_items + block - 1) / block;
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cu * 8));
}

// ----------------------------------------------------------------------------------------------
// host functions shared between the units
// ----------------------------------------------------------------------------------------------
constexpr int kQuadsUseChunked = 2;   // launch_partition*_quads (AUTO): the sample shows a feed for the round-1 pipeline
constexpr int kSplitBatch = 1;        // launch_partition2 / launch_partition*_quads: the caller halves the piece
int launch_partition_quads(kpal_ctx *ctx, const Span &s);                 // kpal_quads.hip
int launch_partition2_quads(kpal_ctx *ctx, const Span &s, bool fresh = false);   // kpal_quads2.hip
int quad_choose_steps(kpal_ctx *ctx, const Span &s, uint32_t *load, int buckets, int slots, int waves, const int *candidates,
                      size_t n_candidates, int *steps_out, std::vector<double> *fine_per_step = nullptr);   // kpal_quads.hip
double quad_expected_backlog(const std::vector<double> &mu, int slots);   // kpal_quads.hip
constexpr double kQuadBacklogMax = 1500.0;   // quad_choose_steps: expected steady-state backlog a tile size may bring (list: 2048)
int quad2_finalize(kpal_ctx *ctx, bool balance);                          // kpal_quads2.hip: no-op unless a finalisation is pending
int quad2_resolve_fresh(kpal_ctx *ctx);                                   // kpal_quads2.hip: a FRESH piece whose lists overflowed is counted again (the fed buffer is read)
int table_ready(kpal_ctx *ctx);                                           // kpal_quads2.hip: zeros materialised, pending finalisation done: the table is the table
int launch_balance(kpal_ctx *ctx, int k, const int64_t *in, int64_t *out);   // kpal_vec.hip
int distance_matrix_core(kpal_ctx *ctx, int P, uint64_t n, const int64_t *prof, int metric, double *out_lower, bool allreduce,
                         int tiled = -1);   // kpal_vec.hip (tiled: -1 decided from n; 0 / 1 agreed between the ranks)
int comm_allreduce_partials(kpal_ctx *ctx, void *dev_partials, size_t count);   // kpal_multi.hip: {double sum, uint64 count} pairs added over the ranks, in place

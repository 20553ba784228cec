// kpal_count.hip -- the counting front end of the C-ABI (begin / feed / feed_device / feed_fasta / records / finish),
// strategy choice, piece sizes, H2D staging, and the launchers of the LDS-direct, global-atomic and round-1
// partition pipelines.  The quad record pipelines live in kpal_quads.hip / kpal_quads2.hip.
#include "kpal_host.hpp"

#include "count_kernels.hpp"
#include "partition_kernels.hpp"
#include "chunk_kernels.hpp"
#include "fasta_kernels.hpp"
#include "fasta_host.hpp"

#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

static_assert(sizeof(ChunkPool) <= sizeof(kpal_ctx::chunk_pool_sent), "kpal_ctx::chunk_pool_sent holds a ChunkPool");

// ----------------------------------------------------------------------------------------------
// counting
// ----------------------------------------------------------------------------------------------
KPAL_API int kpal_count_begin(kpal_ctx *ctx, int k)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);
    if (ctx->merged && (ctx->merged_bins != (1ULL << (2 * k)) || ctx->merged == ctx->table.p)) {
        // another k, or the merged table WAS the count table (serial reduce) which is about to be zeroed / reallocated
        ctx->merged = nullptr;
        ctx->merged_bins = 0;
    }
    ctx->k = k;
    ctx->bins = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->table, ctx->bins * sizeof(int64_t)));
    // k >= 13: the zeroing of the table (8.6 GB at k = 15) is deferred -- a first piece on the two-level quad pipeline writes the whole
    // table in its finalisation and never needs it (kpal_quads2.hip, FRESH); whatever else touches the table first zeroes it (table_ready)
    static const bool allow_fresh = [] { const char *e = getenv("KPAL_FRESH"); return !e || atoi(e) != 0; }();
    ctx->table_zero_pending = allow_fresh && k >= 13 && (ctx->strategy == KPAL_STRATEGY_AUTO || ctx->strategy == KPAL_STRATEGY_PARTITION2_QUADS);
    ctx->finalize_fresh = false;
    if (!ctx->table_zero_pending) HIPCHK(hipMemsetAsync(ctx->table.p, 0, ctx->bins * sizeof(int64_t), ctx->stream));
    if (ctx->chunk_error_word) HIPCHK(hipMemsetAsync(ctx->chunk_error_word, 0, sizeof(uint32_t), ctx->stream));
    if (ctx->quad_error_word) HIPCHK(hipMemsetAsync(ctx->quad_error_word, 0, sizeof(uint32_t), ctx->stream));
    ctx->chunk_error_armed = false;
    ctx->finalize_pending = false;          // (staged forms of an abandoned count)
    ctx->cached_steps1 = ctx->cached_steps2 = 0;
    ctx->sample_hot_rows = false;           // (the verdict of a sample of THIS count only: kpal_quads2.hip reads it for cached tile sizes)
    ctx->plan_strategy = ctx->plan_steps1 = ctx->plan_steps2 = 0;
    ctx->counting = true;
    return KPAL_OK;
}

KPAL_API int kpal_count_set_strategy(kpal_ctx *ctx, int strategy)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (strategy < KPAL_STRATEGY_AUTO || strategy > KPAL_STRATEGY_PARTITION2_QUADS)
        return set_err(KPAL_E_INVALID, "unknown strategy %d", strategy);
    ctx->strategy = strategy;
    return KPAL_OK;
}

static int resolve_strategy(kpal_ctx *ctx, int *out)
{
    int s = ctx->strategy;
    const int k = ctx->k;
    if (s == KPAL_STRATEGY_AUTO)
        s = k <= 7 ? KPAL_STRATEGY_LDS_DIRECT : (k <= 12 ? KPAL_STRATEGY_PARTITION_QUADS : KPAL_STRATEGY_PARTITION2_QUADS);
    if (s == KPAL_STRATEGY_LDS_DIRECT && k > 7) return set_err(KPAL_E_INVALID, "LDS-direct strategy needs k <= 7 (k=%d)", k);
    if ((s == KPAL_STRATEGY_PARTITION || s == KPAL_STRATEGY_PARTITION_CHUNKED || s == KPAL_STRATEGY_PARTITION_QUADS) && (k < 8 || k > 12))
        return set_err(KPAL_E_INVALID, "partition strategy needs 8 <= k <= 12 (k=%d)", k);
    if ((s == KPAL_STRATEGY_PARTITION2 || s == KPAL_STRATEGY_PARTITION2_QUADS) && (k < 13 || k > 16))
        return set_err(KPAL_E_INVALID, "two-level partition strategy needs 13 <= k <= 16 (k=%d)", k);
    *out = s;
    return KPAL_OK;
}

// Span for emitting the k-mers that end in [addr, addr+n), with `halo` readable bytes of the
// same feed to the left of addr.
static Span make_span(const uint8_t *addr, size_t n, size_t halo)
{
    const uintptr_t first = (uintptr_t)addr - halo;
    const uintptr_t base = first & ~(uintptr_t)15;
    Span s;
    s.base = reinterpret_cast<const uint4 *>(base);
    s.lo = first - base;
    s.emit_from = s.lo + halo;
    s.hi = s.emit_from + n;
    s.nchunks = (s.hi + 15) / 16;
    return s;
}

static int launch_global_atomic(kpal_ctx *ctx, const Span &s)
{
    const uint64_t steps = (s.nchunks + 63) / 64;
    const uint64_t max_waves = (uint64_t)ctx->num_cu * 8 * 4;  // 8 blocks of 4 waves per CU
    const uint64_t spw = std::max<uint64_t>(1, (steps + max_waves - 1) / max_waves);
    const uint64_t waves = (steps + spw - 1) / spw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    DISPATCH_K_1_16(ctx->k, LAUNCH(ctx, "count_global_atomic", (count_global_atomic_kernel<K>), dim3(grid), dim3(256), s, spw, table));
    return KPAL_OK;
}

static int launch_lds_direct(kpal_ctx *ctx, const Span &s)
{
    const uint64_t steps = (s.nchunks + 63) / 64;
    const uint64_t max_waves = (uint64_t)ctx->num_cu * 2 * 8;  // 2 blocks of 8 waves per CU
    const uint64_t spw = std::max<uint64_t>(1, (steps + max_waves - 1) / max_waves);
    const uint64_t waves = (steps + spw - 1) / spw;
    const unsigned grid = (unsigned)((waves + 7) / 8);
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    DISPATCH_K_1_7(ctx->k, LAUNCH(ctx, "count_lds_direct", (count_lds_direct_kernel<K>), dim3(grid), dim3(512), s, spw, table));
    return KPAL_OK;
}

// One-level partition, k = 8..12 (partition_kernels.hpp: A1, A2, A3, B).
static int launch_partition(kpal_ctx *ctx, const Span &s)
{
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (total_steps == 0) return KPAL_OK;
    // steps per block: a multiple of 24 (8 waves x 3 steps per tile), ~4 blocks per CU
    const uint64_t want_blocks = (uint64_t)ctx->num_cu * 4;
    uint64_t spb = (total_steps + want_blocks - 1) / want_blocks;
    spb = (spb + kStepsPerBlockQuantum - 1) / kStepsPerBlockQuantum * kStepsPerBlockQuantum;
    const uint32_t G = (uint32_t)((total_steps + spb - 1) / spb);
    const uint64_t max_keys = s.nchunks * 16;
    CHK(ensure(ctx, ctx->keys, max_keys * sizeof(uint16_t) + 64));
    CHK(ensure(ctx, ctx->cntmat, (size_t)kNumBuckets * G * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->offs, (size_t)kNumBuckets * G * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->bucket_start, (size_t)(2 * kNumBuckets + 2) * sizeof(uint64_t)));
    CHK(ensure(ctx, ctx->slice_start, (size_t)(kNumBuckets + 1) * sizeof(uint32_t)));
    uint32_t *cntmat = (uint32_t *)ctx->cntmat.p;
    uint32_t *offs = (uint32_t *)ctx->offs.p;
    uint64_t *bstart = (uint64_t *)ctx->bucket_start.p;
    uint64_t *btotal = bstart + kNumBuckets + 1;
    uint32_t *sstart = (uint32_t *)ctx->slice_start.p;
    uint16_t *keys = (uint16_t *)ctx->keys.p;
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    const uint64_t *no_base = nullptr;
    DISPATCH_K_8_12(ctx->k, {
        LAUNCH(ctx, "part_count", (part_count_kernel<K>), dim3(G), dim3(kScatterThreads), s, spb, cntmat);
        LAUNCH(ctx, "part_rowscan", part_rowscan_kernel, dim3(kNumBuckets), dim3(256), (const uint32_t *)cntmat, G, offs, btotal);
        LAUNCH(ctx, "part_bucketscan", part_bucketscan_kernel, dim3(1), dim3(kNumBuckets), (const uint64_t *)btotal,
               (uint32_t)kNumBuckets, no_base, bstart, sstart);
        LAUNCH(ctx, "part_scatter", (part_scatter_kernel<K>), dim3(G), dim3(kScatterThreads), s, spb,
               (const uint32_t *)offs, (const uint64_t *)bstart, keys);
        // one workgroup per bucket (exclusive table slice -> plain read-modify-write merge, measured
        // fastest); oversized buckets of skewed input are cut into slices by the bucket scan
        LAUNCH(ctx, "part_hist", (part_hist_kernel<PartCfg<K>::kKeyBits>), dim3(kHistGridX), dim3(1024),
               (const uint16_t *)keys, (const uint64_t *)bstart, (const uint32_t *)sstart, table);
    });
    return KPAL_OK;
}

// Workspace of one chunked scatter + histogram over Y coarse buckets (Y = 1: one-level path):
// pool of 8 KiB key chunks, table rows, overflow lists, per-coarse-bucket meta words and the device
// copy of the pool descriptor.  meta words per coarse bucket y: nlist[512] ovf_n[512] (all y first,
// so one memset clears them), then ovf_count[Y] error, then ostart[Y][513] ocur[Y][512]
// slice_start[Y][513], then the descriptor.
struct ChunkLaunch {
    ChunkPool p;
    ChunkPool *dpool;
    uint32_t *ostart, *ocur, *sstart;
    uint32_t Y;
};

static int chunk_prepare(kpal_ctx *ctx, uint32_t Y, uint32_t G, uint64_t R, ChunkLaunch &cl)
{
    const uint64_t per_y = (uint64_t)G * R;
    if (per_y >= (1ull << kChunkIdBits)) return set_err(KPAL_E_INVALID, "chunked partition: batch too large");
    const uint64_t cap = per_y * Y;
    CHK(ensure(ctx, ctx->keys, cap * kChunkKeys * sizeof(uint16_t)));
    CHK(ensure(ctx, ctx->chunk_table, (size_t)Y * kNumBuckets * G * kChunkRow * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->chunk_ovf, cap * sizeof(uint2)));
    CHK(ensure(ctx, ctx->chunk_sorted, cap * sizeof(uint32_t)));
    const size_t clear_words = (size_t)Y * 2 * kNumBuckets + Y;   // nlist, ovf_n, ovf_count
    const size_t pool_words = (sizeof(ChunkPool) + 3) / 4 + 8;
    const size_t meta_words = clear_words + 1 + (size_t)Y * (2 * (kNumBuckets + 1) + kNumBuckets) + 4 + pool_words;
    const bool fresh = ctx->chunk_meta.cap < meta_words * sizeof(uint32_t);
    CHK(ensure(ctx, ctx->chunk_meta, meta_words * sizeof(uint32_t)));
    uint32_t *meta = (uint32_t *)ctx->chunk_meta.p;
    if (fresh || Y != ctx->chunk_meta_y) {
        HIPCHK(hipMemsetAsync(meta, 0, meta_words * sizeof(uint32_t), ctx->stream));
        ctx->chunk_meta_y = Y;
        ctx->chunk_pool_dev = nullptr;
    }
    ChunkPool &p = cl.p;
    memset(&p, 0, sizeof(p));   // padding too: the descriptor is compared bytewise below
    p.keys = (uint16_t *)ctx->keys.p;
    p.per_block = (uint32_t)R;
    p.groups = G;
    p.table = (uint32_t *)ctx->chunk_table.p;
    p.nlist = meta;
    p.ovf_n = meta + (size_t)Y * kNumBuckets;
    p.ovf_count = meta + (size_t)Y * 2 * kNumBuckets;
    p.error = meta + clear_words;
    p.ovf = (uint2 *)ctx->chunk_ovf.p;
    cl.ostart = meta + clear_words + 1;
    cl.ocur = cl.ostart + (size_t)Y * (kNumBuckets + 1);
    cl.sstart = cl.ocur + (size_t)Y * kNumBuckets;
    cl.dpool = (ChunkPool *)(((uintptr_t)(cl.sstart + (size_t)Y * (kNumBuckets + 1)) + 15) & ~(uintptr_t)15);
    cl.Y = Y;
    ctx->chunk_error_word = p.error;
    // the device copy changes only when a buffer was reallocated or the geometry changed: a
    // synchronous copy then -- an asynchronous one would read this stack frame after it is gone
    if (memcmp(&p, ctx->chunk_pool_sent, sizeof(ChunkPool)) != 0 || (void *)cl.dpool != ctx->chunk_pool_dev) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(cl.dpool, &p, sizeof(ChunkPool), hipMemcpyHostToDevice));
        memcpy(ctx->chunk_pool_sent, &p, sizeof(ChunkPool));
        ctx->chunk_pool_dev = cl.dpool;
    }
    ctx->chunk_error_armed = true;
    // per batch: counts restart at 0; the error word is sticky until count_finish
    HIPCHK(hipMemsetAsync(meta, 0, clear_words * sizeof(uint32_t), ctx->stream));
    return KPAL_OK;
}

// The kernels after the scatter: slice plan, overflow grouping, histogram + merge.
template <int KB>
static int chunk_histogram(kpal_ctx *ctx, const ChunkLaunch &cl)
{
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    LAUNCH(ctx, "chunk_plan", chunk_plan_kernel, dim3(cl.Y), dim3(kNumBuckets), (const uint32_t *)cl.p.nlist,
           (const uint32_t *)cl.p.ovf_n, cl.ostart, cl.ocur, cl.sstart);
    LAUNCH(ctx, "chunk_list", chunk_list_kernel, dim3(64, cl.Y), dim3(256), cl.p, (const uint32_t *)cl.ostart, cl.ocur,
           (uint32_t *)ctx->chunk_sorted.p);
    LAUNCH(ctx, "chunk_hist", (chunk_hist_kernel<KB>), dim3(kHistGridX, cl.Y), dim3(1024), cl.p, (const uint32_t *)cl.ostart,
           (const uint32_t *)ctx->chunk_sorted.p, (const uint32_t *)cl.sstart, table);
    return KPAL_OK;
}

// Chunked one-level partition, k = 8..12 (chunk_kernels.hpp): scatter into per-workgroup 8 KiB
// chunks, record them in (bucket, workgroup) table rows, histogram every bucket's chunks.
static int launch_partition_chunked(kpal_ctx *ctx, const Span &s)
{
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (total_steps == 0) return KPAL_OK;
    // one round of two resident workgroups per CU: every workgroup leaves a partly filled and an
    // unused chunk per bucket behind, so fewer, longer workgroups than the exact-offset path
    const uint64_t want_blocks = (uint64_t)ctx->num_cu * 2;
    uint64_t spb = (total_steps + want_blocks - 1) / want_blocks;
    spb = (spb + kStepsPerBlockQuantum - 1) / kStepsPerBlockQuantum * kStepsPerBlockQuantum;
    const uint32_t G = (uint32_t)((total_steps + spb - 1) / spb);
    // chunks per workgroup, worst case: spb*1024/4096 full ones + a partly filled and a
    // pre-assigned next one per bucket (+ slack)
    // (the stride of the ranges is harmless except at exact powers of two: R = 2048 -> 16 MiB costs 10 %)
    const uint64_t R = spb * 1024 / kChunkKeys + 2 * kNumBuckets + 64;
    ChunkLaunch cl;
    CHK(chunk_prepare(ctx, 1, G, R, cl));
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    DISPATCH_K_8_12(ctx->k, {
        LAUNCH(ctx, "chunk_scatter", (chunk_scatter_kernel<K>), dim3(G), dim3(kScatterThreads), s, spb, (const ChunkPool *)cl.dpool,
               cl.p.keys, cl.p.per_block, table);
        CHK(chunk_histogram<PartCfg<K>::kKeyBits>(ctx, cl));
    });
    return KPAL_OK;
}



// Two-level partition, k = 13..16: coarse count/scan/scatter into 24-bit residuals, then the
// one-level pipeline on every coarse bucket's residual stream (2-D launches over coarse buckets).
static int launch_partition2(kpal_ctx *ctx, const Span &s)
{
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (total_steps == 0) return KPAL_OK;
    const int NB1 = 1 << (2 * ctx->k - kResidualBits);
    const uint64_t want_blocks = (uint64_t)ctx->num_cu * 8;   // measured: coarse_count 8 % faster than with 4 per CU, coarse_scatter indifferent
    uint64_t spb = (total_steps + want_blocks - 1) / want_blocks;
    spb = (spb + 7) / 8 * 8;   // 8 waves, one step per wave per tile
    const uint32_t G1 = (uint32_t)((total_steps + spb - 1) / spb);
    const uint64_t max_keys = s.nchunks * 16;
    if (ensure(ctx, ctx->residuals, max_keys * sizeof(uint32_t) + 64) != KPAL_OK ||
        (ctx->level2_mode == 0 && ensure(ctx, ctx->keys, max_keys * sizeof(uint16_t) + 64) != KPAL_OK)) {
        if (max_keys <= ((uint64_t)1 << 30)) return KPAL_E_NOMEM;
        return kSplitBatch;   // not enough HBM for a batch of this size: retry with half
    }
    CHK(ensure(ctx, ctx->cnt1, (size_t)NB1 * G1 * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->offs1, (size_t)NB1 * G1 * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->start1, (size_t)(2 * NB1 + 2) * sizeof(uint64_t)));
    uint32_t *res = (uint32_t *)ctx->residuals.p;
    uint32_t *cnt1 = (uint32_t *)ctx->cnt1.p;
    uint32_t *offs1 = (uint32_t *)ctx->offs1.p;
    uint64_t *start1 = (uint64_t *)ctx->start1.p;
    uint64_t *total1 = start1 + NB1 + 1;
    uint16_t *keys = (uint16_t *)ctx->keys.p;
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    const uint64_t *no_base = nullptr;
    DISPATCH_K_13_16(ctx->k, {
        LAUNCH(ctx, "coarse_count", (coarse_count_kernel<K>), dim3(G1), dim3(kCoarseThreads), s, spb, cnt1);
        LAUNCH(ctx, "part_rowscan", part_rowscan_kernel, dim3(NB1), dim3(256), (const uint32_t *)cnt1, G1, offs1, total1);
        LAUNCH(ctx, "part_bucketscan", part_bucketscan_kernel, dim3(1), dim3(kNumBuckets), (const uint64_t *)total1,
               (uint32_t)NB1, no_base, start1, (uint32_t *)nullptr);
    });
    // coarse bucket sizes: they size the level-2 launches and guard the 32-bit in-bucket offsets
    // (one small D2H + sync per batch)
    std::vector<uint64_t> h1((size_t)NB1 + 1);
    HIPCHK(hipMemcpyAsync(h1.data(), start1, h1.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t maxn = 0;
    for (int c = 0; c < NB1; ++c) maxn = std::max(maxn, h1[c + 1] - h1[c]);
    if (maxn == 0) return KPAL_OK;
    if (maxn > ctx->split_above && s.nchunks > 64) return kSplitBatch;   // skewed batch: the caller halves it
    DISPATCH_K_13_16(ctx->k, {
        LAUNCH(ctx, "coarse_scatter", (coarse_scatter_kernel<K>), dim3(G1), dim3(kCoarseThreads), s, spb,
               (const uint32_t *)offs1, (const uint64_t *)start1, res);
    });
    if (ctx->level2_mode != 0) {
        // level 2 as a chunked scatter (chunk_kernels.hpp): no counting pass over the residuals.
        // Two resident workgroups per CU in total; every workgroup leaves ~1000 unused 8 KiB chunks.
        const bool lines = ctx->level2_mode == 2;   // aligned-line staging: one 1024-thread workgroup per CU
        // workgroups per coarse bucket: 8 per CU in total when the coarse buckets are equal (fewer, longer
        // workgroups are 6 % faster); unequal buckets (AT-rich input) leave most workgroups of the small
        // ones empty, so the granularity is doubled (AT-rich 1 GiB: 3.1 -> 2.2 ms)
        const uint64_t total1 = h1[NB1] - h1[0];
        const bool unequal = (double)maxn * NB1 > 1.25 * (double)total1;
        uint64_t g2t = std::max<uint64_t>(2, (uint64_t)ctx->num_cu * (lines ? (unequal ? 16 : 8) : 2) / NB1);
        g2t = std::min<uint64_t>(g2t, std::max<uint64_t>(2, 4096 / NB1));   // every workgroup reserves ~1000 chunks (9 MB) of pool address space
        const uint64_t quantum = lines ? (uint64_t)kKeysPerBlockQuantum : (uint64_t)kScatterWaves * kScatterSteps * kMacroKeys;
        uint64_t kpb2 = 0, R2 = 0;
        uint32_t G2c = 0;
        for (;; g2t = (g2t + 1) / 2) {   // few coarse buckets (k = 13): keep a coarse bucket's chunk ids below 2^20
            kpb2 = (maxn + g2t - 1) / g2t;
            kpb2 = (kpb2 + quantum - 1) / quantum * quantum;
            if (kpb2 > 0xFFFFFFFFull) return set_err(KPAL_E_INVALID, "two-level partition: batch too large");
            G2c = (uint32_t)((maxn + kpb2 - 1) / kpb2);
            R2 = kpb2 / kChunkKeys + 2 * kNumBuckets + 64;
            if ((uint64_t)G2c * R2 < (1ull << kChunkIdBits) || g2t <= 2) break;
        }
        if ((uint64_t)G2c * R2 >= (1ull << kChunkIdBits) && max_keys > ((uint64_t)1 << 30)) return kSplitBatch;   // one coarse bucket holds (almost) everything
        ChunkLaunch cl;
        const int rc = chunk_prepare(ctx, (uint32_t)NB1, G2c, R2, cl);
        if (rc == KPAL_E_NOMEM && max_keys > ((uint64_t)1 << 30)) return kSplitBatch;   // retry with half the batch
        if (rc != KPAL_OK) return rc;
        if (lines)
            LAUNCH(ctx, "chunk_key_lines", chunk_key_lines_kernel, dim3(G2c, NB1), dim3(kLineThreads), (const uint32_t *)res,
                   (const uint64_t *)start1, (uint32_t)kpb2, (const ChunkPool *)cl.dpool, cl.p.keys, cl.p.per_block, table);
        else
            LAUNCH(ctx, "chunk_key_scatter", chunk_key_scatter_kernel, dim3(G2c, NB1), dim3(kScatterThreads), (const uint32_t *)res,
                   (const uint64_t *)start1, (uint32_t)kpb2, (const ChunkPool *)cl.dpool, cl.p.keys, cl.p.per_block, table);
        return chunk_histogram<kResKeyBits>(ctx, cl);
    }
    const uint64_t g2_target = std::max<uint64_t>(8, (uint64_t)ctx->num_cu * 8 / NB1);
    uint64_t kpb = (maxn + g2_target - 1) / g2_target;
    kpb = (kpb + kKeysPerBlockQuantum - 1) / kKeysPerBlockQuantum * kKeysPerBlockQuantum;
    if (kpb > 0xFFFFFFFFull) return set_err(KPAL_E_INVALID, "two-level partition: batch too large");
    const uint32_t G2 = (uint32_t)((maxn + kpb - 1) / kpb);
    const size_t rows2 = (size_t)NB1 * kNumBuckets;
    CHK(ensure(ctx, ctx->cntmat, rows2 * G2 * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->offs, rows2 * G2 * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->bucket_start, ((size_t)NB1 * (kNumBuckets + 1) + rows2) * sizeof(uint64_t)));
    uint32_t *cntmat2 = (uint32_t *)ctx->cntmat.p;
    uint32_t *offs2 = (uint32_t *)ctx->offs.p;
    uint64_t *bstart2 = (uint64_t *)ctx->bucket_start.p;
    uint64_t *total2 = bstart2 + (size_t)NB1 * (kNumBuckets + 1);
    CHK(ensure(ctx, ctx->slice_start, (size_t)NB1 * (kNumBuckets + 1) * sizeof(uint32_t)));
    uint32_t *sstart2 = (uint32_t *)ctx->slice_start.p;
    LAUNCH(ctx, "key_count", key_count_kernel, dim3(G2, NB1), dim3(kScatterThreads), (const uint32_t *)res,
           (const uint64_t *)start1, (uint32_t)kpb, cntmat2);
    LAUNCH(ctx, "part_rowscan", part_rowscan_kernel, dim3(kNumBuckets, NB1), dim3(256), (const uint32_t *)cntmat2, G2, offs2, total2);
    LAUNCH(ctx, "part_bucketscan", part_bucketscan_kernel, dim3(NB1), dim3(kNumBuckets), (const uint64_t *)total2,
           (uint32_t)kNumBuckets, (const uint64_t *)start1, bstart2, sstart2);
    LAUNCH(ctx, "key_scatter", key_scatter_kernel, dim3(G2, NB1), dim3(kLineThreads), (const uint32_t *)res,
           (const uint64_t *)start1, (uint32_t)kpb, (const uint32_t *)offs2, (const uint64_t *)bstart2, keys);
    LAUNCH(ctx, "part_hist", (part_hist_kernel<kResKeyBits>), dim3(kHistGridX, NB1), dim3(1024),
           (const uint16_t *)keys, (const uint64_t *)bstart2, (const uint32_t *)sstart2, table);
    return KPAL_OK;
}

static int count_device_range(kpal_ctx *ctx, const uint8_t *addr, size_t n, size_t halo);

// A piece whose record pool would be too large (kSplitBatch): as two halves.
static int count_device_halves(kpal_ctx *ctx, const uint8_t *addr, size_t n, size_t halo)
{
    const size_t half = (n / 2 + 15) & ~(size_t)15;
    CHK(count_device_range(ctx, addr, half, halo));
    if (n > half) CHK(count_device_range(ctx, addr + half, n - half, halo + half));
    return KPAL_OK;
}

// Count all k-mers ending in [addr, addr+n) of a device buffer; `halo` bytes left of addr are
// readable and belong to the same feed.
static int count_device_range(kpal_ctx *ctx, const uint8_t *addr, size_t n, size_t halo)
{
    int strat = 0;
    CHK(resolve_strategy(ctx, &strat));
    // tiny feeds (single records, short reads lists): the partition pipelines cost a fixed
    // 0.1 - 0.5 ms (launches, one merge of the whole table); a quarter million atomics do not
    if (ctx->strategy == KPAL_STRATEGY_AUTO && ctx->k >= 8 && n <= ((size_t)1 << 18)) strat = KPAL_STRATEGY_GLOBAL_ATOMIC;
    // the quad pipeline pays a fixed histogram stage (one 128 KiB workgroup per bucket): medium feeds take the chunked one
    else if (ctx->strategy == KPAL_STRATEGY_AUTO && strat == KPAL_STRATEGY_PARTITION_QUADS && n < ((size_t)32 << 20)) strat = KPAL_STRATEGY_PARTITION_CHUNKED;
    // The two-level quad pipeline pays per FEED for the whole table -- its forms are staged (4 bytes per entry) and the finalisation
    // reads them and the table and writes the table -- where the round-1 two-level pipeline adds into the table with atomics and
    // pays for the table once per count (memset, Profile.balance).  Measured at the end of round 4 (same box, count + balance,
    // 68 MB .. 15 GB of reads): a feed that is the FIRST piece of a count and a whole device buffer (FRESH: no memset, the table
    // not read, the balance fused) is faster through the quads at every size from 64 MiB up -- k = 15: 4.7 vs 9.5 ms on 68 MB, 8.6
    // vs 23.7 on 4.2 GB; k = 16: 19.3 vs 30.5 and 24.2 vs 54.6, and 34.9 vs 92.8 ms on the 15.1 GB of BASELINE's reads, which the
    // earlier rule (feed >= 4 bytes per table entry, from round-2 timings of both pipelines) still sent to the old pipeline --
    // except that a count that is never balanced loses ~7 % below an eighth of a byte per entry (k = 16).  Any other feed (a later
    // piece, a piece of a host feed, a FASTA chunk) takes the quads once it holds about three bytes per table entry (the per-feed
    // crossover computed from the same timings: 2.8 B per entry at k = 15, 1.5 at k = 16).
    else if (ctx->strategy == KPAL_STRATEGY_AUTO && strat == KPAL_STRATEGY_PARTITION2_QUADS) {
        const bool fresh_piece = ctx->table_zero_pending && ctx->fresh_feed && halo == 0;
        const size_t need = fresh_piece ? (size_t)(ctx->bins / 8) : (size_t)(ctx->bins * 3);
        if (n < std::max<size_t>((size_t)64 << 20, need)) strat = KPAL_STRATEGY_PARTITION2;
    }
    const size_t km1 = (size_t)ctx->k - 1;
    size_t piece = n;
    if (strat == KPAL_STRATEGY_PARTITION) piece = ctx->batch_bytes;
    else if (strat == KPAL_STRATEGY_PARTITION_CHUNKED) {
        // as large as the 20-bit chunk ids allow (G workgroups x R chunks each < 2^20, R = steps/4 + 1088 in
        // launch_partition_chunked): every piece ends with a merge of the whole table and four launches.
        // 1.86 GiB on 256 CUs; KPAL_BATCH_BYTES lowers it.
        const uint64_t G = (uint64_t)ctx->num_cu * 2;
        const uint64_t r_max = ((1ull << kChunkIdBits) - 1) / G;
        const uint64_t fixed = 2 * kNumBuckets + 64;
        uint64_t spb_max = r_max > fixed + 64 ? (r_max - fixed) * (kChunkKeys / 1024) : 64;
        spb_max = spb_max > 3 * kStepsPerBlockQuantum ? spb_max - 2 * kStepsPerBlockQuantum : spb_max;   // margin: the halo may add a step
        spb_max = spb_max / kStepsPerBlockQuantum * kStepsPerBlockQuantum;
        const size_t cap = (size_t)(spb_max * G * 1024);
        piece = ctx->batch_bytes_set ? std::min<size_t>(ctx->batch_bytes, cap) : cap;
    }
    else if (strat == KPAL_STRATEGY_PARTITION_QUADS) {
        // the record pool takes 4/3 of the input bytes (up to 8 x that for heavily skewed input, whose tiles are
        // smaller): pieces of up to 16 GiB (KPAL_BATCH_BYTES lowers it)
        piece = ctx->batch_bytes_set ? std::min<size_t>(ctx->batch_bytes, (size_t)16 << 30) : (size_t)16 << 30;
    }
    else if (strat == KPAL_STRATEGY_PARTITION2_QUADS) {
        // two record pools of ~4/3 of the input bytes each: pieces of up to 16 GiB
        piece = ctx->batch_bytes_set ? std::min<size_t>(ctx->batch_bytes * 16, (size_t)16 << 30) : (size_t)16 << 30;
    }
    else if (strat == KPAL_STRATEGY_PARTITION2) {
        // every batch ends with a read-modify-write of the whole 4^k table (0.5 - 32 GiB): few, large
        // batches.  In-bucket offsets are 32-bit: below 2^32 keys per batch always safe (k = 13 has
        // only four coarse buckets); larger batches are checked per coarse bucket and halved if needed.
        piece = ctx->k == 13 ? std::min<size_t>(ctx->batch_bytes * 4, (size_t)0xF0000000u)
                             : std::min<size_t>(ctx->batch_bytes * 16, (size_t)16 << 30);
    }
    else if (strat == KPAL_STRATEGY_LDS_DIRECT) piece = (size_t)1 << 31;
    piece &= ~(size_t)15;
    if (piece == 0) piece = 16;
    for (size_t off = 0; off < n; off += piece) {
        const size_t len = std::min(piece, n - off);
        // FRESH: the first piece of a count, a whole device feed on the two-level quad pipeline, leaves the table unzeroed
        const bool fresh = ctx->table_zero_pending && ctx->fresh_feed && strat == KPAL_STRATEGY_PARTITION2_QUADS && off == 0 && len == n && halo == 0;
        if (!fresh) CHK(table_ready(ctx));   // zeros materialised; the staged forms of the previous piece added before their buffer is reused
        const size_t h = std::min(km1, halo + off);
        const Span s = make_span(addr + off, len, h);
        if (strat != KPAL_STRATEGY_PARTITION_QUADS && strat != KPAL_STRATEGY_PARTITION2_QUADS) {
            ctx->plan_strategy = strat;
            ctx->plan_steps1 = ctx->plan_steps2 = 0;
        }
        if (strat == KPAL_STRATEGY_PARTITION || strat == KPAL_STRATEGY_PARTITION_CHUNKED || strat == KPAL_STRATEGY_PARTITION2) ++ctx->stat_chunked_pieces;
        if (strat == KPAL_STRATEGY_GLOBAL_ATOMIC) CHK(launch_global_atomic(ctx, s));
        else if (strat == KPAL_STRATEGY_LDS_DIRECT) CHK(launch_lds_direct(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION) CHK(launch_partition(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION_CHUNKED) CHK(launch_partition_chunked(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION2_QUADS) {
            const int rc = launch_partition2_quads(ctx, s, fresh);
            if (rc == KPAL_OK) {
                ++ctx->stat_quad_pieces;
                if (fresh) ++ctx->stat_fresh_pieces;
            }
            if (rc == kSplitBatch) ++ctx->stat_split_pieces;
            if (rc == kQuadsUseChunked || rc == kSplitBatch) CHK(table_ready(ctx));   // (nothing was launched: the other paths need the zeros)
            if (rc == kQuadsUseChunked) {   // (AUTO only) this piece through the round-1 two-level pipeline
                ctx->strategy = KPAL_STRATEGY_PARTITION2;
                const int r2 = count_device_range(ctx, addr + off, len, halo + off);
                ctx->strategy = KPAL_STRATEGY_AUTO;
                if (r2 != KPAL_OK) return r2;
            } else if (rc == kSplitBatch) {
                CHK(count_device_halves(ctx, addr + off, len, halo + off));
            } else if (rc != KPAL_OK) {
                return rc;
            }
        }
        else if (strat == KPAL_STRATEGY_PARTITION_QUADS) {
            const int rc = launch_partition_quads(ctx, s);
            if (rc == KPAL_OK) ++ctx->stat_quad_pieces;
            if (rc == kSplitBatch) ++ctx->stat_split_pieces;
            if (rc == kQuadsUseChunked) {   // (AUTO only) this piece through the chunked pipeline, in its own piece size
                ctx->strategy = KPAL_STRATEGY_PARTITION_CHUNKED;
                const int r2 = count_device_range(ctx, addr + off, len, halo + off);
                ctx->strategy = KPAL_STRATEGY_AUTO;
                if (r2 != KPAL_OK) return r2;
            } else if (rc == kSplitBatch) {
                CHK(count_device_halves(ctx, addr + off, len, halo + off));
            } else if (rc != KPAL_OK) {
                return rc;
            }
        }
        else {
            const int rc = launch_partition2(ctx, s);
            if (rc == kSplitBatch) {   // rare: process this piece as two halves
                const size_t half = (len / 2 + 15) & ~(size_t)15;
                const size_t saved = ctx->batch_bytes;
                ctx->batch_bytes = std::max<size_t>(half / (ctx->k == 13 ? 4 : 16), 16);
                int r2 = count_device_range(ctx, addr + off, half, halo + off);
                if (r2 == KPAL_OK && len > half) r2 = count_device_range(ctx, addr + off + half, len - half, halo + off + half);
                ctx->batch_bytes = saved;
                if (r2 != KPAL_OK) return r2;
            } else if (rc != KPAL_OK) {
                return rc;
            }
        }
    }
    return KPAL_OK;
}

KPAL_API int kpal_count_feed_device(kpal_ctx *ctx, const void *dev_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed_device before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!dev_buf) return set_err(KPAL_E_INVALID, "dev_buf is NULL");
    ctx->fresh_feed = true;
    int rc = count_device_range(ctx, (const uint8_t *)dev_buf, nbytes, 0);
    ctx->fresh_feed = false;
    // a FRESH piece whose lists overflowed is counted again from dev_buf: settled here, so that the buffer is the caller's again
    // (stream-ordered, as with every other pipeline) when this call returns
    if (rc == KPAL_OK) rc = quad2_resolve_fresh(ctx);
    return rc;
}

// Host copy into a pinned staging buffer on several cores (host_pool.hpp): one core's memcpy (~9 GB/s) is what limits a
// pageable-memory feed otherwise, the PCIe link takes six times that.  Pieces below 4 MiB are not split.
static void staged_memcpy(void *dst, const void *src, size_t n)
{
    HostPool &pool = HostPool::instance();
    const int parts = (int)std::min<size_t>((size_t)pool.size(), n / ((size_t)4 << 20));
    if (parts <= 1) {
        memcpy(dst, src, n);
        return;
    }
    const size_t part = (((n + (size_t)parts - 1) / (size_t)parts) + 4095) & ~(size_t)4095;
    pool.run(parts, [=](int i) {
        const size_t off = (size_t)i * part;
        if (off < n) memcpy((uint8_t *)dst + off, (const uint8_t *)src + off, std::min(part, n - off));
    });
}

// Page-locked host memory on the NUMA node the GPU is attached to: the calling thread's memory policy is set to "prefer that node"
// around the allocation (raw set_mempolicy: libnuma is not a dependency) and hipHostMallocNumaUser lets the runtime honour it.  A
// staging buffer on the other socket puts the inter-socket link into every DMA.  The policy the thread had is saved and put back
// (a policy that cannot be read is left alone: plain allocation).  Any failure falls back to a plain allocation.
int host_alloc_near_gpu(kpal_ctx *ctx, void **out, size_t nbytes)
{
    *out = nullptr;
    const int node = ctx->numa_node;
    hipError_t e = hipErrorUnknown;
    if (node >= 0 && node < 64) {
        unsigned long mask = 1ul << node;
        const long kPreferred = 1;   // MPOL_PREFERRED
        // the thread's own policy (numactl --membind / --interleave, the application's set_mempolicy) is put back afterwards
        int old_mode = 0;
        unsigned long old_mask[16] = {};   // 1024 nodes
        const bool saved = syscall(SYS_get_mempolicy, &old_mode, old_mask, (unsigned long)(sizeof(old_mask) * 8), nullptr, 0ul) == 0;
        if (saved && syscall(SYS_set_mempolicy, kPreferred, &mask, 65ul) == 0) {
            e = hipHostMalloc(out, nbytes, hipHostMallocNumaUser);
            (void)syscall(SYS_set_mempolicy, (long)old_mode, old_mask, (unsigned long)(sizeof(old_mask) * 8 + 1));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                *out = nullptr;
            }
        }
    }
    if (!*out) e = hipHostMalloc(out, nbytes, hipHostMallocDefault);
    if (e != hipSuccess) return set_err(KPAL_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", nbytes, hipGetErrorString(e));
    return KPAL_OK;
}

static int ensure_pinned(kpal_ctx *ctx)
{
    for (int i = 0; i < 2; ++i)
        if (!ctx->pinned[i]) CHK(host_alloc_near_gpu(ctx, &ctx->pinned[i], kpal_ctx::kStage + kpal_ctx::kStagePad));
    return KPAL_OK;
}

// pinned_source: host_buf came from kpal_host_alloc -- the DMA engine reads it in place, no staging copy; the call returns
// when the last copy has left it (the kernels may still run).
static int count_feed_host(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes, bool pinned_source)
{
    const size_t km1 = (size_t)ctx->k - 1;
    const size_t stage = kpal_ctx::kStage;
    const size_t pad = kpal_ctx::kStagePad;  // room for the halo, keeps the payload 16-byte aligned
    CHK(ensure_pinned(ctx));
    for (int i = 0; i < 2; ++i) CHK(ensure(ctx, ctx->dstage[i], stage + pad));
    int slot = 0;
    for (size_t off = 0; off < nbytes; off += stage, slot ^= 1) {
        const size_t len = std::min(stage, nbytes - off);
        const size_t h = std::min(km1, off);
        // the pinned/device slot is free once the H2D copy (pinned) and the kernels (device) that used it are done
        if (ctx->stage_used[slot]) {
            HIPCHK(hipEventSynchronize(ctx->ev_copied[slot]));
            HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_done[slot], 0));
        }
        const uint8_t *hp = host_buf + off - h;
        if (!pinned_source) {
            uint8_t *staged = (uint8_t *)ctx->pinned[slot] + (pad - h);
            staged_memcpy(staged, host_buf + off - h, len + h);
            hp = staged;
        }
        uint8_t *dp = (uint8_t *)ctx->dstage[slot].p + (pad - h);
        HIPCHK(hipMemcpyAsync(dp, hp, len + h, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        CHK(count_device_range(ctx, dp + h, len, h));
        HIPCHK(hipEventRecord(ctx->ev_done[slot], ctx->stream));
        ctx->stage_used[slot] = true;
    }
    if (pinned_source) HIPCHK(hipStreamSynchronize(ctx->copy_stream));   // the caller may refill its buffer
    return KPAL_OK;
}

KPAL_API int kpal_count_feed(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!host_buf) return set_err(KPAL_E_INVALID, "host_buf is NULL");
    return count_feed_host(ctx, host_buf, nbytes, false);
}

KPAL_API int kpal_count_feed_pinned(kpal_ctx *ctx, const uint8_t *pinned_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed_pinned before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!pinned_buf) return set_err(KPAL_E_INVALID, "pinned_buf is NULL");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, pinned_buf) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return set_err(KPAL_E_INVALID, "kpal_count_feed_pinned needs memory from kpal_host_alloc");
    }
    return count_feed_host(ctx, pinned_buf, nbytes, true);
}

KPAL_API int kpal_host_alloc(kpal_ctx *ctx, size_t nbytes, void **host_out)
{
    CTX_ENTER(ctx);
    if (!host_out) return set_err(KPAL_E_INVALID, "host_out is NULL");
    CHK(host_alloc_near_gpu(ctx, host_out, nbytes ? nbytes : 16));
    ctx->host_allocs.push_back(*host_out);
    return KPAL_OK;
}

KPAL_API int kpal_host_free(kpal_ctx *ctx, void *host)
{
    CTX_ENTER(ctx);
    if (!host) return KPAL_OK;
    auto it = std::find(ctx->host_allocs.begin(), ctx->host_allocs.end(), host);
    if (it == ctx->host_allocs.end()) return set_err(KPAL_E_INVALID, "not a buffer of kpal_host_alloc of this context");
    HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    HIPCHK(hipHostFree(host));
    ctx->host_allocs.erase(it);
    return KPAL_OK;
}

// ----------------------------------------------------------------------------------------------
// FASTA ingest: text (a byte range of a file, or host memory) -> pinned staging -> device -> flattened on the device -> counted,
// chunk i+1 being read, copied and flattened while chunk i is counted.  Nothing in the loop waits for the GPU except for the
// flattened SIZE of the chunk before (read back asynchronously, needed on the host to launch its count), which is one whole
// chunk old by then.
// ----------------------------------------------------------------------------------------------
// The pipeline.  count: the flattened chunks are counted into the running count (windows span chunk seams through the saved
// tail of the chunk before, never a record boundary: every header leaves a '\n' in the stream); else they are copied to host_out.
static int fasta_pipeline(kpal_ctx *ctx, FaSource &src, bool count, uint8_t *host_out, uint64_t *n_out)
{
    const size_t stage = ctx->fa_chunk, pad = kpal_ctx::kStagePad;
    const size_t km1 = count ? (size_t)ctx->k - 1 : 0;
    CHK(ensure_pinned(ctx));
    if (!ctx->fa_nflat_host) {
        hipError_t e = hipHostMalloc((void **)&ctx->fa_nflat_host, 64, hipHostMallocDefault);
        if (e != hipSuccess) return set_err(KPAL_E_NOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    CHK(ensure(ctx, ctx->fa_tail, 64));
    const uint32_t max_blocks = (uint32_t)((stage + kFaBlockBytes - 1) / kFaBlockBytes);
    for (int i = 0; i < 2; ++i) {
        CHK(ensure(ctx, ctx->fa_raw[i], stage + 64));
        CHK(ensure(ctx, ctx->fa_flat[i], stage + pad + 64));
        CHK(ensure(ctx, ctx->fa_meta[i], (size_t)max_blocks * (8 + 8 + 4) + (size_t)(max_blocks + 1) * 8 + 64));
    }
    int prev_slot = -1;          // the chunk that has been flattened but not consumed yet
    uint64_t flat_total = 0;     // flattened bytes of the chunks consumed so far (this feed)
    uint64_t out_total = 0;

    auto consume = [&](int slot) -> int {
        HIPCHK(hipEventSynchronize(ctx->ev_done[slot]));   // (its flattening finished about one chunk ago)
        const uint64_t nf = ctx->fa_nflat_host[slot];
        uint8_t *flat = (uint8_t *)ctx->fa_flat[slot].p + pad;
        if (!count) {
            if (nf) HIPCHK(hipMemcpyAsync(host_out + out_total, flat, nf, hipMemcpyDeviceToHost, ctx->stream));
            out_total += nf;
            return KPAL_OK;
        }
        const size_t h = (size_t)std::min<uint64_t>(km1, flat_total);    // flattened bytes of this feed that precede the chunk
        if (h) HIPCHK(hipMemcpyAsync(flat - h, ctx->fa_tail.p, h, hipMemcpyDeviceToDevice, ctx->stream));
        const size_t h2 = (size_t)std::min<uint64_t>(km1, h + nf);       // ... and the next one: the last bytes of [flat - h, flat + nf)
        if (h2) HIPCHK(hipMemcpyAsync(ctx->fa_tail.p, flat + nf - h2, h2, hipMemcpyDeviceToDevice, ctx->stream));
        if (nf) CHK(count_device_range(ctx, flat, (size_t)nf, h));
        flat_total += nf;
        return KPAL_OK;
    };

    // The chunker (fasta_host.hpp) reads chunk i + 1 into the other pinned buffer while the launches of chunk i are issued; a
    // pinned buffer is written again only after the DMA out of it has finished.
    FaChunker chunker(src, (uint8_t *)ctx->pinned[0], (uint8_t *)ctx->pinned[1], stage, [ctx](int slot) -> int {
        if (ctx->stage_used[slot] && hipEventSynchronize(ctx->ev_copied[slot]) != hipSuccess) return KPAL_E_HIP;
        return 0;
    });
    FaChunk ck;
    for (;;) {
        const int got = chunker.next(ck);
        if (got == 0) break;
        if (got == -1) return set_err(KPAL_E_IO, "reading the FASTA input failed: %s", strerror(chunker.io_errno()));
        if (got < 0) return set_err(KPAL_E_HIP, "hipEventSynchronize failed while reading the FASTA input");
        const int slot = ck.slot;
        const uint8_t *chunk = ck.data;
        const size_t m = ck.n;
        const int state = ck.state;
        const int tail = ck.tail_trailing ? 1 : 0;
        const uint32_t nblocks = (uint32_t)((m + kFaBlockBytes - 1) / kFaBlockBytes);
        uint8_t *raw = (uint8_t *)ctx->fa_raw[slot].p;
        uint8_t *flat = (uint8_t *)ctx->fa_flat[slot].p + pad;
        long long *last_eol = (long long *)ctx->fa_meta[slot].p;
        long long *eol_before = last_eol + nblocks;
        uint64_t *offs = (uint64_t *)(eol_before + nblocks);
        uint32_t *kept = (uint32_t *)(offs + nblocks + 1);
        // the device copy of the raw text is free once the flattening that read it is done (two chunks ago)
        if (ctx->stage_used[slot]) HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_done[slot], 0));
        HIPCHK(hipMemcpyAsync(raw, chunk, m, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        LAUNCH(ctx, "fa_last_eol", fa_last_eol_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, last_eol);
        LAUNCH(ctx, "fa_carry", fa_carry_kernel, dim3(1), dim3(256), (const long long *)last_eol, nblocks, eol_before);
        LAUNCH(ctx, "fa_count", fa_count_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, (const long long *)eol_before, state, tail, kept);
        LAUNCH(ctx, "fa_offset", fa_offset_kernel, dim3(1), dim3(256), (const uint32_t *)kept, nblocks, offs);
        LAUNCH(ctx, "fa_scatter", fa_scatter_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m,
               (const long long *)eol_before, state, tail, (const uint64_t *)offs, flat);
        HIPCHK(hipMemcpyAsync(&ctx->fa_nflat_host[slot], offs + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev_done[slot], ctx->stream));
        ctx->stage_used[slot] = true;
        // the chunk before: its flattened size has long arrived; its count is queued behind this chunk's flattening
        if (prev_slot >= 0) CHK(consume(prev_slot));
        prev_slot = slot;
    }
    if (prev_slot >= 0) CHK(consume(prev_slot));
    if (!count) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *n_out = out_total;
    }
    return KPAL_OK;
}

KPAL_API int kpal_count_feed_fasta(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed_fasta before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!host_buf) return set_err(KPAL_E_INVALID, "host_buf is NULL");
    FaSource src;
    src.mem = host_buf;
    src.end = nbytes;
    return fasta_pipeline(ctx, src, true, nullptr, nullptr);
}

KPAL_API int kpal_count_feed_fasta_file(kpal_ctx *ctx, const char *path, uint64_t begin, uint64_t end, const uint8_t *prefix, size_t prefix_len)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed_fasta_file before kpal_count_begin");
    if (!path) return set_err(KPAL_E_INVALID, "path is NULL");
    if (prefix_len && !prefix) return set_err(KPAL_E_INVALID, "prefix is NULL");
    if (prefix_len > ((size_t)1 << 20)) return set_err(KPAL_E_INVALID, "prefix longer than 1 MiB");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return set_err(KPAL_E_IO, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return set_err(KPAL_E_IO, "%s is not a regular file", path);
    }
    const uint64_t size = (uint64_t)st.st_size;
    if (end == 0) end = size;
    if (begin > end || end > size) {
        close(fd);
        return set_err(KPAL_E_INVALID, "byte range %llu..%llu outside %s (%llu bytes)", (unsigned long long)begin, (unsigned long long)end, path,
                       (unsigned long long)size);
    }
    (void)posix_fadvise(fd, (off_t)begin, (off_t)(end - begin), POSIX_FADV_SEQUENTIAL);
    FaSource src;
    src.fd = fd;
    src.pos = begin;
    src.end = end;
    src.prefix = prefix;
    src.prefix_left = prefix_len;
    const int rc = fasta_pipeline(ctx, src, true, nullptr, nullptr);
    close(fd);
    return rc;
}

KPAL_API int kpal_fasta_flatten(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes, uint8_t *host_out, uint64_t *n_out)
{
    CTX_ENTER(ctx);
    if (!n_out || (nbytes && (!host_buf || !host_out))) return set_err(KPAL_E_INVALID, "NULL pointer");
    *n_out = 0;
    if (nbytes == 0) return KPAL_OK;
    FaSource src;
    src.mem = host_buf;
    src.end = nbytes;
    return fasta_pipeline(ctx, src, false, host_out, n_out);
}

KPAL_API int kpal_count_records(kpal_ctx *ctx, int k, const uint8_t *host_flat, size_t nbytes, const uint64_t *host_starts,
                                size_t n_records, int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);
    if (n_records == 0) return KPAL_OK;
    if (!host_starts || !host_out || (nbytes && !host_flat)) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (n_records >= 0xFFFFFFFFull) return set_err(KPAL_E_INVALID, "too many records in one batch");
    if (host_starts[0] != 0 || host_starts[n_records] != nbytes) return set_err(KPAL_E_INVALID, "starts must run from 0 to nbytes");
    for (size_t r = 0; r < n_records; ++r)
        if (host_starts[r] > host_starts[r + 1]) return set_err(KPAL_E_INVALID, "starts must be ascending");
    const uint64_t bins = 1ULL << (2 * k);
    const size_t out_bytes = n_records * bins * sizeof(int64_t);
    CHK(ensure(ctx, ctx->scratch[0], out_bytes));
    CHK(ensure(ctx, ctx->scratch[1], nbytes + 64));
    CHK(ensure(ctx, ctx->scratch[2], (n_records + 1) * sizeof(uint64_t)));
    HIPCHK(hipMemsetAsync(ctx->scratch[0].p, 0, out_bytes, ctx->stream));
    if (nbytes) {
        HIPCHK(hipMemcpyAsync(ctx->scratch[1].p, host_flat, nbytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->scratch[2].p, host_starts, (n_records + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        const Span s = make_span((const uint8_t *)ctx->scratch[1].p, nbytes, 0);
        const uint64_t steps = (s.nchunks + 63) / 64;
        const uint64_t max_waves = (uint64_t)ctx->num_cu * 8 * 4;
        const uint64_t spw = std::max<uint64_t>(1, (steps + max_waves - 1) / max_waves);
        const uint64_t waves = (steps + spw - 1) / spw;
        const unsigned grid = (unsigned)((waves + 3) / 4);
        DISPATCH_K_1_16(k, LAUNCH(ctx, "count_records", (count_records_kernel<K>), dim3(grid), dim3(256), s, spw,
                                  (const uint64_t *)ctx->scratch[2].p, (uint32_t)n_records, (unsigned long long *)ctx->scratch[0].p));
    }
    HIPCHK(hipMemcpyAsync(host_out, ctx->scratch[0].p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

// ----------------------------------------------------------------------------------------------
// Profile.from_fasta_by_record (kpal/klib.py:114-133) with the records tokenised on the device: the text (whole records; the
// caller cuts at record boundaries) is flattened by the kernels of the FASTA ingest, and two compactions list where every record
// starts in the flattened stream and where its header line starts in the text -- the host reads the header lines only (names).
// kpal_fasta_records_count then counts batches of records into one table each (count_records_kernel), as many as the caller has
// room for.
// ----------------------------------------------------------------------------------------------
// in_pinned0: host_text lies in ctx->pinned[0] (the file reader put it there): the DMA engine reads it in place
static int fasta_records_index_text(kpal_ctx *ctx, const uint8_t *host_text, size_t nbytes, bool in_pinned0, uint64_t *n_records, uint64_t *flat_bytes)
{
    *n_records = *flat_bytes = 0;
    ctx->rec_n = ctx->rec_nf = 0;
    ctx->rec_starts_host.clear();
    ctx->rec_hdr_host.clear();
    const size_t first = nbytes ? fasta_first_header(host_text, nbytes, true) : 0;   // text before the first header is no record (klib.py:131: SeqIO)
    if (first >= nbytes) return KPAL_OK;
    const uint8_t *text = host_text + first;
    const size_t m = nbytes - first;
    const size_t pad = kpal_ctx::kStagePad;
    const uint32_t nblocks = (uint32_t)((m + kFaBlockBytes - 1) / kFaBlockBytes);
    CHK(ensure(ctx, ctx->rec_raw, m + 64));
    CHK(ensure(ctx, ctx->rec_flat, m + pad + 64));
    CHK(ensure(ctx, ctx->rec_meta, (size_t)nblocks * (8 + 8 + 4 + 4) + 2 * (size_t)(nblocks + 1) * 8 + 128));
    uint8_t *raw = (uint8_t *)ctx->rec_raw.p;
    uint8_t *flat = (uint8_t *)ctx->rec_flat.p + pad;
    long long *last_eol = (long long *)ctx->rec_meta.p;
    long long *eol_before = last_eol + nblocks;
    uint64_t *offs = (uint64_t *)(eol_before + nblocks);
    uint64_t *offs2 = offs + nblocks + 1;
    uint32_t *kept = (uint32_t *)(offs2 + nblocks + 1);
    uint32_t *marks = kept + nblocks;
    // text -> device through the pinned staging buffers (host threads copy piece i + 1 while the DMA takes piece i)
    CHK(ensure_pinned(ctx));
    if (in_pinned0) {
        if (ctx->stage_used[0]) HIPCHK(hipEventSynchronize(ctx->ev_copied[0]));   // (an earlier DMA out of the buffer: long done, the caller has refilled it)
        HIPCHK(hipMemcpyAsync(raw, text, m, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(hipEventRecord(ctx->ev_copied[0], ctx->copy_stream));
        ctx->stage_used[0] = true;
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[0], 0));
    } else {
        const size_t stage = kpal_ctx::kStage;
        int slot = 0;
        for (size_t off = 0; off < m; off += stage, slot ^= 1) {
            const size_t len = std::min(stage, m - off);
            if (ctx->stage_used[slot]) HIPCHK(hipEventSynchronize(ctx->ev_copied[slot]));
            staged_memcpy(ctx->pinned[slot], text + off, len);
            HIPCHK(hipMemcpyAsync(raw + off, ctx->pinned[slot], len, hipMemcpyHostToDevice, ctx->copy_stream));
            HIPCHK(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
            ctx->stage_used[slot] = true;
            if (off + stage >= m) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        }
    }
    LAUNCH(ctx, "fa_last_eol", fa_last_eol_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, last_eol);
    LAUNCH(ctx, "fa_carry", fa_carry_kernel, dim3(1), dim3(256), (const long long *)last_eol, nblocks, eol_before);
    LAUNCH(ctx, "fa_count", fa_count_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, (const long long *)eol_before, 0, 1, kept);
    LAUNCH(ctx, "fa_offset", fa_offset_kernel, dim3(1), dim3(256), (const uint32_t *)kept, nblocks, offs);
    LAUNCH(ctx, "fa_scatter", fa_scatter_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, (const long long *)eol_before, 0, 1,
           (const uint64_t *)offs, flat);
    // header lines of the text
    LAUNCH(ctx, "fa_mark_count", (fa_mark_count_kernel<1>), dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, marks);
    LAUNCH(ctx, "fa_offset", fa_offset_kernel, dim3(1), dim3(256), (const uint32_t *)marks, nblocks, offs2);
    uint64_t sizes[2] = {0, 0};   // flattened bytes, records
    HIPCHK(hipMemcpyAsync(&sizes[0], offs + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&sizes[1], offs2 + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t nf = sizes[0], R = sizes[1];
    if (R == 0 || nf < R) return set_err(KPAL_E_HIP, "record index: %llu records in %llu flattened bytes", (unsigned long long)R, (unsigned long long)nf);
    CHK(ensure(ctx, ctx->rec_hdr, (size_t)R * 8));
    CHK(ensure(ctx, ctx->rec_starts, (size_t)(R + 1) * 8));
    LAUNCH(ctx, "fa_mark_scatter", (fa_mark_scatter_kernel<1>), dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, (uint64_t)m, (const uint64_t *)offs2,
           (uint64_t *)ctx->rec_hdr.p);
    // record starts of the flattened stream: its '\n' bytes (offs / marks are free again: the flattening is done)
    const uint32_t fblocks = (uint32_t)((nf + kFaBlockBytes - 1) / kFaBlockBytes);   // <= nblocks
    LAUNCH(ctx, "fa_mark_count", (fa_mark_count_kernel<0>), dim3(fblocks), dim3(kFaThreads), (const uint8_t *)flat, nf, marks);
    LAUNCH(ctx, "fa_offset", fa_offset_kernel, dim3(1), dim3(256), (const uint32_t *)marks, fblocks, offs);
    LAUNCH(ctx, "fa_mark_scatter", (fa_mark_scatter_kernel<0>), dim3(fblocks), dim3(kFaThreads), (const uint8_t *)flat, nf, (const uint64_t *)offs,
           (uint64_t *)ctx->rec_starts.p);
    uint64_t seps = 0;
    HIPCHK(hipMemcpyAsync(&seps, offs + fblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync((uint64_t *)ctx->rec_starts.p + R, &nf, sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->rec_hdr_host.resize((size_t)R);
    ctx->rec_starts_host.resize((size_t)R + 1);
    HIPCHK(hipMemcpyAsync(ctx->rec_hdr_host.data(), ctx->rec_hdr.p, (size_t)R * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->rec_starts_host.data(), ctx->rec_starts.p, (size_t)R * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (seps != R) return set_err(KPAL_E_HIP, "record index: %llu header lines but %llu separators", (unsigned long long)R, (unsigned long long)seps);
    ctx->rec_starts_host[(size_t)R] = nf;
    for (uint64_t &h : ctx->rec_hdr_host) h += first;
    ctx->rec_n = R;
    ctx->rec_nf = nf;
    *n_records = R;
    *flat_bytes = nf;
    return KPAL_OK;
}

KPAL_API int kpal_fasta_records_begin(kpal_ctx *ctx, const uint8_t *host_text, size_t nbytes, uint64_t *n_records, uint64_t *flat_bytes)
{
    CTX_ENTER(ctx);
    if (!n_records || !flat_bytes || (nbytes && !host_text)) return set_err(KPAL_E_INVALID, "NULL pointer");
    return fasta_records_index_text(ctx, host_text, nbytes, false, n_records, flat_bytes);
}

// ---- the same over a FILE the library reads itself (the pool's threads pread into the pinned staging buffer: no byte of the text
// passes through Python): every kpal_fasta_records_file_next indexes the next piece of WHOLE records -- up to the end of line
// before the last header line of what fits the 64 MiB staging buffer; the unfinished record behind it is carried to the next
// piece; a record longer than the buffer is gathered in pageable memory first.
static void fasta_records_file_reset(kpal_ctx *ctx)
{
    if (ctx->rec_fd >= 0) close(ctx->rec_fd);
    ctx->rec_fd = -1;
    ctx->rec_pos = ctx->rec_end = ctx->rec_piece_at = 0;
    ctx->rec_carry.clear();
    ctx->rec_carry.shrink_to_fit();
}

KPAL_API int kpal_fasta_records_file_open(kpal_ctx *ctx, const char *path, uint64_t begin, uint64_t end)
{
    CTX_ENTER(ctx);
    if (!path) return set_err(KPAL_E_INVALID, "path is NULL");
    fasta_records_file_reset(ctx);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return set_err(KPAL_E_IO, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return set_err(KPAL_E_IO, "%s is not a regular file", path);
    }
    const uint64_t size = (uint64_t)st.st_size;
    if (end == 0) end = size;
    if (begin > end || end > size) {
        close(fd);
        return set_err(KPAL_E_INVALID, "byte range %llu..%llu outside %s (%llu bytes)", (unsigned long long)begin, (unsigned long long)end, path,
                       (unsigned long long)size);
    }
    (void)posix_fadvise(fd, (off_t)begin, (off_t)(end - begin), POSIX_FADV_SEQUENTIAL);
    ctx->rec_fd = fd;
    ctx->rec_pos = ctx->rec_piece_at = begin;
    ctx->rec_end = end;
    return KPAL_OK;
}

// index of the end-of-line byte before the LAST header line of buf[0, n) that begins at or after `from` (a '>' behind an
// end of line), or n when there is none
static size_t fasta_last_boundary(const uint8_t *buf, size_t n, size_t from)
{
    size_t i = n;
    while (i > from + 1) {
        const void *p = memrchr(buf + from + 1, '>', i - from - 1);
        if (!p) break;
        const size_t at = (size_t)((const uint8_t *)p - buf);
        if (fa_host_is_eol(buf[at - 1])) return at - 1;
        i = at;
    }
    return n;
}

KPAL_API int kpal_fasta_records_file_next(kpal_ctx *ctx, uint64_t *n_records, uint64_t *flat_bytes, uint64_t *text_offset, int *done)
{
    CTX_ENTER(ctx);
    if (!n_records || !flat_bytes || !text_offset || !done) return set_err(KPAL_E_INVALID, "NULL pointer");
    *n_records = *flat_bytes = *text_offset = 0;
    *done = 1;
    if (ctx->rec_fd < 0) return set_err(KPAL_E_STATE, "kpal_fasta_records_file_next without kpal_fasta_records_file_open");
    CHK(ensure_pinned(ctx));
    const size_t stage = std::min<size_t>(kpal_ctx::kStage, ctx->fa_chunk);   // (KPAL_FASTA_CHUNK: tests put the seams everywhere)
    FaSource src;
    src.fd = ctx->rec_fd;
    for (;;) {
        if (ctx->rec_carry.empty() && ctx->rec_pos >= ctx->rec_end) {   // the end
            fasta_records_file_reset(ctx);
            *n_records = *flat_bytes = *text_offset = 0;
            *done = 1;
            return KPAL_OK;
        }
        const size_t c = ctx->rec_carry.size();
        const bool fits = c < stage;
        uint8_t *buf;
        size_t n;
        if (fits) {   // the carried tail + the next bytes of the file into the pinned buffer
            if (ctx->stage_used[0]) HIPCHK(hipEventSynchronize(ctx->ev_copied[0]));
            buf = (uint8_t *)ctx->pinned[0];
            if (c) memcpy(buf, ctx->rec_carry.data(), c);
            const size_t want = (size_t)std::min<uint64_t>(stage - c, ctx->rec_end - ctx->rec_pos);
            if (want) {
                std::vector<int> ok;
                fa_copy_start(src, buf + c, ctx->rec_pos, want, ok);
                HostPool::instance().wait();
                for (int e : ok)
                    if (e) return set_err(KPAL_E_IO, "reading the FASTA input failed: %s", strerror(e));
                ctx->rec_pos += want;
            }
            n = c + want;
        } else {      // a record longer than the staging buffer: gathered in pageable memory, 64 MiB at a time
            const size_t want = (size_t)std::min<uint64_t>(stage, ctx->rec_end - ctx->rec_pos);
            ctx->rec_carry.resize(c + want);
            if (want) {
                std::vector<int> ok;
                fa_copy_start(src, ctx->rec_carry.data() + c, ctx->rec_pos, want, ok);
                HostPool::instance().wait();
                for (int e : ok)
                    if (e) return set_err(KPAL_E_IO, "reading the FASTA input failed: %s", strerror(e));
                ctx->rec_pos += want;
            }
            buf = ctx->rec_carry.data();
            n = c + want;
        }
        const bool at_end = ctx->rec_pos >= ctx->rec_end;
        // whole records: up to the end of line before the last header line (searched in the new bytes only; a boundary is two bytes)
        const size_t cut = at_end ? n : fasta_last_boundary(buf, n, c ? c - 1 : 0);
        if (cut >= n && !at_end) {   // no record ends in this piece: keep gathering
            if (fits) ctx->rec_carry.assign(buf, buf + n);
            continue;
        }
        const size_t piece = at_end ? n : cut + 1;
        const uint64_t at = ctx->rec_piece_at;
        const int rc = fasta_records_index_text(ctx, buf, piece, fits, n_records, flat_bytes);
        if (rc != KPAL_OK) return rc;
        // the unfinished record behind the piece is carried (the DMA out of the pinned buffer has been waited for: the index is complete)
        std::vector<uint8_t> tail(buf + piece, buf + n);
        ctx->rec_carry.swap(tail);
        ctx->rec_piece_at = at + piece;
        *text_offset = at;
        *done = 0;
        if (*n_records == 0 && !(ctx->rec_carry.empty() && at_end)) continue;   // (text before the first header only: next piece)
        return KPAL_OK;
    }
}

// Where the scan stands: the file offset of the first byte that no piece has covered yet (the start of the carried,
// unfinished record).  A caller that lets other work use the context between two pieces keeps THIS, closes the scan and
// opens it again there: the scan state of the context -- descriptor, position, carried bytes -- then never outlives a call.
KPAL_API int kpal_fasta_records_file_tell(kpal_ctx *ctx, uint64_t *offset)
{
    CTX_ENTER(ctx);
    if (!offset) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (ctx->rec_fd < 0) return set_err(KPAL_E_STATE, "kpal_fasta_records_file_tell without an open scan");
    *offset = ctx->rec_piece_at;
    return KPAL_OK;
}

KPAL_API int kpal_fasta_records_file_close(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    fasta_records_file_reset(ctx);
    return KPAL_OK;
}

KPAL_API int kpal_fasta_records_index(kpal_ctx *ctx, uint64_t *header_off, uint64_t *flat_start)
{
    CTX_ENTER(ctx);
    if (ctx->rec_n == 0) return set_err(KPAL_E_STATE, "kpal_fasta_records_index without records (kpal_fasta_records_begin)");
    if (header_off) memcpy(header_off, ctx->rec_hdr_host.data(), (size_t)ctx->rec_n * 8);
    if (flat_start) memcpy(flat_start, ctx->rec_starts_host.data(), (size_t)(ctx->rec_n + 1) * 8);
    return KPAL_OK;
}

// the tables of records [first, first + n) of the indexed text into n x 4^k int64 of DEVICE memory (queued on the context's stream)
static int fasta_records_count_into(kpal_ctx *ctx, int k, uint64_t first, uint64_t n, unsigned long long *dev_out)
{
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);
    if (first > ctx->rec_n || n > ctx->rec_n - first) return set_err(KPAL_E_INVALID, "records %llu..%llu of %llu", (unsigned long long)first,
                                                                      (unsigned long long)(first + n), (unsigned long long)ctx->rec_n);
    if (n >= 0xFFFFFFFFull) return set_err(KPAL_E_INVALID, "too many records in one batch");
    const uint64_t bins = 1ULL << (2 * k);
    const size_t out_bytes = (size_t)n * bins * sizeof(int64_t);
    const uint64_t b0 = ctx->rec_starts_host[(size_t)first], b1 = ctx->rec_starts_host[(size_t)(first + n)];
    CHK(ensure(ctx, ctx->scratch[2], (size_t)(n + 1) * sizeof(uint64_t)));
    HIPCHK(hipMemsetAsync(dev_out, 0, out_bytes, ctx->stream));
    if (b1 > b0) {
        LAUNCH(ctx, "fa_rebase", fa_rebase_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), (const uint64_t *)ctx->rec_starts.p + first, n + 1, b0,
               (uint64_t *)ctx->scratch[2].p);
        const Span s = make_span((const uint8_t *)ctx->rec_flat.p + kpal_ctx::kStagePad + b0, (size_t)(b1 - b0), 0);
        const uint64_t steps = (s.nchunks + 63) / 64;
        const uint64_t max_waves = (uint64_t)ctx->num_cu * 8 * 4;
        const uint64_t spw = std::max<uint64_t>(1, (steps + max_waves - 1) / max_waves);
        const uint64_t waves = (steps + spw - 1) / spw;
        const unsigned grid = (unsigned)((waves + 3) / 4);
        DISPATCH_K_1_16(k, LAUNCH(ctx, "count_records", (count_records_kernel<K>), dim3(grid), dim3(256), s, spw,
                                  (const uint64_t *)ctx->scratch[2].p, (uint32_t)n, dev_out));
    }
    return KPAL_OK;
}

KPAL_API int kpal_fasta_records_count(kpal_ctx *ctx, int k, uint64_t first, uint64_t n, int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (n == 0) return KPAL_OK;
    if (!host_out) return set_err(KPAL_E_INVALID, "host_out is NULL");
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);
    const size_t out_bytes = (size_t)n * ((size_t)1 << (2 * k)) * sizeof(int64_t);
    CHK(ensure(ctx, ctx->scratch[0], out_bytes));
    CHK(fasta_records_count_into(ctx, k, first, n, (unsigned long long *)ctx->scratch[0].p));
    HIPCHK(hipMemcpyAsync(host_out, ctx->scratch[0].p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

// ... into the CALLER's device memory (kpal_dev_alloc): the profiles of a by-record scan that stay in HBM until something on the
// host asks for their counts (kpal_amd/klib.py: Profile.counts is materialised lazily; distances and matrices of such profiles
// read the device copies)
KPAL_API int kpal_fasta_records_count_device(kpal_ctx *ctx, int k, uint64_t first, uint64_t n, int64_t *dev_out)
{
    CTX_ENTER(ctx);
    if (n == 0) return KPAL_OK;
    if (!dev_out) return set_err(KPAL_E_INVALID, "dev_out is NULL");
    CHK(fasta_records_count_into(ctx, k, first, n, (unsigned long long *)dev_out));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // (the index may be overwritten by the caller's next piece)
    return KPAL_OK;
}

KPAL_API int kpal_count_finish(kpal_ctx *ctx, int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_finish before kpal_count_begin");
    CHK(table_ready(ctx));
    uint32_t pool_error = 0, quad_error = 0;
    if (ctx->chunk_error_armed)
        HIPCHK(hipMemcpyAsync(&pool_error, ctx->chunk_error_word, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->quad_error_word)
        HIPCHK(hipMemcpyAsync(&quad_error, ctx->quad_error_word, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    // (a double-buffered pinned staging path was measured slower than the runtime's pageable copy)
    if (host_out)
        HIPCHK(hipMemcpyAsync(host_out, ctx->table.p, ctx->bins * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (pool_error) {
        HIPCHK(hipMemsetAsync(ctx->chunk_error_word, 0, sizeof(uint32_t), ctx->stream));
        return set_err(KPAL_E_HIP, "chunked partition: the chunk pool ran out (internal sizing error %u); counts are invalid", pool_error);
    }
    if (quad_error) {
        HIPCHK(hipMemsetAsync(ctx->quad_error_word, 0, sizeof(uint32_t), ctx->stream));
        return set_err(KPAL_E_HIP, "quad partition: internal sizing error %u; counts are invalid", quad_error);
    }
    return KPAL_OK;
}

KPAL_API int kpal_count_balance(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_balance before kpal_count_begin");
    // two-level quad pipeline: the pending finalisation of the table balances it in the same pass
    if (ctx->finalize_pending) return quad2_finalize(ctx, true);
    CHK(table_ready(ctx));
    return launch_balance(ctx, ctx->k, (const int64_t *)ctx->table.p, (int64_t *)ctx->table.p);
}

KPAL_API int kpal_count_last_plan(kpal_ctx *ctx, int *strategy, int *steps1, int *steps2)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (strategy) *strategy = ctx->plan_strategy;
    if (steps1) *steps1 = ctx->plan_steps1;
    if (steps2) *steps2 = ctx->plan_steps2;
    return KPAL_OK;
}

KPAL_API int kpal_count_stats(kpal_ctx *ctx, uint64_t *out, int n)
{
    CTX_ENTER(ctx);
    if (!out || n < 1) return set_err(KPAL_E_INVALID, "out is NULL");
    uint32_t words[4] = {0, 0, 0, 0};
    if (ctx->quad_error_word) {
        HIPCHK(hipMemcpyAsync(words, ctx->quad_error_word, sizeof(words), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    const uint64_t all[KPAL_COUNT_STATS] = {words[1], words[2], words[3], ctx->stat_fresh_pieces, ctx->stat_fresh_reruns, ctx->stat_quad_pieces,
                                            ctx->stat_chunked_pieces, ctx->stat_split_pieces, ctx->stat_repeat_pieces};
    for (int i = 0; i < n; ++i) out[i] = i < KPAL_COUNT_STATS ? all[i] : 0;
    return KPAL_OK;
}

KPAL_API int kpal_count_table(kpal_ctx *ctx, void **dev_table, uint64_t *n_bins)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (!ctx->counting) return set_err(KPAL_E_STATE, "no count table (call kpal_count_begin)");
    HIPCHK(hipSetDevice(ctx->device));
    CHK(table_ready(ctx));   // the caller is about to use the table
    if (dev_table) *dev_table = ctx->table.p;
    if (n_bins) *n_bins = ctx->bins;
    return KPAL_OK;
}

KPAL_API int kpal_synth_reads_device(kpal_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                     int read_len, int noisy, void *dev_out)
{
    CTX_ENTER(ctx);
    if (read_len < 1) return set_err(KPAL_E_INVALID, "read_len must be >= 1");
    if (n_reads == 0) return KPAL_OK;
    if (!dev_out || ((uintptr_t)dev_out & 15)) return set_err(KPAL_E_INVALID, "dev_out must be a 16-byte aligned device pointer");
    const uint64_t total = n_reads * (uint64_t)(read_len + 1);
    const uint64_t nvec = (total + 15) / 16;
    const unsigned grid = (unsigned)std::min<uint64_t>((nvec + 255) / 256, (uint64_t)ctx->num_cu * 16);
    LAUNCH(ctx, "synth_reads", synth_reads_kernel, dim3(grid), dim3(256), seed, first_read, n_reads,
           (uint32_t)read_len, noisy, (uint8_t *)dev_out);
    return KPAL_OK;
}


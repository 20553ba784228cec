// kpal_quads2.hip -- launch planning of the two-level quad record pipeline, k = 13..16 (quad_kernels.hpp, second half).
#include "kpal_host.hpp"

#include "quad_kernels.hpp"

// level 1, eight steps per tile: input chunks requested FOUR steps ahead (4 KiB per wave in flight) -- with a ring of eight the
// kernel sits at 128 registers and the sink handling of its epilogue spills; kernels that use scratch at all ran 8 % slower
#ifndef KPAL_L1_DEPTH8
#define KPAL_L1_DEPTH8 4
#endif

// Two-level partition of quads, k = 13..16 (quad_kernels.hpp, end): level-1 records by coarse bucket, level-2 records
// by (coarse, fine) bucket, histogram per (coarse, fine) bucket.
int launch_partition2_quads(kpal_ctx *ctx, const Span &s, bool fresh)
{
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (total_steps == 0) return KPAL_OK;
    const int K = ctx->k;
    const uint32_t NB1 = 1u << (2 * K - 22);
    const uint32_t REP = NB1 >= 256 ? 1u : 256u / NB1;
    const uint32_t S1 = (uint32_t)kQuadRowWords / (NB1 * REP);
    CHK(ensure(ctx, ctx->quad_meta, ((size_t)ctx->num_cu + 4 + 2048 + 512) * sizeof(uint32_t)));
    uint32_t *nrounds1 = (uint32_t *)ctx->quad_meta.p;
    uint32_t *error = nrounds1 + ctx->num_cu;
    if (!ctx->quad_error_word) {
        HIPCHK(hipMemsetAsync(error, 0, 4 * sizeof(uint32_t), ctx->stream));
        ctx->quad_error_word = error;
    }
    // level 1: 16 waves x 7 wave-steps per tile bring 107 items per 128-slot row (26.7 per 32 at k = 16) for uniform
    // k-mers; the sampled row loads say whether THIS feed needs a smaller tile, or (AUTO) the round-1 pipeline
    static const int candidates[] = {8, 7, 6, 3};
    int steps1 = 0;
    for (int c : candidates)
        if (c == ctx->quad_steps_forced) steps1 = c;
    std::vector<double> fine;                    // items per fine row of level 2 per wave-step of INPUT (sorted)
    // a later feed of the same count of about the same size reuses the tile sizes of the sampled one (re-sampled every 16 feeds)
    const size_t feed_bytes = (size_t)(s.hi - s.emit_from);
    const bool cached = ctx->cached_steps1 && ctx->cached_steps2 && ctx->cached_uses < 16 && feed_bytes <= 2 * ctx->cached_bytes &&
                        2 * feed_bytes >= ctx->cached_bytes;
    if (cached) {
        if (!steps1) steps1 = ctx->cached_steps1;
        ++ctx->cached_uses;
    } else {
        int chosen = 0;
        const int rc = quad_choose_steps(ctx, s, error + 4, (int)(NB1 * REP), (int)S1, 16, candidates, 4, &chosen, &fine);
        if (rc != KPAL_OK) return rc;
        if (!steps1) steps1 = chosen;
    }
    const bool repeat1 = ctx->quad_repeat_forced >= 0 ? ctx->quad_repeat_forced != 0 : ctx->sample_hot_rows;   // (the last sample's verdict)
    if (repeat1 && steps1 != 7) ++ctx->stat_repeat_pieces;
    const uint64_t tile_steps = 16ull * steps1;
    const uint64_t tiles1 = (total_steps + tile_steps - 1) / tile_steps;
    const uint32_t G1 = (uint32_t)std::min<uint64_t>((uint64_t)std::min(ctx->num_cu, 256), tiles1);
    const uint64_t tpb1 = (tiles1 + G1 - 1) / G1;
    if (tpb1 > 0xFFFFull) return set_err(KPAL_E_INVALID, "quad partition: batch too large");
    // capacity (stride) of a level-1 workgroup's run of records per row: rounded up so that a unit of level 2 is a whole
    // number of KiB -- a wave-step of quad2_scatter_kernel then never straddles two units
    const uint64_t per_kib = 1024 / (S1 * 4);                                   // records per KiB: 2 (8 at k = 16)
    const uint64_t cap1 = (tpb1 + 1 + per_kib - 1) / per_kib * per_kib;     // (+ 1: the tail round of what the last tile carried over)
    if ((size_t)kQuadRowWords * 4 * G1 * cap1 > ctx->quad_pool_max && s.nchunks > 64) return kSplitBatch;
    CHK(ensure(ctx, ctx->residuals, (size_t)kQuadRowWords * 4 * G1 * cap1));
    uint32_t *pool1 = (uint32_t *)ctx->residuals.p;
    // level 2: ~4 workgroups per CU in total; workgroup (g2, c) takes `upw` of the REP x G1 units of coarse bucket c
    const uint32_t units = REP * G1;
    uint32_t G2 = std::max<uint32_t>(1, std::min<uint32_t>(units, (uint32_t)ctx->num_cu * 4 / NB1));
    const uint32_t upw = (units + G2 - 1) / G2;
    G2 = (units + upw - 1) / upw;
    const uint64_t unit_cap = cap1 * S1 * 4;                                   // bytes
    if ((uint64_t)upw * unit_cap >= (1ull << 32)) return set_err(KPAL_E_INVALID, "quad partition: batch too large");
    // level-2 tile: 16 waves x steps2 KiB of level-1 records.  A wave-step of records holds 256 item slots, filled to
    // f1 = (items of a level-1 tile) / 32768; a fine row (512 rows of 64 slots) receives its share of them.  Same queue
    // model as level 1 (the 32 fullest fine rows are left to the spill list and the hot-item table).
    constexpr int kWaves2 = 16;
    int steps2 = 2;
    if (cached) {
        steps2 = ctx->cached_steps2;
    } else {
        double all = 0.0;
        for (double v : fine) all += v;
        const double f1 = std::min(1.0, all * 16.0 * steps1 / (double)kQuadRowWords);
        static const int candidates2[] = {8, 7, 6, 4, 3, 2};
        std::vector<double> mu(fine.size() > 32 ? fine.size() - 32 : 0);
        for (int c : candidates2) {
            for (size_t b = 0; b < mu.size(); ++b) mu[b] = all > 0.0 ? fine[b] / all * (256.0 * f1) * kWaves2 * c : 0.0;
            if (quad_expected_backlog(mu, 64) <= kQuadBacklogMax) {
                steps2 = c;
                break;
            }
        }
        ctx->cached_steps1 = steps1;
        ctx->cached_steps2 = steps2;
        ctx->cached_uses = 0;
        ctx->cached_bytes = feed_bytes;
    }
    for (int c : {8, 7, 6, 4, 3, 2})
        if (c == ctx->quad_steps2_forced) steps2 = c;
    ctx->plan_strategy = KPAL_STRATEGY_PARTITION2_QUADS;
    ctx->plan_steps1 = steps1;
    ctx->plan_steps2 = steps2;
    const uint64_t tile2_bytes = (uint64_t)kWaves2 * steps2 * 1024;
    const uint64_t tiles2 = ((uint64_t)upw * unit_cap + tile2_bytes - 1) / tile2_bytes;
    const uint64_t cap2 = tiles2 + 1;                                           // rounds per level-2 workgroup: its tiles + the tail round
    CHK(ensure(ctx, ctx->keys, (size_t)512 * kQuadPackedRecordBytes * NB1 * G2 * cap2));   // level-2 records: 64 items packed into 192 bytes
    CHK(ensure(ctx, ctx->quad_meta2, (size_t)NB1 * G2 * sizeof(uint32_t)));
    // the staged forms of the histogram stage (four 8-bit counts per table entry: 4.3 GB at k = 15) reuse the level-1 pool's buffer:
    // level 2 has read it completely before the histogram kernel starts (same stream)
    CHK(ensure(ctx, ctx->residuals, std::max<size_t>((size_t)kQuadRowWords * 4 * G1 * cap1, (size_t)ctx->bins * 4 * sizeof(quad2_stage_t))));
    pool1 = (uint32_t *)ctx->residuals.p;
    uint32_t *stage = pool1;
    uint32_t *pool2 = (uint32_t *)ctx->keys.p;
    uint32_t *nrounds2 = (uint32_t *)ctx->quad_meta2.p;
    // Counts that bypass the records (TableSink, quad_kernels.hpp): classic -> atomic adds into the (zeroed) table; FRESH -> lists,
    // one segment per scatter workgroup (level 1: G1, level 2: G2 x NB1) and one shared segment for the histogram stage
    const uint32_t nseg = G1 + G2 * NB1 + 1;
    TableSink table = {(unsigned long long *)ctx->table.p, nullptr, nullptr, 0u, nullptr};
    TableSink table2 = table, table_h = table;
    if (fresh) {
        // the histogram stage's shared segment also takes every count that does not fit a staged form (>= 256 within one form and
        // piece: repeats of real genomes): room for 4 M entries (unless the tests force small segments)
        // (a feed whose sample showed hot rows sends more counts past the records -- the k-mer entries of the hot-item tables leave as
        // they age, a few hundred per workgroup and four tiles: sixteen times the room (2.7 GB at k = 15), or the piece would overflow its lists and run again)
        const uint32_t seg = (repeat1 && !ctx->direct_seg_forced) ? ctx->direct_seg * 16u : ctx->direct_seg;
        ctx->direct_seg_used = seg;
        const uint32_t seg_h = ctx->direct_seg_forced ? seg : std::max<uint32_t>(seg, 1u << 22);
        ctx->direct_seg_hist = seg_h;
        CHK(ensure(ctx, ctx->direct_list, ((size_t)(nseg - 1) * seg + seg_h) * sizeof(unsigned long long)));
        CHK(ensure(ctx, ctx->direct_meta, ((size_t)nseg + 4) * sizeof(uint32_t)));
        unsigned long long *list = (unsigned long long *)ctx->direct_list.p;
        uint32_t *counts = (uint32_t *)ctx->direct_meta.p, *overflow = counts + nseg;
        HIPCHK(hipMemsetAsync(counts, 0, ((size_t)nseg + 4) * sizeof(uint32_t), ctx->stream));
        table = TableSink{nullptr, list, counts, seg, overflow};                                             // (the kernels add their workgroup's offset)
        table2 = TableSink{nullptr, list + (size_t)G1 * seg, counts + G1, seg, overflow};
        table_h = TableSink{nullptr, list + (size_t)(nseg - 1) * seg, counts + (nseg - 1), seg_h, overflow};   // global counter
    }
#define KPAL_QUAD2_LAUNCH(S2)                                                                                                          \
    do {                                                                                                                               \
        if (repeat1)                                                                                                                   \
            LAUNCH(ctx, "quad2_scatter", (quad2_scatter_kernel<K, kWaves2, S2, true>), dim3(G2, NB1), dim3(kWaves2 * 64), (const uint32_t *)pool1, \
                   (const uint32_t *)nrounds1, G1, (uint32_t)cap1, upw, (uint32_t)tiles2, pool2, (uint32_t)cap2, nrounds2, error, table2); \
        else                                                                                                                           \
            LAUNCH(ctx, "quad2_scatter", (quad2_scatter_kernel<K, kWaves2, S2, false>), dim3(G2, NB1), dim3(kWaves2 * 64), (const uint32_t *)pool1, \
                   (const uint32_t *)nrounds1, G1, (uint32_t)cap1, upw, (uint32_t)tiles2, pool2, (uint32_t)cap2, nrounds2, error, table2); \
    } while (0)
    // (REPEAT: the level-1 instantiation with the repeat lanes' shortcut, when the sample shows hot rows -- kpal_quads.hip; the
        // seven-step tile has no such instantiation: at 128 registers the call site cost it spilled ones)
#define KPAL_QUAD1_LAUNCH(S, D)                                                                                                        \
    do {                                                                                                                               \
        if (repeat1 && S != 7)                                                                                                         \
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, S, D, TableSink, (S != 7)>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table); \
        else                                                                                                                           \
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, S, D, TableSink, false>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table);    \
    } while (0)
    DISPATCH_K_13_16(ctx->k, {
        if (steps1 == 8) KPAL_QUAD1_LAUNCH(8, KPAL_L1_DEPTH8);
        else if (steps1 == 7) KPAL_QUAD1_LAUNCH(7, 7);
        else if (steps1 == 6) KPAL_QUAD1_LAUNCH(6, 6);
        else KPAL_QUAD1_LAUNCH(3, 3);
        switch (steps2) {
        case 8: KPAL_QUAD2_LAUNCH(8); break;
        case 7: KPAL_QUAD2_LAUNCH(7); break;
        case 6: KPAL_QUAD2_LAUNCH(6); break;
        case 4: KPAL_QUAD2_LAUNCH(4); break;
        case 3: KPAL_QUAD2_LAUNCH(3); break;
        default: KPAL_QUAD2_LAUNCH(2); break;
        }

        // (packed bins: two forms per word, two workgroups per CU -- safe while a histogram workgroup's stream holds < 2^16 item slots)
        if (K >= 15 && (uint64_t)G2 * cap2 * 64 < 65536 && !ctx->quad_hist_unpacked)
            LAUNCH(ctx, "quad_hist", (quad_hist_kernel<K, TableSink, true>), dim3(512, NB1), dim3(1024), (const uint32_t *)pool2, (const uint32_t *)nrounds2,
                   G2, (uint32_t)cap2, table_h, stage);
        else
            LAUNCH(ctx, "quad_hist", (quad_hist_kernel<K, TableSink, false>), dim3(512, NB1), dim3(1024), (const uint32_t *)pool2, (const uint32_t *)nrounds2,
                   G2, (uint32_t)cap2, table_h, stage);
    });
#undef KPAL_QUAD2_LAUNCH
    // The staged forms are added to the table by quad2_finalize_kernel -- LATER: kpal_count_balance fuses Profile.balance into
    // that pass; anything else that needs the table (the next piece or feed, kpal_count_finish / _table, kpal_sync) flushes it
    // first (quad2_finalize(ctx, false)).
    ctx->finalize_pending = true;
    ctx->finalize_stage = stage;
    ctx->finalize_fresh = fresh;
    ctx->fresh_resolved = false;
    if (fresh) {
        ctx->table_zero_pending = false;   // the finalisation writes every entry
        ctx->fresh_span = s;
        ctx->direct_nseg = nseg;
    }
    return KPAL_OK;
}

// The table as every consumer other than a FRESH finalisation expects it: zeroed if nothing has been counted into it yet, the
// staged forms of the last two-level quad piece added.
int table_ready(kpal_ctx *ctx)
{
    if (ctx->table_zero_pending) {
        HIPCHK(hipMemsetAsync(ctx->table.p, 0, ctx->bins * sizeof(int64_t), ctx->stream));
        ctx->table_zero_pending = false;
    }
    return quad2_finalize(ctx, false);
}

// A pending FRESH piece: did every bypassing count fit its list segment?  (One word; the host waits for the piece's kernels --
// it would soon anyway: the finalisation is launched from the host.)  If not -- a heavily skewed piece; the lists are sized for
// the usual few hundred entries per workgroup -- the classic way after all: zero the table, count the piece AGAIN with atomic
// adds for what bypasses the records.  That re-reads the fed buffer, so kpal_count_feed_device calls this before it returns: the
// caller's buffer is free again when the feed call is back, as with every other pipeline.
int quad2_resolve_fresh(kpal_ctx *ctx)
{
    if (!ctx->finalize_pending || !ctx->finalize_fresh || ctx->fresh_resolved) return KPAL_OK;
    uint32_t overflow = 0;
    const uint32_t *word = (const uint32_t *)ctx->direct_meta.p + ctx->direct_nseg;
    HIPCHK(hipMemcpyAsync(&overflow, word, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->fresh_resolved = true;
    if (overflow) {
        ++ctx->stat_fresh_reruns;
        ctx->finalize_pending = false;
        ctx->finalize_fresh = false;
        HIPCHK(hipMemsetAsync(ctx->table.p, 0, ctx->bins * sizeof(int64_t), ctx->stream));
        const int rc = launch_partition2_quads(ctx, ctx->fresh_span, false);   // (classic: finalize_pending again, not fresh)
        if (rc != KPAL_OK) return rc == kQuadsUseChunked || rc == kSplitBatch ? set_err(KPAL_E_HIP, "two-level quad pipeline: cannot repeat a piece") : rc;
    }
    return KPAL_OK;
}

// Adds the staged forms of the last two-level quad piece to the count table (and balances the table in the same pass).
int quad2_finalize(kpal_ctx *ctx, bool balance)
{
    if (!ctx->finalize_pending) return KPAL_OK;
    CHK(quad2_resolve_fresh(ctx));
    ctx->finalize_pending = false;
    const bool fresh = ctx->finalize_fresh;
    ctx->finalize_fresh = false;
    const quad2_stage_t *stage = (const quad2_stage_t *)ctx->finalize_stage;
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    if (ctx->k == 12) {   // the one-level pipeline's staged forms (kpal_quads.hip): never FRESH (its table is zeroed by kpal_count_begin)
        if (balance) LAUNCH(ctx, "quad2_finalize_balanced", (quad2_finalize_kernel<12, true, false>), dim3(Quad2Index<12>::kSets), dim3(1024), stage, table);
        else LAUNCH(ctx, "quad2_finalize", (quad2_finalize_kernel<12, false, false>), dim3(Quad2Index<12>::kSets), dim3(1024), stage, table);
        return KPAL_OK;
    }
    DISPATCH_K_13_16(ctx->k, {
        if (balance && fresh)
            LAUNCH(ctx, "quad2_finalize_balanced", (quad2_finalize_kernel<K, true, true>), dim3(Quad2Index<K>::kSets), dim3(1024), stage, table);
        else if (balance)
            LAUNCH(ctx, "quad2_finalize_balanced", (quad2_finalize_kernel<K, true, false>), dim3(Quad2Index<K>::kSets), dim3(1024), stage, table);
        else if (fresh)
            LAUNCH(ctx, "quad2_finalize", (quad2_finalize_kernel<K, false, true>), dim3(Quad2Index<K>::kSets), dim3(1024), stage, table);
        else
            LAUNCH(ctx, "quad2_finalize", (quad2_finalize_kernel<K, false, false>), dim3(Quad2Index<K>::kSets), dim3(1024), stage, table);
        if (fresh)
            LAUNCH(ctx, "quad2_apply_list", (quad2_apply_list_kernel<K>), dim3(ctx->direct_nseg - 1 + kQuad2ListTailBlocks), dim3(256),
                   (const unsigned long long *)ctx->direct_list.p, (const uint32_t *)ctx->direct_meta.p, ctx->direct_seg_used, ctx->direct_nseg - 1,
                   ctx->direct_seg_hist, balance ? 1u : 0u, table);
    });
    return KPAL_OK;
}

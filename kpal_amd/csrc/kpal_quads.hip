// kpal_quads.hip -- launch planning of the quad record pipeline, k = 8..12 (quad_kernels.hpp): tile size from a
// sample of the row loads, one scatter launch, one histogram launch.
#include "kpal_host.hpp"

#include "quad_kernels.hpp"


// Expected number of items in the spill list of a workgroup in the steady state.  A row is a queue: Poisson(mu) items
// arrive per round, `slots` leave with the record, the rest is carried to the next round.  The single-round overflow
// E[max(X - slots, 0)] underestimates the backlog of a well-filled row (carried items arrive again: at 83 % fill of a
// 16-slot row the backlog is twice the overflow, at 95 % six times; a row whose load exceeds its slots grows without
// bound until the list is full and the slow direct path takes over -- measured 2x slower on AT-rich input with tiles
// chosen by the single-round figure).  backlog = overflow x r(fill, slots), r tabulated from a simulation of the queue
// (tools/diag/spill_queue.py).
double quad_expected_backlog(const std::vector<double> &mu, int slots)
{
    static const double rho_grid[10] = {0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9, 0.925, 0.95, 0.975};
    static const double ratio[4][10] = {
        {1.02, 1.08, 1.25, 1.43, 1.70, 2.19, 3.21, 4.24, 6.34, 12.7},   // 16 slots
        {1.00, 1.01, 1.08, 1.17, 1.33, 1.65, 2.34, 3.05, 4.52, 8.9},    // 32
        {1.00, 1.00, 1.01, 1.04, 1.12, 1.30, 1.75, 2.24, 3.25, 6.35},   // 64
        {1.00, 1.00, 1.00, 1.02, 1.02, 1.10, 1.36, 1.68, 2.37, 4.55}};  // 128
    const int ti = slots <= 24 ? 0 : (slots <= 32 ? 1 : (slots <= 64 ? 2 : 3));   // (20 slots: the 16-slot row of the table, on the safe side)
    double total = 0.0;
    // mu is sorted: rows whose load lies within 1 % of each other are evaluated once, at their mid-point (this runs
    // on the host inside every large feed: 2048 Poisson tails per candidate cost 0.6 ms of a 12 ms step)
    for (size_t at = 0; at < mu.size();) {
        size_t end = at + 1;
        while (end < mu.size() && mu[end] <= mu[at] * 1.01) ++end;
        const double m = 0.5 * (mu[at] + mu[end - 1]), weight = (double)(end - at);
        at = end;
        if (m <= 0.0) continue;
        const double rho = m / slots;
        if (rho >= 0.995) {   // the row cannot keep up
            total += 1e6 * weight;
            continue;
        }
        // E[max(X - c, 0)] = sum_{x > c} (x - c) p(x); p by recurrence from p(0) = exp(-m)
        double p = std::exp(-m), acc = 0.0;
        const int upto = (int)(m + 12.0 * std::sqrt(m) + 40.0);
        for (int x = 1; x <= upto; ++x) {
            p *= m / x;
            if (x > slots) acc += (x - slots) * p;
        }
        double r = 1.0;
        if (rho >= rho_grid[9]) {
            r = ratio[ti][9];
            acc = std::max(acc * r, m / (2.0 * (slots - m)));   // heavy traffic
            r = 1.0;
        } else if (rho > rho_grid[0]) {
            int j = 0;
            while (rho > rho_grid[j + 1]) ++j;
            const double f = (rho - rho_grid[j]) / (rho_grid[j + 1] - rho_grid[j]);
            r = ratio[ti][j] + f * (ratio[ti][j + 1] - ratio[ti][j]);
        }
        total += acc * r * weight;
    }
    return total;
}

// Tile size of a quad scatter from the row loads of a ~1/64 sample of the feed (quad_sample_kernel): the largest
// candidate (wave-steps per wave per tile) whose expected steady-state backlog stays well inside the spill list.
// Returns kQuadsUseChunked (AUTO only) when a few rows hold more than 1.5 % of all items.
int quad_choose_steps(kpal_ctx *ctx, const Span &s, uint32_t *load, int buckets, int slots, int waves, const int *candidates,
                             size_t n_candidates, int *steps_out, std::vector<double> *fine_per_step)
{
    const int extra = fine_per_step ? 512 : 0;   // (two-level path: the sample also returns the loads of the 512 fine rows)
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint32_t sample_steps = 4;                                       // per wave: 32 KiB per workgroup
    const uint64_t want = std::max<uint64_t>(1, total_steps / (64ull * 8 * sample_steps));   // ~1/64 of the input
    const uint32_t groups = (uint32_t)std::min<uint64_t>(want, 1024);
    const uint64_t stride = std::max<uint64_t>(8 * sample_steps, total_steps / groups);
    HIPCHK(hipMemsetAsync(load, 0, (size_t)(buckets + extra + 1) * sizeof(uint32_t), ctx->stream));   // (+ 1: the items of repeat lanes)
    DISPATCH_K_8_16(ctx->k, LAUNCH(ctx, "quad_sample", (quad_sample_kernel<K>), dim3(groups), dim3(512), s, stride, sample_steps, load));
    std::vector<uint32_t> h((size_t)(buckets + extra + 1));
    HIPCHK(hipMemcpyAsync(h.data(), load, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const double sampled_steps = (double)std::min<uint64_t>((uint64_t)groups * 8 * sample_steps, total_steps);
    // the 32 fullest rows are left out: a handful of very hot rows (poly-A, an adapter shared by every read) cannot be
    // helped by smaller tiles -- their items are counted in the workgroup's hot-item table instead
    const bool verbose = ctx->quad_verbose;
    double budget = kQuadBacklogMax;
    std::vector<double> per_step((size_t)buckets);
    for (int b = 0; b < buckets; ++b) per_step[b] = h[b] / sampled_steps;   // items per row per wave-step
    if (fine_per_step) {
        fine_per_step->resize(512);
        for (int b = 0; b < 512; ++b) (*fine_per_step)[b] = h[(size_t)buckets + b] / sampled_steps;
        std::sort(fine_per_step->begin(), fine_per_step->end());
    }
    std::sort(per_step.begin(), per_step.end());
    // When those hot rows hold more than 1.5 % of all items (reads that share an adapter / primer prefix, several
    // per cent of low-complexity reads) the slow path of the scatter would run in nearly every placement step --
    // measured 20-50x slower on a 20..40-base prefix shared by all reads.  The round-1 pipelines take such a feed
    // in their stride (their buckets simply own more chunks), so AUTO hands the feed over; an explicitly chosen quad
    // strategy stays (tests, A/B).
    {
        double all = 0.0, hot = 0.0;
        const double median = per_step[(size_t)buckets / 2];
        for (int b = 0; b < buckets; ++b) all += per_step[b];
        for (int b = buckets - 32; b < buckets; ++b) hot += std::max(0.0, per_step[b] - median);
        // ... unless nearly all of that excess sits in one to three rows (a homopolymer run, a two-letter repeat): then a
        // wave's hot items are all the same, one ballot round counts them into the workgroup's hot-item table, and the
        // quad path is the faster one (homopolymer feed: 520 vs 230 Gbases/s).  A shared prefix spreads over a dozen rows.
        double top3 = 0.0;
        for (int b = buckets - 3; b < buckets; ++b) top3 += std::max(0.0, per_step[b] - median);
        const bool concentrated = top3 >= 0.8 * hot;
        if (verbose)
            fprintf(stderr, "[kpal quad] sample: %.2f %% of the items are the excess of the 32 fullest rows, %.0f %% of it in three rows\n",
                    all > 0.0 ? 100.0 * hot / all : 0.0, hot > 0.0 ? 100.0 * top3 / hot : 0.0);
        if (ctx->strategy == KPAL_STRATEGY_AUTO && all > 0.0 && hot > 0.015 * all && !concentrated) return kQuadsUseChunked;
        // hot rows fill the spill list first (their excess is carried every round before it is counted directly): the
        // ordinary rows then get a quarter of the list (k = 13, 2 % low-complexity reads: level 1 0.55 instead of 2.9 ms)
        if (all > 0.0 && hot > 0.003 * all) budget = kQuadBacklogMax / 4;
        // the scatter takes its REPEAT instantiation when the sample holds repeat lanes (their items are not in the row loads: the
        // scatter sends them past the rows) or hot rows of another kind
        const double repeats = h[(size_t)(buckets + extra)] / sampled_steps;
        ctx->sample_hot_rows = (all > 0.0 && hot > 0.003 * all) || repeats > 0.001 * (all + repeats);
    }
    per_step.resize((size_t)buckets - 32);
    std::vector<double> mu(per_step.size());
    *steps_out = candidates[n_candidates - 1];
    for (size_t ci = 0; ci < n_candidates; ++ci) {
        const int c = candidates[ci];
        for (size_t b = 0; b < mu.size(); ++b) mu[b] = per_step[b] * waves * c;
        const double backlog = quad_expected_backlog(mu, slots);
        if (verbose) fprintf(stderr, "[kpal quad] sample: %d steps per wave -> expected backlog %.0f items (fullest row %.1f of %d)\n", c, backlog, mu.back(), slots);
        if (backlog <= budget) {                                           // list: 2048 entries
            *steps_out = c;
            break;
        }
    }
    return KPAL_OK;
}

// Partition of quads into aligned records, k = 8..12 (quad_kernels.hpp): one workgroup per CU scatters,
// one workgroup per bucket histograms.  pool[bucket][workgroup][round] holds one record per flush round.
int launch_partition_quads(kpal_ctx *ctx, const Span &s)
{
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (total_steps == 0) return KPAL_OK;
    const int buckets = ctx->k == 12 ? QuadCfg<12>::kBuckets : 512;                                     // ROWS of the scatter
    const int slots = ctx->k == 12 ? QuadCfg<12>::kItems : kQuadRowWords / buckets;                     // items a row holds
    CHK(ensure(ctx, ctx->quad_meta, ((size_t)ctx->num_cu + 4 + 2048 + 512) * sizeof(uint32_t)));
    uint32_t *nrounds = (uint32_t *)ctx->quad_meta.p;
    uint32_t *error = nrounds + ctx->num_cu;
    uint32_t *load = error + 4;
    if (!ctx->quad_error_word) {
        HIPCHK(hipMemsetAsync(error, 0, 4 * sizeof(uint32_t), ctx->stream));
        ctx->quad_error_word = error;
    }
    // ---- tile size.  A tile of 16 waves x STEPS wave-steps brings ~0.119 x 16 x STEPS items per 16-slot row at k = 12 when
    // the k-mers are uniform (7 steps: 13.3 of 16, records 83 % full, ~3 % of the items spill to the list); the row loads of a
    // 1/64 sample say what THIS input brings.  The largest STEPS whose expected overflow per round stays well inside the
    // spill list is used (KPAL_QUAD_STEPS forces one: A/B timing, tests).  16 waves = four per SIMD with 128 registers each
    // (8 record vectors + 7 prefetched chunks live): measured 3 % faster than 8 waves x 13 steps and the records are fuller.
    static const int candidates[] = {8, 7, 6, 4, 3, 2, 1};
    constexpr int waves = 16;
    int steps = 0;
    for (int c : candidates)
        if (c == ctx->quad_steps_forced) steps = c;
    // a later feed of the same count of about the same size reuses the tile size of the sampled one (re-sampled every 16 feeds)
    const size_t feed_bytes = (size_t)(s.hi - s.emit_from);
    if (!steps && ctx->cached_steps1 && ctx->cached_uses < 16 && feed_bytes <= 2 * ctx->cached_bytes && 2 * feed_bytes >= ctx->cached_bytes) {
        steps = ctx->cached_steps1;
        ++ctx->cached_uses;
    }
    bool repeat = true;                          // (forced tile sizes: no sample -- the instantiation that knows repeats)
    if (steps && ctx->cached_steps1 == steps && !ctx->quad_steps_forced) repeat = ctx->sample_hot_rows;
    if (!steps) {
        const int rc = quad_choose_steps(ctx, s, load, buckets, slots, waves, candidates, sizeof(candidates) / sizeof(candidates[0]), &steps);
        if (rc != KPAL_OK) return rc;
        repeat = ctx->sample_hot_rows;
        ctx->cached_steps1 = steps;
        ctx->cached_uses = 0;
        ctx->cached_bytes = feed_bytes;
    }
    if (ctx->quad_repeat_forced >= 0) repeat = ctx->quad_repeat_forced != 0;
    // (there is no REPEAT instantiation of the 7-step tile -- it would spill registers: the 6-step tile, unless the size is forced)
    if (repeat && steps == 7 && !ctx->quad_steps_forced) steps = 6;
    if (repeat && steps != 7) ++ctx->stat_repeat_pieces;
    ctx->plan_strategy = KPAL_STRATEGY_PARTITION_QUADS;
    ctx->plan_steps1 = steps;
    ctx->plan_steps2 = 0;
    const uint64_t tile_steps = (uint64_t)waves * steps;
    const uint64_t tiles = (total_steps + tile_steps - 1) / tile_steps;
    const uint32_t G = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cu, tiles);
    const uint64_t tpb = (tiles + G - 1) / G;          // tiles (= flush rounds) per workgroup
    if (tpb > 0xFFFFFFull) return set_err(KPAL_E_INVALID, "quad partition: batch too large");
    const size_t pool_bytes = (size_t)kQuadRowWords * 4 * G * tpb;   // every round writes all rows: 128 KiB per workgroup
    if (pool_bytes > ctx->quad_pool_max && s.nchunks > 64) return kSplitBatch;   // (heavily skewed 16 GiB piece: small tiles)
    CHK(ensure(ctx, ctx->keys, pool_bytes));
    uint32_t *pool = (uint32_t *)ctx->keys.p;
    const TableOnly table = {(unsigned long long *)ctx->table.p};   // counts that bypass the records: atomics into the (zeroed) table
    // k = 12: every table entry receives FOUR adds from the histogram stage (one per form, from four different workgroups): 67 M
    // global atomics per piece, 0.25 ms of the 3.7 ms histogram (A/B `hist_nomerge`) -- and Profile.balance then reads and writes the
    // table once more.  As on the two-level path the forms are STAGED instead (8-bit counts, 64 MiB; quad2_index.hpp with the
    // 11-bit bucket read as coarse : fine) and quad2_finalize_kernel gathers the four of every entry -- and balances in the same
    // pass when kpal_count_balance asks (the pending finalisation: kpal_quads2.hip).  A form count >= 256 goes to the table
    // directly; the mean per form is bytes / (4 x 4^12), so pieces whose mean could pass 240 (16 GB of unbroken sequence) keep the
    // atomic merge.  KPAL_K12_STAGED=0: the atomic merge everywhere (A/B, tests).
    static const bool allow_staged = [] { const char *e = getenv("KPAL_K12_STAGED"); return !e || atoi(e) != 0; }();
    uint32_t *stage = nullptr;
    if (ctx->k == 12 && allow_staged && (double)feed_bytes <= 240.0 * 4.0 * (double)ctx->bins) {
        CHK(ensure(ctx, ctx->residuals, (size_t)ctx->bins * 4 * sizeof(quad2_stage_t)));
        stage = (uint32_t *)ctx->residuals.p;
    }
    // (input chunks are requested S steps ahead; at k <= 11 with eight steps that ring costs the registers the kernel does not have --
    // 12 spilled, and a kernel that uses scratch memory at all ran ~10 % slower in same-box comparisons -- so four steps ahead there;
    // k = 12 fits its 128 registers either way and a four-step ring changed nothing: 7.44 vs 7.40 ms)
    // REPEAT: the instantiation that sends the repeat lanes of low-complexity sequence straight to the hot-item table (quad_kernels.hpp);
    // taken when the sample shows hot rows (KPAL_QUAD_REPEAT=0 / 1 forces one: A/B, tests).  Its call site costs registers: a
    // four-step input ring in the 8-step tile (and no such instantiation of the 7-step tile: it would spill).
#define KPAL_QUAD_LAUNCH(S)                                                                                                  \
    do {                                                                                                                     \
        if (repeat)                                                                                                          \
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, S, (S == 8 ? 4 : S), TableOnly, (S != 7)>), dim3(G), dim3(1024), s, tpb, pool, (uint32_t)tpb, \
                   nrounds, error, table);                                                                                   \
        else                                                                                                                 \
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, S, (S == 8 && K != 12 ? 4 : S), TableOnly, false>), dim3(G), dim3(1024), s, tpb, pool,    \
                   (uint32_t)tpb, nrounds, error, table);                                                                    \
    } while (0)
    DISPATCH_K_8_12(ctx->k, {
        switch (steps) {
        case 8: KPAL_QUAD_LAUNCH(8); break;
        case 7: KPAL_QUAD_LAUNCH(7); break;
        case 4: KPAL_QUAD_LAUNCH(4); break;
        case 3: KPAL_QUAD_LAUNCH(3); break;
        case 2: KPAL_QUAD_LAUNCH(2); break;
        case 1: KPAL_QUAD_LAUNCH(1); break;
        default: KPAL_QUAD_LAUNCH(6); break;
        }
        LAUNCH(ctx, "quad_hist", (quad_hist_kernel<K>), dim3(QuadCfg<K>::kHistBuckets), dim3(1024), (const uint32_t *)pool,
               (const uint32_t *)nrounds, G, (uint32_t)tpb, table, stage);
    });
#undef KPAL_QUAD_LAUNCH
    if (stage) {   // added to the table (and balanced) by quad2_finalize_kernel<12, ...> when something needs the table
        ctx->finalize_pending = true;
        ctx->finalize_stage = stage;
        ctx->finalize_fresh = false;
        ctx->fresh_resolved = false;
    }
    if (ctx->quad_verbose) {   // diagnostics: tile size chosen, tiles abandoned to the direct path
        uint32_t st[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(st, error, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[kpal quad] k=%d steps/wave/tile=%d tiles=%llu workgroups=%u hot-table entries used so far=%u\n", ctx->k, steps,
                (unsigned long long)tiles, G, st[1]);
    }
    return KPAL_OK;
}


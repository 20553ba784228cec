// kpal_hex.hip -- launch planning of the HEX pipeline, k = 12 (hex_kernels.hpp): tile size from a sample of the row loads, one
// scatter launch, one histogram launch; the staged forms are added to the table by hex_finalize when something needs the table.
#include "kpal_host.hpp"

#include "hex_kernels.hpp"

// Tile size (wave-steps of 3 KiB per wave and tile) from the row loads of a ~1/64 sample: see quad_choose_steps.  The second items
// of partial groups ride in the spill list as well (hex_scatter_kernel): their expected number per tile comes off the backlog budget.
static int hex_choose_steps(kpal_ctx *ctx, const Span &s, uint32_t *load, int *steps_out)
{
    static const int candidates[] = {4, 3, 2, 1};
    constexpr int buckets = HexIndex::kRows, slots = kHexRowItems, waves = kHexWaves;
    const uint64_t total_steps = (s.nchunks + kHexStepChunks - 1) / kHexStepChunks;
    const uint32_t sample_steps = 2;                                       // per wave: 48 KiB per workgroup
    const uint64_t want = std::max<uint64_t>(1, total_steps / (64ull * 8 * sample_steps));   // ~1/64 of the input
    const uint32_t groups = (uint32_t)std::min<uint64_t>(want, 1024);
    const uint64_t stride = std::max<uint64_t>(8 * sample_steps, total_steps / groups);
    HIPCHK(hipMemsetAsync(load, 0, (size_t)buckets * sizeof(uint32_t), ctx->stream));
    LAUNCH(ctx, "hex_sample", hex_sample_kernel, dim3(groups), dim3(512), s, stride, sample_steps, load);
    std::vector<uint32_t> h((size_t)buckets);
    HIPCHK(hipMemcpyAsync(h.data(), load, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const double sampled_steps = (double)std::min<uint64_t>((uint64_t)groups * 8 * sample_steps, total_steps);
    std::vector<double> per_step((size_t)buckets);
    for (int b = 0; b < buckets; ++b) per_step[b] = h[b] / sampled_steps;   // items per row per wave-step
    std::sort(per_step.begin(), per_step.end());
    double budget = kQuadBacklogMax;
    double all = 0.0, hot = 0.0, top3 = 0.0;
    const double median = per_step[(size_t)buckets / 2];
    for (int b = 0; b < buckets; ++b) all += per_step[b];
    for (int b = buckets - 32; b < buckets; ++b) hot += std::max(0.0, per_step[b] - median);
    for (int b = buckets - 3; b < buckets; ++b) top3 += std::max(0.0, per_step[b] - median);
    const bool concentrated = top3 >= 0.8 * hot;
    if (ctx->quad_verbose)
        fprintf(stderr, "[kpal hex] sample: %.2f %% of the items are the excess of the 32 fullest rows, %.0f %% of it in three rows\n",
                all > 0.0 ? 100.0 * hot / all : 0.0, hot > 0.0 ? 100.0 * top3 / hot : 0.0);
    if (ctx->strategy == KPAL_STRATEGY_AUTO && all > 0.0 && hot > 0.015 * all && !concentrated) return kQuadsUseChunked;
    if (all > 0.0 && hot > 0.003 * all) budget = kQuadBacklogMax / 4;
    per_step.resize((size_t)buckets - 32);
    std::vector<double> mu(per_step.size());
    *steps_out = candidates[3];
    for (int c : candidates) {
        for (size_t b = 0; b < mu.size(); ++b) mu[b] = per_step[b] * waves * c;
        // ~2.7 % of the items of 150-base reads are second items (hex_index.hpp): they enter the list whatever the rows hold
        const double seconds = 0.027 * all * waves * c;
        const double backlog = quad_expected_backlog(mu, slots) + seconds;
        if (ctx->quad_verbose) fprintf(stderr, "[kpal hex] sample: %d steps per wave -> expected backlog %.0f items (fullest row %.1f of %d)\n", c, backlog, mu.back(), slots);
        if (backlog <= budget) {
            *steps_out = c;
            break;
        }
    }
    return KPAL_OK;
}

int launch_partition_hex(kpal_ctx *ctx, const Span &s)
{
    if (ctx->k != 12) return set_err(KPAL_E_INVALID, "the hex pipeline is for k = 12 (k=%d)", ctx->k);
    const uint64_t total_steps = (s.nchunks + kHexStepChunks - 1) / kHexStepChunks;
    if (total_steps == 0) return KPAL_OK;
    CHK(ensure(ctx, ctx->quad_meta, ((size_t)ctx->num_cu + 4 + 2048 + 512) * sizeof(uint32_t)));
    uint32_t *nrounds = (uint32_t *)ctx->quad_meta.p;
    uint32_t *error = nrounds + ctx->num_cu;
    uint32_t *load = error + 4;
    if (!ctx->quad_error_word) {
        HIPCHK(hipMemsetAsync(error, 0, 4 * sizeof(uint32_t), ctx->stream));
        ctx->quad_error_word = error;
    }
    int steps = 0;
    for (int c : {4, 3, 2, 1})
        if (c == ctx->quad_steps_forced) steps = c;
    const size_t feed_bytes = (size_t)(s.hi - s.emit_from);
    if (!steps && ctx->cached_steps1 && ctx->cached_uses < 16 && feed_bytes <= 2 * ctx->cached_bytes && 2 * feed_bytes >= ctx->cached_bytes) {
        steps = ctx->cached_steps1;
        ++ctx->cached_uses;
    }
    if (!steps) {
        const int rc = hex_choose_steps(ctx, s, load, &steps);
        if (rc != KPAL_OK) return rc;
        ctx->cached_steps1 = steps;
        ctx->cached_uses = 0;
        ctx->cached_bytes = feed_bytes;
    }
    ctx->plan_strategy = KPAL_STRATEGY_PARTITION_HEX;
    ctx->plan_steps1 = steps;
    ctx->plan_steps2 = 0;
    const uint64_t tile_steps = (uint64_t)kHexWaves * steps;
    const uint64_t tiles = (total_steps + tile_steps - 1) / tile_steps;
    const uint32_t G = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cu, tiles);
    const uint64_t tpb = (tiles + G - 1) / G;          // tiles (= flush rounds) per workgroup
    if (tpb > 0xFFFFFFull) return set_err(KPAL_E_INVALID, "hex partition: batch too large");
    const size_t pool_bytes = (size_t)kQuadRowWords * 4 * G * tpb;   // every round writes all rows: 128 KiB per workgroup
    if (pool_bytes > ctx->quad_pool_max && s.nchunks > kHexStepChunks) return kSplitBatch;
    CHK(ensure(ctx, ctx->keys, pool_bytes));
    uint32_t *pool = (uint32_t *)ctx->keys.p;
    const TableOnly table = {(unsigned long long *)ctx->table.p};
    // staged forms: six planes of 8-bit counts in table order; a form count >= 256 goes to the table directly, so pieces whose
    // mean count per form could pass ~240 (24 GB of unbroken sequence) keep the atomic merge (KPAL_K12_STAGED=0: everywhere)
    static const bool allow_staged = [] { const char *e = getenv("KPAL_K12_STAGED"); return !e || atoi(e) != 0; }();
    hex_stage_t *stage = nullptr;
    if (allow_staged && (double)feed_bytes <= 240.0 * 6.0 * (double)ctx->bins) {
        CHK(ensure(ctx, ctx->residuals, (size_t)ctx->bins * HexIndex::kForms * sizeof(hex_stage_t)));
        stage = (hex_stage_t *)ctx->residuals.p;
    }
#define KPAL_HEX_LAUNCH(S, D) LAUNCH(ctx, "hex_scatter", (hex_scatter_kernel<S, D>), dim3(G), dim3(1024), s, tpb, pool, (uint32_t)tpb, nrounds, error, table)
    switch (steps) {
    case 4: KPAL_HEX_LAUNCH(4, 2); break;
    case 3: KPAL_HEX_LAUNCH(3, 3); break;
    case 2: KPAL_HEX_LAUNCH(2, 2); break;
    default: KPAL_HEX_LAUNCH(1, 1); break;
    }
#undef KPAL_HEX_LAUNCH
    if (stage)
        LAUNCH(ctx, "hex_hist", (hex_hist_kernel<true>), dim3(HexIndex::kRows), dim3(1024), (const uint32_t *)pool, (const uint32_t *)nrounds, G, (uint32_t)tpb, table, stage);
    else
        LAUNCH(ctx, "hex_hist", (hex_hist_kernel<false>), dim3(HexIndex::kRows), dim3(1024), (const uint32_t *)pool, (const uint32_t *)nrounds, G, (uint32_t)tpb, table, stage);
    if (stage) {
        ctx->finalize_pending = true;
        ctx->finalize_hex = true;
        ctx->finalize_stage = stage;
        ctx->finalize_fresh = false;
        ctx->fresh_resolved = false;
    }
    if (ctx->quad_verbose) {
        uint32_t st[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(st, error, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[kpal hex] steps/wave/tile=%d tiles=%llu workgroups=%u hot-table entries=%u spilled=%u unlisted=%u\n", steps, (unsigned long long)tiles, G, st[1], st[2], st[3]);
    }
    return KPAL_OK;
}

// the staged forms of the last hex piece into the table (and Profile.balance, klib.py:285-298, behind it when asked)
int hex_finalize(kpal_ctx *ctx, bool balance)
{
    ctx->finalize_pending = false;
    ctx->finalize_hex = false;
    LAUNCH(ctx, "hex_finalize", hex_finalize_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), (const hex_stage_t *)ctx->finalize_stage, (unsigned long long *)ctx->table.p);
    if (balance) return launch_balance(ctx, ctx->k, (const int64_t *)ctx->table.p, (int64_t *)ctx->table.p);
    return KPAL_OK;
}

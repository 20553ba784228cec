// kpal_vec.hip -- everything of the C-ABI that works on count VECTORS: balance, split, strand balance, pair
// distances, distance matrices (register / LDS tiles, fp64 Gram on the matrix cores), the ProfileDistance option
// pipeline, profile summaries, merge and shrink.
#include "kpal_host.hpp"

#include "vec_kernels.hpp"
#include "matrix_all_kernels.hpp"
#include "gram_kernels.hpp"
#include "option_kernels.hpp"
#include "stat_kernels.hpp"

KPAL_API uint64_t kpal_reverse_complement(uint64_t number, int k)
{
    if (k < 1 || k > 32) return 0;
    return revcomp(number, k);
}

// out[i] = in[i] + in[rc(i)]; in == out allowed.  LDS-tiled for k >= 6, pairwise kernels below that.
// The canonical tile pairs of the LDS-tiled balance family (balance_tiled_kernel, pair_distance_balanced_kernel; k >= 6): the
// tiles M of k - 6 digits with M <= rc(M), as a device list in the order the persistent workgroups take them, and a grid that
// gives every workgroup the same number of pairs.
//   ORDER (k >= 13).  A tile's 64 runs lie 4^(k-3) entries apart -- 64 pages whatever M is -- and the partner's page numbers are
// the reverse complement of M's LOW digits: with M counting up, every workgroup in flight had 64 partner pages of its own and the
// balance ran at 2.9 TB/s (k = 15) -- address translation, as in quad2_finalize_kernel.  So the sequence runs through M with the
// bits that are page bits on NEITHER side (M bits 2 md - 12 .. 11: index bits below 18 here and in the partner) fastest: tiles
// worked on at the same time share their pages on both sides.
static int canon_tiles(kpal_ctx *ctx, int k, const uint32_t **list, uint32_t *count, unsigned *grid)
{
    const int md = k - 6;
    if (ctx->canon_k != k) {
        const uint64_t nM = 1ULL << (2 * md);
        const int lo = 2 * md - 12 > 0 ? 2 * md - 12 : 0, hi = 2 * md - 1 < 11 ? 2 * md - 1 : 11;
        const int nn = hi - lo + 1;
        auto tile_of = [&](uint64_t m) -> uint64_t {
            if (lo == 0 || nn <= 0) return m;
            return ((m & ((1ULL << nn) - 1ULL)) << lo) | ((m >> nn) & ((1ULL << lo) - 1ULL)) | ((m >> (nn + lo)) << (nn + lo));
        };
        std::vector<uint32_t> host;
        host.reserve((size_t)(nM / 2 + 1024));
        for (uint64_t m = 0; m < nM; ++m) {
            const uint64_t M = tile_of(m);
            if (md == 0 || M <= kpal_reverse_complement(M, md)) host.push_back((uint32_t)M);
        }
        CHK(ensure(ctx, ctx->canon, host.size() * sizeof(uint32_t)));
        HIPCHK(hipStreamSynchronize(ctx->stream));   // (a previous list may still be read; and `canon_host` is the copy's source)
        ctx->canon_host.swap(host);
        HIPCHK(hipMemcpyAsync(ctx->canon.p, ctx->canon_host.data(), ctx->canon_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));   // (once per k; the list is then read by launches on either of the context's streams)
        ctx->canon_k = k;
    }
    const uint32_t n = (uint32_t)ctx->canon_host.size();
    const uint32_t slots = (uint32_t)ctx->num_cu * 2;                  // two 66 KiB workgroups per CU
    const uint32_t rounds = (n + slots - 1) / slots;
    *list = (const uint32_t *)ctx->canon.p;
    *count = n;
    *grid = (n + rounds - 1) / rounds;                                  // every workgroup `rounds` pairs (the last ones one fewer)
    return KPAL_OK;
}

int launch_balance(kpal_ctx *ctx, int k, const int64_t *in, int64_t *out)
{
    const uint64_t n = 1ULL << (2 * k);
    if (k >= 6) {
        const uint32_t *canon = nullptr;
        uint32_t ncanon = 0;
        unsigned grid = 1;
        CHK(canon_tiles(ctx, k, &canon, &ncanon, &grid));
        LAUNCH(ctx, "balance_tiled", balance_tiled_kernel, dim3(grid), dim3(1024), in, out, k, canon, ncanon);
    } else if (in == out) {
        LAUNCH(ctx, "balance_inplace", balance_inplace_kernel, dim3(stream_grid(ctx, n)), dim3(256), out, k, n);
    } else {
        LAUNCH(ctx, "balance_oop", balance_oop_kernel, dim3(stream_grid(ctx, n)), dim3(256), in, out, k, n);
    }
    return KPAL_OK;
}

KPAL_API int kpal_balance_device(kpal_ctx *ctx, int k, int64_t *dev_inout)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!dev_inout) return set_err(KPAL_E_INVALID, "dev_inout is NULL");
    if (ctx->counting && dev_inout == (int64_t *)ctx->table.p && k == ctx->k) return kpal_count_balance(ctx);   // (fuses with a pending finalisation)
    return launch_balance(ctx, k, dev_inout, dev_inout);
}

KPAL_API int kpal_balance(kpal_ctx *ctx, int k, int64_t *host_inout)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!host_inout) return set_err(KPAL_E_INVALID, "host_inout is NULL");
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_inout, n * 8, hipMemcpyHostToDevice, ctx->stream));
    CHK(kpal_balance_device(ctx, k, (int64_t *)ctx->scratch[0].p));
    HIPCHK(hipMemcpyAsync(host_inout, ctx->scratch[0].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_split(kpal_ctx *ctx, int k, const int64_t *host_counts, int64_t *host_forward,
                        int64_t *host_reverse, uint64_t *n_out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!host_counts || !host_forward || !host_reverse) return set_err(KPAL_E_INVALID, "NULL pointer");
    const uint64_t n = 1ULL << (2 * k);
    const uint64_t pal = (k % 2 == 0) ? (1ULL << k) : 0ULL;   // 4^(k/2) palindromes for even k
    const uint64_t m = (n + pal) / 2;
    const uint32_t nseg = (uint32_t)((n + kSplitSeg - 1) / kSplitSeg);
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], m * 8));
    CHK(ensure(ctx, ctx->scratch[2], m * 8));
    CHK(ensure(ctx, ctx->scratch[3], (size_t)nseg * 16));
    int64_t *dc = (int64_t *)ctx->scratch[0].p;
    uint32_t *dcount = (uint32_t *)ctx->scratch[3].p;
    uint64_t *doffs = (uint64_t *)((uint8_t *)ctx->scratch[3].p + (size_t)nseg * 4 + ((size_t)nseg * 4) % 8);
    HIPCHK(hipMemcpyAsync(dc, host_counts, n * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, "split_count", split_count_kernel, dim3(nseg), dim3(256), k, n, dcount);
    std::vector<uint32_t> hc(nseg);
    HIPCHK(hipMemcpyAsync(hc.data(), dcount, (size_t)nseg * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> ho(nseg);
    uint64_t run = 0;
    for (uint32_t i = 0; i < nseg; ++i) {
        ho[i] = run;
        run += hc[i];
    }
    if (run != m) return set_err(KPAL_E_HIP, "split: canonical count %llu != expected %llu", (unsigned long long)run, (unsigned long long)m);
    HIPCHK(hipMemcpyAsync(doffs, ho.data(), (size_t)nseg * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, "split_write", split_write_kernel, dim3(nseg), dim3(256), (const int64_t *)dc, k, n,
           (const uint64_t *)doffs, (int64_t *)ctx->scratch[1].p, (int64_t *)ctx->scratch[2].p);
    HIPCHK(hipMemcpyAsync(host_forward, ctx->scratch[1].p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(host_reverse, ctx->scratch[2].p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n_out) *n_out = m;
    return KPAL_OK;
}

// Reduce `nq` groups of `nblocks` partials and fetch them.
static int finish_partials(kpal_ctx *ctx, uint32_t nq, uint32_t nblocks, std::vector<Partial> &out, bool allreduce = false)
{
    CHK(ensure(ctx, ctx->result, (size_t)nq * sizeof(Partial)));
    LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(nq), dim3(256), (const Partial *)ctx->partials.p,
           nblocks, (Partial *)ctx->result.p);
    if (allreduce) CHK(comm_allreduce_partials(ctx, ctx->result.p, nq));   // bin-range shards: the sums and counts of all ranks
    out.resize(nq);
    HIPCHK(hipMemcpyAsync(out.data(), ctx->result.p, (size_t)nq * sizeof(Partial), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

static double finish_value(int metric, const Partial &p, int64_t *aux)
{
    if (metric == KPAL_EUCLIDEAN) {
        if (aux) *aux = (int64_t)p.m;
        return std::sqrt((double)(int64_t)p.m);  // metrics.py:46: np.sqrt(np.dot(v, v))
    }
    if (aux) *aux = (int64_t)p.m;
    return p.s / (double)(p.m + 1ULL);  // metrics.py:123
}

KPAL_API int kpal_strand_balance(kpal_ctx *ctx, int k, const int64_t *host_counts, int pairwise, double *out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!host_counts || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (pairwise != KPAL_PAIRWISE_PROD && pairwise != KPAL_PAIRWISE_SUM) return set_err(KPAL_E_INVALID, "pairwise must be prod or sum");
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_counts, n * 8, hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = k >= 6 ? (1u << (2 * (k - 6))) : stream_grid(ctx, n);
    CHK(ensure(ctx, ctx->partials, (size_t)grid * sizeof(Partial)));
    const int64_t *dc = (const int64_t *)ctx->scratch[0].p;
    Partial *pp = (Partial *)ctx->partials.p;
    if (k >= 6) {
        if (pairwise == KPAL_PAIRWISE_PROD) LAUNCH(ctx, "strand_balance_tiled", (strand_balance_tiled_kernel<0>), dim3(grid), dim3(1024), dc, k, pp);
        else LAUNCH(ctx, "strand_balance_tiled", (strand_balance_tiled_kernel<1>), dim3(grid), dim3(1024), dc, k, pp);
    } else {
        if (pairwise == KPAL_PAIRWISE_PROD) LAUNCH(ctx, "strand_balance", (strand_balance_kernel<0>), dim3(grid), dim3(256), dc, k, n, pp);
        else LAUNCH(ctx, "strand_balance", (strand_balance_kernel<1>), dim3(grid), dim3(256), dc, k, n, pp);
    }
    std::vector<Partial> res;
    CHK(finish_partials(ctx, 1, grid, res));
    *out = finish_value(pairwise, res[0], nullptr);
    return KPAL_OK;
}

template <typename T>
static int pair_distance_launch(kpal_ctx *ctx, size_t n, const T *dl, const T *dr, int metric, double *out, int64_t *aux)
{
    const unsigned grid = stream_grid(ctx, (n + 1) / 2);
    CHK(ensure(ctx, ctx->partials, (size_t)grid * sizeof(Partial)));
    Partial *pp = (Partial *)ctx->partials.p;
    if (metric == KPAL_PAIRWISE_PROD) LAUNCH(ctx, "pair_distance", (pair_distance_kernel<0, T>), dim3(grid), dim3(256), dl, dr, (uint64_t)n, pp);
    else if (metric == KPAL_PAIRWISE_SUM) LAUNCH(ctx, "pair_distance", (pair_distance_kernel<1, T>), dim3(grid), dim3(256), dl, dr, (uint64_t)n, pp);
    else {
        if constexpr (std::is_same<T, int64_t>::value)
            LAUNCH(ctx, "pair_distance", (pair_distance_kernel<2, T>), dim3(grid), dim3(256), dl, dr, (uint64_t)n, pp);
        else
            return set_err(KPAL_E_INVALID, "euclidean is int64 only");
    }
    std::vector<Partial> res;
    CHK(finish_partials(ctx, 1, grid, res));
    *out = finish_value(metric, res[0], aux);
    return KPAL_OK;
}

KPAL_API int kpal_pair_distance_device(kpal_ctx *ctx, size_t n, const int64_t *dev_left, const int64_t *dev_right,
                                       int metric, int do_balance, int k, double *out, int64_t *aux_out)
{
    CTX_ENTER(ctx);
    if (!dev_left || !dev_right || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (metric < 0 || metric > 2) return set_err(KPAL_E_INVALID, "unknown metric %d", metric);
    if (((uintptr_t)dev_left & 15) || ((uintptr_t)dev_right & 15)) return set_err(KPAL_E_INVALID, "device vectors must be 16-byte aligned");
    const int64_t *l = dev_left, *r = dev_right;
    if (do_balance) {
        if (k < 1 || k > KPAL_MAX_K || n != (1ULL << (2 * k))) return set_err(KPAL_E_INVALID, "do_balance needs n == 4^k");
        if (k >= 6) {   // fused balance + distance: balanced values are formed in LDS tiles, never written
            const uint32_t *canon = nullptr;
            uint32_t ncanon = 0;
            unsigned grid = 1;
            CHK(canon_tiles(ctx, k, &canon, &ncanon, &grid));
            // persistent workgroups with prefetch, ONE per CU (120 registers), the pairs dealt round robin: every workgroup the same
            // number of pairs (k = 12: 0.099 -> 0.059 ms with the balanced deal)
            {
                const uint32_t slots = (uint32_t)ctx->num_cu, rounds = (ncanon + slots - 1) / slots;
                grid = (ncanon + rounds - 1) / rounds;
            }
            CHK(ensure(ctx, ctx->partials, (size_t)grid * sizeof(Partial)));
            Partial *pp = (Partial *)ctx->partials.p;
            if (metric == KPAL_PAIRWISE_PROD) LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<0>), dim3(grid), dim3(1024), l, r, k, canon, ncanon, pp);
            else if (metric == KPAL_PAIRWISE_SUM) LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<1>), dim3(grid), dim3(1024), l, r, k, canon, ncanon, pp);
            else LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<2>), dim3(grid), dim3(1024), l, r, k, canon, ncanon, pp);
            std::vector<Partial> res;
            CHK(finish_partials(ctx, 1, grid, res));
            *out = finish_value(metric, res[0], aux_out);
            return KPAL_OK;
        }
        CHK(ensure(ctx, ctx->scratch[2], n * 8));
        CHK(ensure(ctx, ctx->scratch[3], n * 8));
        CHK(launch_balance(ctx, k, l, (int64_t *)ctx->scratch[2].p));
        CHK(launch_balance(ctx, k, r, (int64_t *)ctx->scratch[3].p));
        l = (const int64_t *)ctx->scratch[2].p;
        r = (const int64_t *)ctx->scratch[3].p;
    }
    return pair_distance_launch<int64_t>(ctx, n, l, r, metric, out, aux_out);
}

KPAL_API int kpal_pair_distance(kpal_ctx *ctx, size_t n, const int64_t *host_left, const int64_t *host_right,
                                int metric, int do_balance, int k, double *out, int64_t *aux_out)
{
    CTX_ENTER(ctx);
    if (!host_left || !host_right || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_left, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->scratch[1].p, host_right, n * 8, hipMemcpyHostToDevice, ctx->stream));
    return kpal_pair_distance_device(ctx, n, (const int64_t *)ctx->scratch[0].p, (const int64_t *)ctx->scratch[1].p,
                                     metric, do_balance, k, out, aux_out);
}

KPAL_API int kpal_pair_distance_f64(kpal_ctx *ctx, size_t n, const double *host_left, const double *host_right,
                                    int pairwise, double *out, int64_t *aux_out)
{
    CTX_ENTER(ctx);
    if (!host_left || !host_right || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (pairwise != KPAL_PAIRWISE_PROD && pairwise != KPAL_PAIRWISE_SUM) return set_err(KPAL_E_INVALID, "pairwise must be prod or sum");
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_left, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->scratch[1].p, host_right, n * 8, hipMemcpyHostToDevice, ctx->stream));
    return pair_distance_launch<double>(ctx, n, (const double *)ctx->scratch[0].p, (const double *)ctx->scratch[1].p,
                                        pairwise, out, aux_out);
}

// Euclidean distances of all pairs from the fp64 Gram matrix (gram_kernels.hpp).  *exact = false (and
// out_lower untouched) when some |x|^2 >= 2^53: the caller then takes the wrapping-int64 path.
static int gram_euclidean(kpal_ctx *ctx, int P, uint64_t n, const int64_t *prof, double *out_lower, bool *exact, bool allreduce)
{
    const int nb = (P + 63) / 64;
    std::vector<int2> diag, off;
    for (int I = 0; I < nb; ++I)
        for (int J = 0; J <= I; ++J) (I == J ? diag : off).push_back(make_int2(I, J));
    const uint32_t nd = (uint32_t)diag.size(), no = (uint32_t)off.size();
    const uint64_t slabs = n / kGramBins;
    // diagonal blocks: two 68 KiB workgroups per CU; off-diagonal ones (P > 64): one
    const unsigned gx_d = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(slabs, (uint64_t)ctx->num_cu * 2 / nd));
    const unsigned gx_o = no ? (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(slabs, (uint64_t)ctx->num_cu / no)) : 0u;
    std::vector<int2> all(diag);
    all.insert(all.end(), off.begin(), off.end());
    CHK(ensure(ctx, ctx->scratch[3], all.size() * sizeof(int2)));
    HIPCHK(hipMemcpyAsync(ctx->scratch[3].p, all.data(), all.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    const size_t part_d = (size_t)nd * 4096 * gx_d, part_o = (size_t)no * 4096 * gx_o;
    CHK(ensure(ctx, ctx->partials, (part_d + part_o) * sizeof(Partial)));
    CHK(ensure(ctx, ctx->result, (size_t)(nd + no) * 4096 * sizeof(Partial)));
    Partial *pp = (Partial *)ctx->partials.p;
    Partial *res_d = (Partial *)ctx->result.p;
    const int2 *dt = (const int2 *)ctx->scratch[3].p;
    // a diagonal block writes only its 10 tiles with gj <= gi; the reduction below runs over all 16: the other six read zeros
    HIPCHK(hipMemsetAsync(pp, 0, part_d * sizeof(Partial), ctx->stream));
    LAUNCH(ctx, "gram_mfma", (gram_mfma_kernel<true>), dim3(gx_d, nd), dim3(256), prof, P, n, dt, pp);
    LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(nd * 4096), dim3(256), (const Partial *)pp, gx_d, res_d);
    if (no) {
        LAUNCH(ctx, "gram_mfma", (gram_mfma_kernel<false>), dim3(gx_o, no), dim3(256), prof, P, n, dt + nd, pp + part_d);
        LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(no * 4096), dim3(256), (const Partial *)(pp + part_d), gx_o,
               res_d + (size_t)nd * 4096);
    }
    if (allreduce) CHK(comm_allreduce_partials(ctx, res_d, (size_t)(nd + no) * 4096));   // (sums of exact integers: exact in any order below 2^53)
    std::vector<Partial> res((size_t)(nd + no) * 4096);
    HIPCHK(hipMemcpyAsync(res.data(), res_d, res.size() * sizeof(Partial), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // also: `all` was read by the asynchronous copy above
    auto gram = [&](int i, int j) -> double {    // i >= j
        const int I = i / 64, J = j / 64;
        size_t blk;
        if (I == J) blk = (size_t)I;             // diag[] is in order of I
        else {
            blk = nd;
            for (size_t t = 0; t < off.size(); ++t)
                if (off[t].x == I && off[t].y == J) blk = nd + t;
        }
        const int gi = (i % 64) / 16, gj = (j % 64) / 16;
        return res[(blk * 16 + (size_t)(gi * 4 + gj)) * 256 + (size_t)((i % 16) * 16 + (j % 16))].s;
    };
    const double limit = 9007199254740992.0;     // 2^53
    std::vector<double> norm(P);
    for (int i = 0; i < P; ++i) {
        norm[i] = gram(i, i);
        if (!(norm[i] < limit)) {
            *exact = false;
            return KPAL_OK;
        }
    }
    for (int i = 1; i < P; ++i)
        for (int j = 0; j < i; ++j) {
            // exact integers below 2^53 each: the int64 expression is the reference's sum of squared differences
            const int64_t d2 = (int64_t)norm[i] + (int64_t)norm[j] - 2 * (int64_t)gram(i, j);
            out_lower[(size_t)i * (i - 1) / 2 + j] = std::sqrt((double)d2);   // metrics.py:46: np.sqrt(np.dot(v, v))
        }
    *exact = true;
    return KPAL_OK;
}

KPAL_API int kpal_distance_matrix_device(kpal_ctx *ctx, int P, int k, const int64_t *dev_profiles, int metric,
                                         int do_balance, double *out_lower)
{
    CTX_ENTER(ctx);
    if (P < 1) return set_err(KPAL_E_INVALID, "P must be >= 1");
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (metric < 0 || metric > 2) return set_err(KPAL_E_INVALID, "unknown metric %d", metric);
    if (P == 1) return KPAL_OK;
    if (!dev_profiles || !out_lower) return set_err(KPAL_E_INVALID, "NULL pointer");
    const uint64_t n = 1ULL << (2 * k);
    const int64_t *prof = dev_profiles;
    if (do_balance) {
        // balance once per profile: identical to the reference balancing copies per pair (kdistlib.py:136-141)
        CHK(ensure(ctx, ctx->scratch[2], (size_t)P * n * 8));
        for (int p = 0; p < P; ++p)
            CHK(launch_balance(ctx, k, dev_profiles + (uint64_t)p * n, (int64_t *)ctx->scratch[2].p + (uint64_t)p * n));
        prof = (const int64_t *)ctx->scratch[2].p;
    }
    return distance_matrix_core(ctx, P, n, prof, metric, out_lower, false);
}

// The lower triangle over n bins per profile (profile p at prof + p * n).  allreduce: this rank holds a bin RANGE of every
// profile -- the per-pair sums / term counts / dot products of all ranks are added (one all-reduce) before they are finished.
int distance_matrix_core(kpal_ctx *ctx, int P, uint64_t n, const int64_t *prof, int metric, double *out_lower, bool allreduce, int tiled_agreed)
{
    // (whole profiles: k >= 6) the LDS-staged kernels take 64 bins at a time.  Bin-range shards: the ranks agreed on it
    // (kpal_comm_distance_matrix_device) -- the Gram path and the others all-reduce different things
    const bool tiled = n >= 4096 && n % 64 == 0 && tiled_agreed != 0;
    // euclidean with enough profiles and bins: fp64 Gram matrix on the matrix cores (gram_kernels.hpp), exact
    // while every |x|^2 < 2^53 (checked on the result); KPAL_MATRIX_MFMA=0 forces the int64 kernels
    static const bool allow_mfma = [] { const char *e = getenv("KPAL_MATRIX_MFMA"); return !e || atoi(e) != 0; }();
    if (metric == KPAL_EUCLIDEAN && allow_mfma && P > 8 && tiled) {
        bool exact = false;
        CHK(gram_euclidean(ctx, P, n, prof, out_lower, &exact, allreduce));
        if (exact) return KPAL_OK;
    }
    constexpr int TILE = 4;
    const int side = (P + TILE - 1) / TILE;
    std::vector<int2> tiles;
    for (int ti = 0; ti < side; ++ti)
        for (int tj = 0; tj <= ti; ++tj) tiles.push_back(make_int2(ti, tj));
    const uint32_t ntiles = (uint32_t)tiles.size();
    // LDS-staged 16 x 16 super-tiles when there are enough profiles and bins to share; KPAL_MATRIX_SUPER=0 forces
    // the register-tile kernel (A/B timing, cross-check)
    static const bool allow_super = [] { const char *e = getenv("KPAL_MATRIX_SUPER"); return !e || atoi(e) != 0; }();
    const bool super = allow_super && P > 8 && tiled;
    // multiset 'prod' of 17..64 profiles: every profile staged once per bin range (matrix_all_kernels.hpp; KPAL_MATRIX_ALL=0
    // forces the super-tile kernels)
    static const bool allow_all = [] { const char *e = getenv("KPAL_MATRIX_ALL"); return !e || atoi(e) != 0; }();
    static const bool allow_rdiff_all = [] { const char *e = getenv("KPAL_MATRIX_RDIFF"); return !e || atoi(e) != 0; }();
    unsigned gx;
    bool all_done = false;
    if (super && allow_all && allow_rdiff_all && metric <= 1 && P > 16 && P <= 64) {
        const bool wide = P > 32;
        gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n / 128, (uint64_t)ctx->num_cu * (wide ? 1 : 4)));
        CHK(ensure(ctx, ctx->scratch[3], 32));
        CHK(ensure(ctx, ctx->partials, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial)));
        Partial *pp = (Partial *)ctx->partials.p;
        uint32_t *big = (uint32_t *)ctx->scratch[3].p;
        HIPCHK(hipMemsetAsync(big, 0, 32, ctx->stream));
        HIPCHK(hipMemsetAsync(pp, 0, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial), ctx->stream));   // (.s / .m of a slot come from different threads)
        if (metric == 0) {
            if (wide) LAUNCH(ctx, "matrix_rdiff_all", (matrix_rdiff_all_kernel<16, kMatrixAllBins, kMatrixAllUnits>), dim3(gx), dim3(1024 / kMatrixAllUnits), prof, P, n, pp, big);
            else LAUNCH(ctx, "matrix_rdiff_all", (matrix_rdiff_all_kernel<8, 64, 1>), dim3(gx), dim3(256), prof, P, n, pp, big);
        } else {
            if (wide) LAUNCH(ctx, "matrix_rsum_all", (matrix_rsum_all_kernel<16, 64>), dim3(gx), dim3(1024), prof, P, n, pp, big);
            else LAUNCH(ctx, "matrix_rsum_all", (matrix_rsum_all_kernel<8, 64>), dim3(gx), dim3(256), prof, P, n, pp, big);
        }
        uint32_t saw_big = 0;
        HIPCHK(hipMemcpyAsync(&saw_big, big, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        all_done = saw_big == 0;
#if defined(KPAL_MALL_CLOCK)
        {
            unsigned long long clk[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpy(clk, big, sizeof(clk), hipMemcpyDeviceToHost));
            fprintf(stderr, "matrix_all: %llu shader cycles in %.1f us = %.0f MHz\n", clk[1], (double)clk[2] / 100.0, 100.0 * (double)clk[1] / (double)clk[2]);
        }
#endif
    }
    if (all_done) {
    } else if (super) {
        const int sside = (P + 15) / 16;
        std::vector<int2> supers;
        for (int si = 0; si < sside; ++si)
            for (int sj = 0; sj <= si; ++sj) supers.push_back(make_int2(si, sj));
        const uint32_t nsuper = (uint32_t)supers.size();
        gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n / kSuperBins, std::max<uint64_t>(1, (uint64_t)ctx->num_cu * 8 / nsuper)));
        gx = std::max(8u, gx / 8u * 8u);   // (matrix_rdiff_kernel deals bin-groups to the 8 XCDs; n / 64 >= 64 for k >= 6)
        CHK(ensure(ctx, ctx->scratch[3], (size_t)nsuper * sizeof(int2) + 16));
        HIPCHK(hipMemcpyAsync(ctx->scratch[3].p, supers.data(), (size_t)nsuper * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
        CHK(ensure(ctx, ctx->partials, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial)));
        Partial *pp = (Partial *)ctx->partials.p;
        const int2 *dt = (const int2 *)ctx->scratch[3].p;
        // multiset 'prod' as a difference of reciprocals (matrix_rdiff_kernel; KPAL_MATRIX_RDIFF=0 forces the pair-of-counts
        // kernel): valid while every count is below 2^16 -- the kernel says whether it saw a larger one
        static const bool allow_rdiff = [] { const char *e = getenv("KPAL_MATRIX_RDIFF"); return !e || atoi(e) != 0; }();
        bool rdiff_done = false;
        if (metric == 0 && allow_rdiff) {
            uint32_t *big = (uint32_t *)((int2 *)ctx->scratch[3].p + nsuper);
            HIPCHK(hipMemsetAsync(big, 0, sizeof(uint32_t), ctx->stream));
            HIPCHK(hipMemsetAsync(pp, 0, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial), ctx->stream));   // (.s / .m of a slot come from different threads)
            LAUNCH(ctx, "matrix_rdiff", matrix_rdiff_kernel, dim3(gx * nsuper), dim3(256), prof, P, n, dt, nsuper, pp, big);   // (gx: a multiple of 8)
            uint32_t saw_big = 0;
            HIPCHK(hipMemcpyAsync(&saw_big, big, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));   // (also: `supers` was read by the asynchronous copy above)
            rdiff_done = saw_big == 0;
        }
        if (metric == 1 && allow_rdiff) {   // multiset 'sum' with the reciprocals of the denominators from a table (matrix_rsum_kernel)
            uint32_t *big = (uint32_t *)((int2 *)ctx->scratch[3].p + nsuper);
            HIPCHK(hipMemsetAsync(big, 0, sizeof(uint32_t), ctx->stream));
            HIPCHK(hipMemsetAsync(pp, 0, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial), ctx->stream));
            LAUNCH(ctx, "matrix_rsum", matrix_rsum_kernel, dim3(gx * nsuper), dim3(256), prof, P, n, dt, nsuper, pp, big);
            uint32_t saw_big = 0;
            HIPCHK(hipMemcpyAsync(&saw_big, big, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            rdiff_done = saw_big == 0;
        }
        if (rdiff_done) {
        } else if (metric == 0) LAUNCH(ctx, "matrix_super", (matrix_super_kernel<0>), dim3(gx, nsuper), dim3(256), prof, P, n, dt, pp);
        else if (metric == 1) LAUNCH(ctx, "matrix_super", (matrix_super_kernel<1>), dim3(gx, nsuper), dim3(256), prof, P, n, dt, pp);
        else LAUNCH(ctx, "matrix_super", (matrix_super_kernel<2>), dim3(gx, nsuper), dim3(256), prof, P, n, dt, pp);
        HIPCHK(hipStreamSynchronize(ctx->stream));   // `supers` is read by the asynchronous copy above
    } else {
        gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, std::max<uint64_t>(1, (uint64_t)ctx->num_cu * 16 / ntiles)));
        CHK(ensure(ctx, ctx->scratch[3], (size_t)ntiles * sizeof(int2)));
        HIPCHK(hipMemcpyAsync(ctx->scratch[3].p, tiles.data(), (size_t)ntiles * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
        CHK(ensure(ctx, ctx->partials, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial)));
        Partial *pp = (Partial *)ctx->partials.p;
        const int2 *dt = (const int2 *)ctx->scratch[3].p;
        if (metric == 0) LAUNCH(ctx, "matrix_tile", (matrix_tile_kernel<0, TILE>), dim3(gx, ntiles), dim3(256), prof, P, n, dt, pp);
        else if (metric == 1) LAUNCH(ctx, "matrix_tile", (matrix_tile_kernel<1, TILE>), dim3(gx, ntiles), dim3(256), prof, P, n, dt, pp);
        else LAUNCH(ctx, "matrix_tile", (matrix_tile_kernel<2, TILE>), dim3(gx, ntiles), dim3(256), prof, P, n, dt, pp);
    }
    std::vector<Partial> res;
    CHK(finish_partials(ctx, ntiles * TILE * TILE, gx, res, allreduce));
    for (int i = 1; i < P; ++i)
        for (int j = 0; j < i; ++j) {
            const int ti = i / TILE, tj = j / TILE;
            const uint32_t t = (uint32_t)(ti * (ti + 1) / 2 + tj);
            const Partial &p = res[(size_t)t * TILE * TILE + (i % TILE) * TILE + (j % TILE)];
            out_lower[(size_t)i * (i - 1) / 2 + j] = finish_value(metric, p, nullptr);
        }
    return KPAL_OK;
}

KPAL_API int kpal_distance_matrix(kpal_ctx *ctx, int P, int k, const int64_t *const *host_profiles, int metric,
                                  int do_balance, double *out_lower)
{
    CTX_ENTER(ctx);
    if (P < 1) return set_err(KPAL_E_INVALID, "P must be >= 1");
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (P == 1) return KPAL_OK;
    if (!host_profiles || !out_lower) return set_err(KPAL_E_INVALID, "NULL pointer");
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->scratch[0], (size_t)P * n * 8));
    for (int p = 0; p < P; ++p) {
        if (!host_profiles[p]) return set_err(KPAL_E_INVALID, "profile %d is NULL", p);
        HIPCHK(hipMemcpyAsync((int64_t *)ctx->scratch[0].p + (uint64_t)p * n, host_profiles[p], n * 8,
                              hipMemcpyHostToDevice, ctx->stream));
    }
    return kpal_distance_matrix_device(ctx, P, k, (const int64_t *)ctx->scratch[0].p, metric, do_balance, out_lower);
}

// ----------------------------------------------------------------------------------------------
// ProfileDistance with options (kdistlib.py:126-161)
// ----------------------------------------------------------------------------------------------
static int check_options(const kpal_distance_options *opt)
{
    if (!opt) return set_err(KPAL_E_INVALID, "options are NULL");
    if (opt->metric < 0 || opt->metric > KPAL_COSINE) return set_err(KPAL_E_INVALID, "unknown metric %d", opt->metric);
    if (opt->do_smooth && (opt->summary < KPAL_SUMMARY_MIN || opt->summary > KPAL_SUMMARY_MEDIAN))
        return set_err(KPAL_E_INVALID, "unknown summary function %d", opt->summary);
    return KPAL_OK;
}

// Dynamic smoothing of (l, r) into (lo, ro); in == out allowed.
static int launch_smooth(kpal_ctx *ctx, int k, const int64_t *l, const int64_t *r, int64_t *lo, int64_t *ro,
                         int summary, double threshold)
{
    // level d = 0..k-1 has 4^d nodes: two int64 sums and one decision byte each
    // (level starts padded to even entries: the kernels read 16 bytes at a time)
    const uint64_t total = ((1ULL << (2 * k)) - 1) / 3 + (uint64_t)k;
    CHK(ensure(ctx, ctx->opt_levels, (size_t)total * 17 + 64));
    int64_t *sl = (int64_t *)ctx->opt_levels.p, *sr = sl + total;
    uint8_t *dec = (uint8_t *)(sr + total);
    SmoothLevels lv = {};
    uint64_t at = 0;
    for (int d = 0; d < k; ++d) {
        lv.sum_l[d] = sl + at;
        lv.sum_r[d] = sr + at;
        lv.decide[d] = dec + at;
        at += (1ULL << (2 * d)) + (d == 0 ? 1 : 0);
    }
    for (int d = k - 1; d >= 0; --d) {
        const uint64_t nparent = 1ULL << (2 * d);
        const int64_t *cl = d == k - 1 ? l : lv.sum_l[d + 1];
        const int64_t *cr = d == k - 1 ? r : lv.sum_r[d + 1];
        LAUNCH(ctx, "smooth_level", smooth_level_kernel, dim3(stream_grid(ctx, nparent)), dim3(256), cl, cr, nparent,
               (int64_t *)lv.sum_l[d], (int64_t *)lv.sum_r[d], (uint8_t *)lv.decide[d], summary, threshold);
    }
    LAUNCH(ctx, "smooth_apply", smooth_apply_kernel, dim3(stream_grid(ctx, 1ULL << (2 * (k - 1)))), dim3(256), l, r, k, lv, lo, ro);
    return KPAL_OK;
}

template <int METRIC>
static void launch_option_distance(kpal_ctx *ctx, unsigned grid, bool scaled, const int64_t *l, const int64_t *r, uint64_t n,
                                   double ls, double rs, Partial *pp)
{
    ProfScope ps_(ctx, "option_distance");
    if (scaled) hipLaunchKernelGGL((option_distance_kernel<METRIC, true>), dim3(grid), dim3(256), 0, ctx->stream, l, r, n, ls, rs, pp);
    else hipLaunchKernelGGL((option_distance_kernel<METRIC, false>), dim3(grid), dim3(256), 0, ctx->stream, l, r, n, ls, rs, pp);
}

// One pair, both vectors on the device and 16-byte aligned; `balanced`: the inputs are already
// balanced (matrix path), so opt->do_balance is not applied again.
static int profile_distance_pair(kpal_ctx *ctx, int k, const int64_t *dl, const int64_t *dr,
                                 const kpal_distance_options *opt, bool balanced, double *out)
{
    const uint64_t n = 1ULL << (2 * k);
    const bool do_balance = opt->do_balance && !balanced;
    if (!opt->do_positive && !opt->do_smooth && !opt->do_scale && opt->metric <= KPAL_EUCLIDEAN)
        return kpal_pair_distance_device(ctx, n, dl, dr, opt->metric, do_balance, k, out, nullptr);
    const int64_t *l = dl, *r = dr;
    if (do_balance || opt->do_positive || opt->do_smooth) {
        CHK(ensure(ctx, ctx->opt_l, n * 8));
        CHK(ensure(ctx, ctx->opt_r, n * 8));
    }
    int64_t *wl = (int64_t *)ctx->opt_l.p, *wr = (int64_t *)ctx->opt_r.p;
    if (do_balance) {
        CHK(launch_balance(ctx, k, l, wl));
        CHK(launch_balance(ctx, k, r, wr));
        l = wl;
        r = wr;
    }
    if (opt->do_positive) {
        LAUNCH(ctx, "positive", positive_kernel, dim3(stream_grid(ctx, n)), dim3(256), l, r, wl, wr, n);
        l = wl;
        r = wr;
    }
    if (opt->do_smooth) {
        CHK(launch_smooth(ctx, k, l, r, wl, wr, opt->summary, opt->threshold));
        l = wl;
        r = wr;
    }
    const unsigned grid = stream_grid(ctx, n);
    CHK(ensure(ctx, ctx->partials, (size_t)grid * 3 * sizeof(Partial)));
    Partial *pp = (Partial *)ctx->partials.p;
    std::vector<Partial> res;
    double ls = 1.0, rs = 1.0;
    if (opt->do_scale) {
        LAUNCH(ctx, "totals", totals_kernel, dim3(grid), dim3(256), l, r, n, pp);
        CHK(finish_partials(ctx, 2, grid, res));
        // metrics.get_scale, metrics.py:49-72: int64 totals, true division
        const int64_t tl = (int64_t)res[0].m, tr = (int64_t)res[1].m;
        if (tl < tr) ls = (double)tr / (double)tl;
        else rs = (double)tl / (double)tr;
        if (opt->down) {   // metrics.scale_down, metrics.py:75-86
            const double top = ls > rs ? ls : rs;
            ls /= top;
            rs /= top;
        }
    }
    const bool scaled = opt->do_scale != 0;
    switch (opt->metric) {
    case KPAL_PAIRWISE_PROD: launch_option_distance<0>(ctx, grid, scaled, l, r, n, ls, rs, pp); break;
    case KPAL_PAIRWISE_SUM: launch_option_distance<1>(ctx, grid, scaled, l, r, n, ls, rs, pp); break;
    case KPAL_EUCLIDEAN: launch_option_distance<2>(ctx, grid, scaled, l, r, n, ls, rs, pp); break;
    default: launch_option_distance<3>(ctx, grid, scaled, l, r, n, ls, rs, pp); break;
    }
    HIPCHK(hipGetLastError());
    CHK(finish_partials(ctx, opt->metric == KPAL_COSINE ? 3 : 1, grid, res));
    if (opt->metric <= KPAL_PAIRWISE_SUM) {
        *out = res[0].s / (double)(res[0].m + 1ULL);   // metrics.py:123
    } else if (opt->metric == KPAL_EUCLIDEAN) {
        *out = scaled ? std::sqrt(res[0].s) : std::sqrt((double)(int64_t)res[0].m);   // metrics.py:135,46
    } else {   // metrics.py:147: dot(l, r) / (|l| * |r|)
        if (scaled) *out = res[0].s / (std::sqrt(res[1].s) * std::sqrt(res[2].s));
        else *out = (double)(int64_t)res[0].m / (std::sqrt((double)(int64_t)res[1].m) * std::sqrt((double)(int64_t)res[2].m));
    }
    return KPAL_OK;
}

KPAL_API int kpal_profile_distance_device(kpal_ctx *ctx, int k, const int64_t *dev_left, const int64_t *dev_right,
                                          const kpal_distance_options *opt, double *out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!dev_left || !dev_right || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (((uintptr_t)dev_left & 15) || ((uintptr_t)dev_right & 15)) return set_err(KPAL_E_INVALID, "device vectors must be 16-byte aligned");
    CHK(check_options(opt));
    return profile_distance_pair(ctx, k, dev_left, dev_right, opt, false, out);
}

KPAL_API int kpal_profile_distance(kpal_ctx *ctx, int k, const int64_t *host_left, const int64_t *host_right,
                                   const kpal_distance_options *opt, double *out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!host_left || !host_right || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    CHK(check_options(opt));
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_left, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->scratch[1].p, host_right, n * 8, hipMemcpyHostToDevice, ctx->stream));
    return profile_distance_pair(ctx, k, (const int64_t *)ctx->scratch[0].p, (const int64_t *)ctx->scratch[1].p, opt, false, out);
}

KPAL_API int kpal_dynamic_smooth(kpal_ctx *ctx, int k, int64_t *host_left_inout, int64_t *host_right_inout,
                                 int summary, double threshold)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (!host_left_inout || !host_right_inout) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (summary < KPAL_SUMMARY_MIN || summary > KPAL_SUMMARY_MEDIAN) return set_err(KPAL_E_INVALID, "unknown summary function %d", summary);
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->opt_l, n * 8));
    CHK(ensure(ctx, ctx->opt_r, n * 8));
    int64_t *wl = (int64_t *)ctx->opt_l.p, *wr = (int64_t *)ctx->opt_r.p;
    HIPCHK(hipMemcpyAsync(wl, host_left_inout, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(wr, host_right_inout, n * 8, hipMemcpyHostToDevice, ctx->stream));
    CHK(launch_smooth(ctx, k, wl, wr, wl, wr, summary, threshold));
    HIPCHK(hipMemcpyAsync(host_left_inout, wl, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(host_right_inout, wr, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_profile_distance_matrix(kpal_ctx *ctx, int P, int k, const int64_t *const *host_profiles,
                                          const kpal_distance_options *opt, double *out_lower)
{
    CTX_ENTER(ctx);
    if (P < 1) return set_err(KPAL_E_INVALID, "P must be >= 1");
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    CHK(check_options(opt));
    if (P == 1) return KPAL_OK;
    if (!host_profiles || !out_lower) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (!opt->do_positive && !opt->do_smooth && !opt->do_scale && opt->metric <= KPAL_EUCLIDEAN)
        return kpal_distance_matrix(ctx, P, k, host_profiles, opt->metric, opt->do_balance, out_lower);
    const uint64_t n = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->opt_profiles, (size_t)P * n * 8));
    int64_t *prof = (int64_t *)ctx->opt_profiles.p;
    for (int p = 0; p < P; ++p) {
        if (!host_profiles[p]) return set_err(KPAL_E_INVALID, "profile %d is NULL", p);
        HIPCHK(hipMemcpyAsync(prof + (uint64_t)p * n, host_profiles[p], n * 8, hipMemcpyHostToDevice, ctx->stream));
        // balancing copies inside every pair (kdistlib.py:136-141) == balancing each profile once
        if (opt->do_balance) CHK(launch_balance(ctx, k, prof + (uint64_t)p * n, prof + (uint64_t)p * n));
    }
    for (int i = 1; i < P; ++i)
        for (int j = 0; j < i; ++j)
            CHK(profile_distance_pair(ctx, k, prof + (uint64_t)i * n, prof + (uint64_t)j * n, opt, true,
                                      &out_lower[(size_t)i * (i - 1) / 2 + j]));
    return KPAL_OK;
}

// ----------------------------------------------------------------------------------------------
// profile summaries, merge, shrink (stat_kernels.hpp)
// ----------------------------------------------------------------------------------------------
KPAL_API int kpal_stats_device(kpal_ctx *ctx, size_t n, const int64_t *dev_counts, kpal_profile_stats *out)
{
    CTX_ENTER(ctx);
    if (!dev_counts || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (n == 0) return set_err(KPAL_E_INVALID, "empty vector");
    const unsigned grid = stream_grid(ctx, n, kStatThreads);
    CHK(ensure(ctx, ctx->partials, (size_t)grid * sizeof(StatPartial) + 256 * 8));
    StatPartial *dp = (StatPartial *)ctx->partials.p;
    LAUNCH(ctx, "stats", stats_kernel, dim3(grid), dim3(kStatThreads), dev_counts, (uint64_t)n, dp);
    std::vector<StatPartial> hp(grid);
    HIPCHK(hipMemcpyAsync(hp.data(), dp, (size_t)grid * sizeof(StatPartial), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    StatPartial t = hp[0];
    for (unsigned b = 1; b < grid; ++b) {
        const uint64_t lo = t.sum_lo + hp[b].sum_lo;
        t.sum_hi += hp[b].sum_hi + (lo < t.sum_lo ? 1 : 0);
        t.sum_lo = lo;
        t.non_zero += hp[b].non_zero;
        t.mn = std::min(t.mn, hp[b].mn);
        t.mx = std::max(t.mx, hp[b].mx);
    }
    out->total = (int64_t)t.sum_lo;
    out->non_zero = (int64_t)t.non_zero;
    out->min = t.mn;
    out->max = t.mx;
    // the 128-bit sum as a double: magnitude first, so that a small negative sum does not cancel
    uint64_t mag_lo = t.sum_lo, mag_hi = (uint64_t)t.sum_hi;
    const bool negative = t.sum_hi < 0;
    if (negative) {
        mag_lo = ~mag_lo + 1ULL;
        mag_hi = ~mag_hi + (mag_lo == 0 ? 1ULL : 0ULL);
    }
    const double magnitude = std::ldexp((double)mag_hi, 64) + (double)mag_lo;
    const double exact_sum = negative ? -magnitude : magnitude;
    out->mean = exact_sum / (double)n;
    // std: sum((x - mean)^2) / n, kpal/klib.py:220-225 (ndarray.std)
    double *dv = (double *)ctx->partials.p;
    LAUNCH(ctx, "stats_var", stats_var_kernel, dim3(grid), dim3(kStatThreads), dev_counts, (uint64_t)n, out->mean, dv);
    std::vector<double> hv(grid);
    HIPCHK(hipMemcpyAsync(hv.data(), dv, (size_t)grid * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double ss = 0.0;
    for (unsigned b = 0; b < grid; ++b) ss += hv[b];
    out->std = std::sqrt(ss / (double)n);
    // median: radix select of rank (n-1)/2 over the bytes in which min and max differ
    const uint64_t r0 = (n - 1) / 2, r1 = n / 2;
    if (t.mn == t.mx) {
        out->median = (double)t.mn;
        return KPAL_OK;
    }
    const uint64_t kmin = select_key(t.mn), kmax = select_key(t.mx);
    int top = 7;
    while (((kmin >> (8 * top)) & 255u) == ((kmax >> (8 * top)) & 255u)) --top;   // kmin != kmax: terminates at >= 0
    uint64_t mask = top == 7 ? 0ULL : ~0ULL << (8 * (top + 1));
    uint64_t prefix = kmin & mask;
    uint64_t below = 0, equal = 0;      // elements with key < / == the digits chosen so far
    unsigned long long *dh = (unsigned long long *)ctx->partials.p;
    unsigned long long hh[256];
    for (int byte = top; byte >= 0; --byte) {
        HIPCHK(hipMemsetAsync(dh, 0, 256 * 8, ctx->stream));
        LAUNCH(ctx, "select_hist", select_hist_kernel, dim3(grid), dim3(kStatThreads), dev_counts, (uint64_t)n, mask, prefix,
               8 * byte, dh);
        HIPCHK(hipMemcpyAsync(hh, dh, sizeof(hh), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        int d = 0;
        uint64_t acc = below;
        for (; d < 256; ++d) {
            if (r0 < acc + hh[d]) break;
            acc += hh[d];
        }
        if (d == 256) return set_err(KPAL_E_HIP, "median: rank %llu not found", (unsigned long long)r0);
        below = acc;
        equal = hh[d];
        prefix |= (uint64_t)d << (8 * byte);
        mask |= 255ULL << (8 * byte);
    }
    const int64_t v0 = (int64_t)(prefix ^ 0x8000000000000000ULL);
    int64_t v1 = v0;
    if (r1 >= below + equal) {   // the upper middle element is the next larger value
        LAUNCH(ctx, "select_next", select_next_kernel, dim3(grid), dim3(kStatThreads), dev_counts, (uint64_t)n, prefix, dh);
        std::vector<unsigned long long> hm(grid);
        HIPCHK(hipMemcpyAsync(hm.data(), dh, (size_t)grid * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        unsigned long long m = ~0ULL;
        for (unsigned b = 0; b < grid; ++b) m = std::min(m, hm[b]);
        v1 = (int64_t)(m ^ 0x8000000000000000ULL);
    }
    out->median = ((double)v0 + (double)v1) / 2.0;   // np.median: mean of the two middle elements
    return KPAL_OK;
}

KPAL_API int kpal_stats(kpal_ctx *ctx, size_t n, const int64_t *host_counts, kpal_profile_stats *out)
{
    CTX_ENTER(ctx);
    if (!host_counts || !out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (n == 0) return set_err(KPAL_E_INVALID, "empty vector");
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_counts, n * 8, hipMemcpyHostToDevice, ctx->stream));
    return kpal_stats_device(ctx, n, (const int64_t *)ctx->scratch[0].p, out);
}

KPAL_API int kpal_merge_device(kpal_ctx *ctx, size_t n, const int64_t *dev_left, const int64_t *dev_right, int merger,
                               int64_t *dev_out)
{
    CTX_ENTER(ctx);
    if (!dev_left || !dev_right || !dev_out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (merger < KPAL_MERGE_SUM || merger > KPAL_MERGE_NINT) return set_err(KPAL_E_INVALID, "unknown merger %d", merger);
    if (n == 0) return KPAL_OK;
    const unsigned grid = stream_grid(ctx, n);
    switch (merger) {
    case KPAL_MERGE_SUM: LAUNCH(ctx, "merge", (merge_kernel<0>), dim3(grid), dim3(256), dev_left, dev_right, (uint64_t)n, dev_out); break;
    case KPAL_MERGE_XOR: LAUNCH(ctx, "merge", (merge_kernel<1>), dim3(grid), dim3(256), dev_left, dev_right, (uint64_t)n, dev_out); break;
    case KPAL_MERGE_INT: LAUNCH(ctx, "merge", (merge_kernel<2>), dim3(grid), dim3(256), dev_left, dev_right, (uint64_t)n, dev_out); break;
    default: LAUNCH(ctx, "merge", (merge_kernel<3>), dim3(grid), dim3(256), dev_left, dev_right, (uint64_t)n, dev_out); break;
    }
    return KPAL_OK;
}

KPAL_API int kpal_merge(kpal_ctx *ctx, size_t n, const int64_t *host_left, const int64_t *host_right, int merger,
                        int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (!host_left || !host_right || !host_out) return set_err(KPAL_E_INVALID, "NULL pointer");
    if (merger < KPAL_MERGE_SUM || merger > KPAL_MERGE_NINT) return set_err(KPAL_E_INVALID, "unknown merger %d", merger);
    if (n == 0) return KPAL_OK;
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], n * 8));
    int64_t *dl = (int64_t *)ctx->scratch[0].p, *dr = (int64_t *)ctx->scratch[1].p;
    HIPCHK(hipMemcpyAsync(dl, host_left, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dr, host_right, n * 8, hipMemcpyHostToDevice, ctx->stream));
    CHK(kpal_merge_device(ctx, n, dl, dr, merger, dl));
    HIPCHK(hipMemcpyAsync(host_out, dl, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_shrink_device(kpal_ctx *ctx, int k, int factor, const int64_t *dev_counts, int64_t *dev_out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (factor < 1 || factor >= k) return set_err(KPAL_E_INVALID, "Reduction factor should be smaller than k-mer size.");
    if (!dev_counts || !dev_out) return set_err(KPAL_E_INVALID, "NULL pointer");
    const uint64_t n = 1ULL << (2 * k), m = 1ULL << (2 * factor), n_out = n / m;
    if (m <= 64) {
        LAUNCH(ctx, "shrink", shrink_small_kernel, dim3(stream_grid(ctx, n / 2)), dim3(256), dev_counts, n / 2, (int)(m / 2), dev_out);
    } else {
        LAUNCH(ctx, "shrink", shrink_large_kernel, dim3(stream_grid(ctx, n_out * 64)), dim3(256), dev_counts, n_out, m, dev_out);
    }
    return KPAL_OK;
}

KPAL_API int kpal_shrink(kpal_ctx *ctx, int k, int factor, const int64_t *host_counts, int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range", k);
    if (factor < 1 || factor >= k) return set_err(KPAL_E_INVALID, "Reduction factor should be smaller than k-mer size.");
    if (!host_counts || !host_out) return set_err(KPAL_E_INVALID, "NULL pointer");
    const uint64_t n = 1ULL << (2 * k), n_out = n >> (2 * factor);
    CHK(ensure(ctx, ctx->scratch[0], n * 8));
    CHK(ensure(ctx, ctx->scratch[1], n_out * 8));
    HIPCHK(hipMemcpyAsync(ctx->scratch[0].p, host_counts, n * 8, hipMemcpyHostToDevice, ctx->stream));
    CHK(kpal_shrink_device(ctx, k, factor, (const int64_t *)ctx->scratch[0].p, (int64_t *)ctx->scratch[1].p));
    HIPCHK(hipMemcpyAsync(host_out, ctx->scratch[1].p, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}


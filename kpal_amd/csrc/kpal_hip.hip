// kpal_hip.hip -- C-ABI (include/kpal_hip.h) over the gfx950 kernels.  Host side: context,
// workspace management, launch planning, H2D staging and per-kernel HIP-event timing.
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

#include "count_kernels.hpp"
#include "partition_kernels.hpp"
#include "chunk_kernels.hpp"
#include "quad_kernels.hpp"
#include "fasta_kernels.hpp"
#include "vec_kernels.hpp"
#include "gram_kernels.hpp"
#include "option_kernels.hpp"
#include "stat_kernels.hpp"

#define KPAL_API extern "C" __attribute__((visibility("default")))

using namespace kpal;

// ----------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

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

KPAL_API const char *kpal_last_error(void) { return g_err; }
KPAL_API const char *kpal_version(void) { return "kpal_amd 0.1 (gfx950)"; }

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
    int level2_mode = 2;                     // level 2 of the two-level path (KPAL_LEVEL2): 0 count + exact offsets, 1 chunked per-tile runs, 2 chunked aligned lines (default)
    ChunkPool chunk_pool_sent = {};          // what the device copy of the pool descriptor holds
    ChunkPool *chunk_pool_dev = nullptr;
    uint32_t chunk_meta_y = 0;               // coarse-bucket count the meta layout was cleared for
    uint32_t *chunk_error_word = nullptr;
    DevBuf residuals, cnt1, offs1, start1;  // two-level path (k = 13..16)
    DevBuf fa_raw, fa_flat, fa_meta;          // FASTA ingest
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
    // profiling
    bool prof = false;
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<uint64_t> prof_launches;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    uint64_t prof_dropped = 0;               // launches whose timing events could not be recorded
};

static int ensure(kpal_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (b.cap >= bytes && b.p) return KPAL_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return set_err(KPAL_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    b.cap = bytes;
    return KPAL_OK;
}

static int prof_name_id(kpal_ctx *ctx, const char *name)
{
    for (size_t i = 0; i < ctx->prof_names.size(); ++i)
        if (ctx->prof_names[i] == name) return (int)i;
    ctx->prof_names.push_back(name);
    ctx->prof_ms.push_back(0.0);
    ctx->prof_launches.push_back(0);
    return (int)ctx->prof_names.size() - 1;
}

static hipEvent_t prof_event(kpal_ctx *ctx)
{
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

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

static int prof_collect(kpal_ctx *ctx)
{
    if (ctx->prof_pending.empty()) return KPAL_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof_pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        ctx->prof_ms[r.name] += ms;
        ctx->prof_launches[r.name] += 1;
        ctx->ev_pool.push_back(r.a);
        ctx->ev_pool.push_back(r.b);
    }
    ctx->prof_pending.clear();
    return KPAL_OK;
}

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

KPAL_API int kpal_device_count(int *n)
{
    if (!n) return set_err(KPAL_E_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return set_err(KPAL_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return KPAL_OK;
}

// Everything of kpal_ctx_create that can fail after the context object exists: on an error the
// caller destroys the half-built context (streams, events), nothing leaks.
static int ctx_init(kpal_ctx *ctx, int device)
{
    ctx->device = device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_copied[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
    }
    if (const char *e = getenv("KPAL_BATCH_BYTES")) {
        unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= (1ULL << 20)) {
            ctx->batch_bytes = (size_t)v;
            ctx->batch_bytes_set = true;
        }
    }
    if (const char *e = getenv("KPAL_LEVEL2")) ctx->level2_mode = atoi(e);
    if (const char *e = getenv("KPAL_SPLIT_ABOVE")) {   // tests: exercise the batch-halving path on small inputs
        unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 1024) ctx->split_above = v;
    }
    if (const char *e = getenv("KPAL_QUAD_POOL_MAX")) {   // tests: exercise the piece-halving path of the quad pipelines
        unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= ((size_t)1 << 20) && v < ((size_t)30 << 30)) ctx->quad_pool_max = (size_t)v;
    }
    return KPAL_OK;
}

KPAL_API void kpal_ctx_destroy(kpal_ctx *ctx);

KPAL_API int kpal_ctx_create(int device, kpal_ctx **out)
{
    if (!out) return set_err(KPAL_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return set_err(KPAL_E_INVALID, "device %d not in 0..%d", device, n - 1);
    HIPCHK(hipSetDevice(device));
    kpal_ctx *ctx = new (std::nothrow) kpal_ctx();
    if (!ctx) return set_err(KPAL_E_NOMEM, "out of host memory");
    const int rc = ctx_init(ctx, device);
    if (rc != KPAL_OK) {
        ctx->device = device;
        kpal_ctx_destroy(ctx);   // keeps g_err: it only releases what was created
        return rc;
    }
    *out = ctx;
    return KPAL_OK;
}

KPAL_API void kpal_ctx_destroy(kpal_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&ctx->table, &ctx->keys, &ctx->cntmat, &ctx->offs, &ctx->bucket_start, &ctx->slice_start, &ctx->chunk_meta, &ctx->chunk_table, &ctx->chunk_ovf, &ctx->chunk_sorted, &ctx->quad_meta, &ctx->quad_meta2, &ctx->residuals, &ctx->cnt1, &ctx->offs1, &ctx->start1, &ctx->fa_raw, &ctx->fa_flat, &ctx->fa_meta, &ctx->dstage[0],
                      &ctx->dstage[1], &ctx->scratch[0], &ctx->scratch[1], &ctx->scratch[2], &ctx->scratch[3],
                      &ctx->partials, &ctx->result, &ctx->opt_l, &ctx->opt_r, &ctx->opt_levels, &ctx->opt_profiles};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < 2; ++i) {
        if (ctx->pinned[i]) (void)hipHostFree(ctx->pinned[i]);
        if (ctx->ev_copied[i]) (void)hipEventDestroy(ctx->ev_copied[i]);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
    }
    for (auto &r : ctx->prof_pending) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

#define CTX_ENTER(ctx)                                            \
    if (!(ctx)) return set_err(KPAL_E_INVALID, "ctx is NULL");    \
    HIPCHK(hipSetDevice((ctx)->device))

KPAL_API int kpal_sync(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_dev_alloc(kpal_ctx *ctx, size_t nbytes, void **dev_out)
{
    CTX_ENTER(ctx);
    if (!dev_out) return set_err(KPAL_E_INVALID, "dev_out is NULL");
    *dev_out = nullptr;
    hipError_t e = hipMalloc(dev_out, nbytes ? nbytes : 16);
    if (e != hipSuccess) return set_err(KPAL_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", nbytes, hipGetErrorString(e));
    return KPAL_OK;
}

KPAL_API int kpal_dev_free(kpal_ctx *ctx, void *dev)
{
    CTX_ENTER(ctx);
    if (!dev) return KPAL_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(dev));
    return KPAL_OK;
}

KPAL_API int kpal_memcpy_h2d(kpal_ctx *ctx, void *dev_dst, const void *host_src, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (nbytes == 0) return KPAL_OK;
    HIPCHK(hipMemcpyAsync(dev_dst, host_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_memcpy_d2h(kpal_ctx *ctx, void *host_dst, const void *dev_src, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (nbytes == 0) return KPAL_OK;
    HIPCHK(hipMemcpyAsync(host_dst, dev_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

// ----------------------------------------------------------------------------------------------
// counting
// ----------------------------------------------------------------------------------------------
KPAL_API int kpal_count_begin(kpal_ctx *ctx, int k)
{
    CTX_ENTER(ctx);
    if (k < 1 || k > KPAL_MAX_K) return set_err(KPAL_E_INVALID, "k=%d out of range 1..%d", k, KPAL_MAX_K);
    ctx->k = k;
    ctx->bins = 1ULL << (2 * k);
    CHK(ensure(ctx, ctx->table, ctx->bins * sizeof(int64_t)));
    HIPCHK(hipMemsetAsync(ctx->table.p, 0, ctx->bins * sizeof(int64_t), ctx->stream));
    if (ctx->chunk_error_word) HIPCHK(hipMemsetAsync(ctx->chunk_error_word, 0, sizeof(uint32_t), ctx->stream));
    if (ctx->quad_error_word) HIPCHK(hipMemsetAsync(ctx->quad_error_word, 0, sizeof(uint32_t), ctx->stream));
    ctx->chunk_error_armed = false;
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
    if (memcmp(&p, &ctx->chunk_pool_sent, sizeof(ChunkPool)) != 0 || cl.dpool != ctx->chunk_pool_dev) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(cl.dpool, &p, sizeof(ChunkPool), hipMemcpyHostToDevice));
        ctx->chunk_pool_sent = p;
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

constexpr double kQuadBacklogMax = 1500.0;   // quad_choose_steps: expected steady-state backlog a tile size may bring (list: 2048)
constexpr int kQuadsUseChunked = 2;   // launch_partition_quads (AUTO): the sample shows a feed for the chunked pipeline
constexpr int kSplitBatch = 1;        // launch_partition2 / launch_partition*_quads: the caller halves the piece

// Expected number of items in the spill list of a workgroup in the steady state.  A row is a queue: Poisson(mu) items
// arrive per round, `slots` leave with the record, the rest is carried to the next round.  The single-round overflow
// E[max(X - slots, 0)] underestimates the backlog of a well-filled row (carried items arrive again: at 83 % fill of a
// 16-slot row the backlog is twice the overflow, at 95 % six times; a row whose load exceeds its slots grows without
// bound until the list is full and the slow direct path takes over -- measured 2x slower on AT-rich input with tiles
// chosen by the single-round figure).  backlog = overflow x r(fill, slots), r tabulated from a simulation of the queue
// (tools/diag/spill_queue.py).
static double quad_expected_backlog(const std::vector<double> &mu, int slots)
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
static int quad_choose_steps(kpal_ctx *ctx, const Span &s, uint32_t *load, int buckets, int slots, int waves, const int *candidates,
                             size_t n_candidates, int *steps_out, std::vector<double> *fine_per_step = nullptr)
{
    const int extra = fine_per_step ? 512 : 0;   // (two-level path: the sample also returns the loads of the 512 fine rows)
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    const uint32_t sample_steps = 4;                                       // per wave: 32 KiB per workgroup
    const uint64_t want = std::max<uint64_t>(1, total_steps / (64ull * 8 * sample_steps));   // ~1/64 of the input
    const uint32_t groups = (uint32_t)std::min<uint64_t>(want, 1024);
    const uint64_t stride = std::max<uint64_t>(8 * sample_steps, total_steps / groups);
    HIPCHK(hipMemsetAsync(load, 0, (size_t)(buckets + extra) * sizeof(uint32_t), ctx->stream));
    DISPATCH_K_8_16(ctx->k, LAUNCH(ctx, "quad_sample", (quad_sample_kernel<K>), dim3(groups), dim3(512), s, stride, sample_steps, load));
    std::vector<uint32_t> h((size_t)(buckets + extra));
    HIPCHK(hipMemcpyAsync(h.data(), load, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const double sampled_steps = (double)std::min<uint64_t>((uint64_t)groups * 8 * sample_steps, total_steps);
    // the 32 fullest rows are left out: a handful of very hot rows (poly-A, an adapter shared by every read) cannot be
    // helped by smaller tiles -- their items are counted in the workgroup's hot-item table instead
    static const bool verbose = [] { const char *e = getenv("KPAL_QUAD_VERBOSE"); return e && atoi(e) != 0; }();
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
static int launch_partition_quads(kpal_ctx *ctx, const Span &s)
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
    static const int steps_env = [] { const char *e = getenv("KPAL_QUAD_STEPS"); return e ? atoi(e) : 0; }();
    static const int candidates[] = {8, 7, 6, 4, 3, 2, 1};
    constexpr int waves = 16;
    int steps = 0;
    for (int c : candidates)
        if (c == steps_env) steps = c;
    if (!steps) {
        const int rc = quad_choose_steps(ctx, s, load, buckets, slots, waves, candidates, sizeof(candidates) / sizeof(candidates[0]), &steps);
        if (rc != KPAL_OK) return rc;
    }
    const uint64_t tile_steps = (uint64_t)waves * steps;
    const uint64_t tiles = (total_steps + tile_steps - 1) / tile_steps;
    const uint32_t G = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cu, tiles);
    const uint64_t tpb = (tiles + G - 1) / G;          // tiles (= flush rounds) per workgroup
    if (tpb > 0xFFFFFFull) return set_err(KPAL_E_INVALID, "quad partition: batch too large");
    const size_t pool_bytes = (size_t)kQuadRowWords * 4 * G * tpb;   // every round writes all rows: 128 KiB per workgroup
    if (pool_bytes > ctx->quad_pool_max && s.nchunks > 64) return kSplitBatch;   // (heavily skewed 16 GiB piece: small tiles)
    CHK(ensure(ctx, ctx->keys, pool_bytes));
    uint32_t *pool = (uint32_t *)ctx->keys.p;
    unsigned long long *table = (unsigned long long *)ctx->table.p;
#define KPAL_QUAD_LAUNCH(S)                                                                                                  \
    LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, S, S>), dim3(G), dim3(1024), s, tpb, pool, (uint32_t)tpb, nrounds, \
           error, table)
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
               (const uint32_t *)nrounds, G, (uint32_t)tpb, table, (uint32_t *)nullptr);
    });
#undef KPAL_QUAD_LAUNCH
    static const bool verbose = [] { const char *e = getenv("KPAL_QUAD_VERBOSE"); return e && atoi(e) != 0; }();
    if (verbose) {   // diagnostics: tile size chosen, tiles abandoned to the direct path
        uint32_t st[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(st, error, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[kpal quad] k=%d steps/wave/tile=%d tiles=%llu workgroups=%u hot-table entries used so far=%u\n", ctx->k, steps,
                (unsigned long long)tiles, G, st[1]);
    }
    return KPAL_OK;
}

// Two-level partition of quads, k = 13..16 (quad_kernels.hpp, end): level-1 records by coarse bucket, level-2 records
// by (coarse, fine) bucket, histogram per (coarse, fine) bucket.
static int launch_partition2_quads(kpal_ctx *ctx, const Span &s)
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
    static const int steps_env = [] { const char *e = getenv("KPAL_QUAD_STEPS"); return e ? atoi(e) : 0; }();
    static const int candidates[] = {8, 7, 6, 3};
    int steps1 = 0;
    for (int c : candidates)
        if (c == steps_env) steps1 = c;
    std::vector<double> fine;                    // items per fine row of level 2 per wave-step of INPUT (sorted)
    {
        int chosen = 0;
        const int rc = quad_choose_steps(ctx, s, error + 4, (int)(NB1 * REP), (int)S1, 16, candidates, 4, &chosen, &fine);
        if (rc != KPAL_OK) return rc;
        if (!steps1) steps1 = chosen;
    }
    const uint64_t tile_steps = 16ull * steps1;
    const uint64_t tiles1 = (total_steps + tile_steps - 1) / tile_steps;
    const uint32_t G1 = (uint32_t)std::min<uint64_t>((uint64_t)std::min(ctx->num_cu, 256), tiles1);
    const uint64_t tpb1 = (tiles1 + G1 - 1) / G1;
    if (tpb1 > 0xFFFFull) return set_err(KPAL_E_INVALID, "quad partition: batch too large");
    // capacity (stride) of a level-1 workgroup's run of records per row: rounded up so that a unit of level 2 is a whole
    // number of KiB -- a wave-step of quad2_scatter_kernel then never straddles two units
    const uint64_t per_kib = 1024 / (S1 * 4);                                   // records per KiB: 2 (8 at k = 16)
    const uint64_t cap1 = (tpb1 + per_kib - 1) / per_kib * per_kib;
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
    static const int steps2_env = [] { const char *e = getenv("KPAL_QUAD_STEPS2"); return e ? atoi(e) : 0; }();
    int steps2 = 2;
    {
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
        for (int c : candidates2)
            if (c == steps2_env) steps2 = c;
    }
    const uint64_t tile2_bytes = (uint64_t)kWaves2 * steps2 * 1024;
    const uint64_t tiles2 = ((uint64_t)upw * unit_cap + tile2_bytes - 1) / tile2_bytes;
    CHK(ensure(ctx, ctx->keys, (size_t)kQuadRowWords * 4 * NB1 * G2 * tiles2));
    CHK(ensure(ctx, ctx->quad_meta2, (size_t)NB1 * G2 * sizeof(uint32_t)));
    // the staged forms of the histogram stage (four 16-bit counts per table entry: 8.6 GB at k = 15) reuse the level-1 pool's buffer:
    // level 2 has read it completely before the histogram kernel starts (same stream)
    CHK(ensure(ctx, ctx->residuals, std::max<size_t>((size_t)kQuadRowWords * 4 * G1 * cap1, (size_t)ctx->bins * 8)));
    pool1 = (uint32_t *)ctx->residuals.p;
    uint32_t *stage = pool1;
    uint32_t *pool2 = (uint32_t *)ctx->keys.p;
    uint32_t *nrounds2 = (uint32_t *)ctx->quad_meta2.p;
    unsigned long long *table = (unsigned long long *)ctx->table.p;
#define KPAL_QUAD2_LAUNCH(S2)                                                                                                          \
    LAUNCH(ctx, "quad2_scatter", (quad2_scatter_kernel<K, kWaves2, S2>), dim3(G2, NB1), dim3(kWaves2 * 64), (const uint32_t *)pool1, \
           (const uint32_t *)nrounds1, G1, (uint32_t)cap1, upw, (uint32_t)tiles2, pool2, (uint32_t)tiles2, nrounds2, error, table)
    DISPATCH_K_13_16(ctx->k, {
        if (steps1 == 8)
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, 8, 8>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table);
        else if (steps1 == 7)
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, 7, 7>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table);
        else if (steps1 == 6)
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, 6, 6>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table);
        else
            LAUNCH(ctx, "quad_scatter", (quad_scatter_kernel<K, 16, 3, 3>), dim3(G1), dim3(1024), s, tpb1, pool1, (uint32_t)cap1, nrounds1, error, table);
        switch (steps2) {
        case 8: KPAL_QUAD2_LAUNCH(8); break;
        case 7: KPAL_QUAD2_LAUNCH(7); break;
        case 6: KPAL_QUAD2_LAUNCH(6); break;
        case 4: KPAL_QUAD2_LAUNCH(4); break;
        case 3: KPAL_QUAD2_LAUNCH(3); break;
        default: KPAL_QUAD2_LAUNCH(2); break;
        }

        LAUNCH(ctx, "quad_hist", (quad_hist_kernel<K>), dim3(512, NB1), dim3(1024), (const uint32_t *)pool2, (const uint32_t *)nrounds2,
               G2, (uint32_t)tiles2, table, stage);
        LAUNCH(ctx, "quad2_combine", (quad2_combine_kernel<K>), dim3((unsigned)(ctx->bins / 2048)), dim3(256), (const uint16_t *)stage, table);
    });
#undef KPAL_QUAD2_LAUNCH
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
    // the two-level quad pipeline stages and combines 16 bytes per TABLE ENTRY whatever the feed holds (k = 15: 17 GB written,
    // 34 GB combined): it wins once the feed is about half as large as the table (k = 15: 44 vs 59 ms on 15 GB, but 11.8 vs
    // 8.7 ms on 1 GiB); below that the round-1 two-level pipeline stays
    else if (ctx->strategy == KPAL_STRATEGY_AUTO && strat == KPAL_STRATEGY_PARTITION2_QUADS &&
             n < std::max<size_t>((size_t)64 << 20, (size_t)(ctx->bins * 4)))
        strat = KPAL_STRATEGY_PARTITION2;
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
        const size_t h = std::min(km1, halo + off);
        const Span s = make_span(addr + off, len, h);
        if (strat == KPAL_STRATEGY_GLOBAL_ATOMIC) CHK(launch_global_atomic(ctx, s));
        else if (strat == KPAL_STRATEGY_LDS_DIRECT) CHK(launch_lds_direct(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION) CHK(launch_partition(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION_CHUNKED) CHK(launch_partition_chunked(ctx, s));
        else if (strat == KPAL_STRATEGY_PARTITION2_QUADS) {
            const int rc = launch_partition2_quads(ctx, s);
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
    return count_device_range(ctx, (const uint8_t *)dev_buf, nbytes, 0);
}

// Host copy into a pinned staging buffer on several cores: one core's memcpy (~10 GB/s) is what limits
// a pageable-memory feed otherwise, the PCIe link takes ~5 times that.  KPAL_COPY_THREADS (default 4, 1 =
// plain memcpy); pieces below 4 MiB are not worth the thread start.
static void staged_memcpy(void *dst, const void *src, size_t n)
{
    static const int configured = [] {
        const char *e = getenv("KPAL_COPY_THREADS");
        int t = e ? atoi(e) : 4;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)t > hw) t = (int)hw;
        return t < 1 ? 1 : (t > 32 ? 32 : t);
    }();
    const size_t min_part = (size_t)4 << 20;
    int parts = (int)std::min<size_t>((size_t)configured, n / min_part);
    if (parts <= 1) {
        memcpy(dst, src, n);
        return;
    }
    const size_t part = ((n / parts) + 4095) & ~(size_t)4095;
    std::vector<std::thread> workers;
    workers.reserve(parts - 1);
    for (int i = 1; i < parts; ++i) {
        const size_t off = (size_t)i * part;
        if (off >= n) break;
        const size_t len = std::min(part, n - off);
        try {
            workers.emplace_back([=] { memcpy((uint8_t *)dst + off, (const uint8_t *)src + off, len); });
        } catch (...) {   // no thread to be had: copy this part here (no exception may cross the C-ABI)
            memcpy((uint8_t *)dst + off, (const uint8_t *)src + off, len);
        }
    }
    memcpy(dst, src, std::min(part, n));
    for (auto &w : workers) w.join();
}

static int ensure_pinned(kpal_ctx *ctx)
{
    for (int i = 0; i < 2; ++i) {
        if (!ctx->pinned[i]) {
            hipError_t e = hipHostMalloc(&ctx->pinned[i], kpal_ctx::kStage + kpal_ctx::kStagePad, hipHostMallocDefault);
            if (e != hipSuccess) return set_err(KPAL_E_NOMEM, "hipHostMalloc staging failed: %s", hipGetErrorString(e));
        }
    }
    return KPAL_OK;
}

// Pageable host memory -> device through the two pinned staging buffers: the memcpy into one
// overlaps the DMA out of the other.  ctx->stream waits for the last piece.
static int h2d_staged(kpal_ctx *ctx, uint8_t *dev_dst, const uint8_t *host_src, size_t n)
{
    CHK(ensure_pinned(ctx));
    const size_t stage = kpal_ctx::kStage;
    int slot = 0, last = -1;
    for (size_t off = 0; off < n; off += stage, slot ^= 1) {
        const size_t len = std::min(stage, n - off);
        if (ctx->stage_used[slot]) HIPCHK(hipEventSynchronize(ctx->ev_copied[slot]));   // its previous DMA is done
        staged_memcpy(ctx->pinned[slot], host_src + off, len);
        HIPCHK(hipMemcpyAsync(dev_dst + off, ctx->pinned[slot], len, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
        ctx->stage_used[slot] = true;
        last = slot;
    }
    if (last >= 0) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[last], 0));
    return KPAL_OK;
}

KPAL_API int kpal_count_feed(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!host_buf) return set_err(KPAL_E_INVALID, "host_buf is NULL");
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
        uint8_t *hp = (uint8_t *)ctx->pinned[slot] + (pad - h);
        staged_memcpy(hp, host_buf + off - h, len + h);
        uint8_t *dp = (uint8_t *)ctx->dstage[slot].p + (pad - h);
        HIPCHK(hipMemcpyAsync(dp, hp, len + h, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        CHK(count_device_range(ctx, dp + h, len, h));
        HIPCHK(hipEventRecord(ctx->ev_done[slot], ctx->stream));
        ctx->stage_used[slot] = true;
    }
    return KPAL_OK;
}

// Position of the first header ('>' at a line start) in a FASTA buffer, or nbytes if none.
static size_t fasta_first_header(const uint8_t *buf, size_t nbytes)
{
    size_t i = 0;
    while (i < nbytes) {
        if (buf[i] == '>') return i;
        const void *nl = memchr(buf + i, '\n', nbytes - i);
        const void *cr = memchr(buf + i, '\r', nbytes - i);
        const uint8_t *e = (const uint8_t *)nl;
        if (cr && (!e || (const uint8_t *)cr < e)) e = (const uint8_t *)cr;
        if (!e) return nbytes;
        i = (size_t)(e - buf) + 1;
    }
    return nbytes;
}

// FASTA text (host) -> flat stream on the device: ctx->fa_flat holds *n_flat bytes afterwards.
static int fasta_flatten_to_device(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes, uint64_t *n_flat)
{
    *n_flat = 0;
    const size_t first = fasta_first_header(host_buf, nbytes);
    if (first >= nbytes) return KPAL_OK;   // no record: nothing to count (klib.py:111 yields nothing)
    const uint64_t n = nbytes - first;
    const uint32_t nblocks = (uint32_t)((n + kFaBlockBytes - 1) / kFaBlockBytes);
    CHK(ensure(ctx, ctx->fa_raw, n + 64));
    CHK(ensure(ctx, ctx->fa_flat, n + 64));
    const size_t meta = (size_t)nblocks * (8 + 8 + 4) + (size_t)(nblocks + 1) * 8 + 64;
    CHK(ensure(ctx, ctx->fa_meta, meta));
    uint8_t *raw = (uint8_t *)ctx->fa_raw.p;
    uint8_t *flat = (uint8_t *)ctx->fa_flat.p;
    long long *last_eol = (long long *)ctx->fa_meta.p;
    long long *carry = last_eol + nblocks;
    uint64_t *offs = (uint64_t *)(carry + nblocks);
    uint32_t *kept = (uint32_t *)(offs + nblocks + 1);
    // fa_raw is free: the previous call synchronised after its last reader (fa_scatter)
    CHK(h2d_staged(ctx, raw, host_buf + first, n));
    LAUNCH(ctx, "fa_last_eol", fa_last_eol_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, n, last_eol);
    LAUNCH(ctx, "fa_carry", fa_carry_kernel, dim3(1), dim3(256), (const long long *)last_eol, nblocks, carry);
    LAUNCH(ctx, "fa_count", fa_count_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, n, (const long long *)carry, kept);
    LAUNCH(ctx, "fa_offset", fa_offset_kernel, dim3(1), dim3(256), (const uint32_t *)kept, nblocks, offs);
    LAUNCH(ctx, "fa_scatter", fa_scatter_kernel, dim3(nblocks), dim3(kFaThreads), (const uint8_t *)raw, n,
           (const long long *)carry, (const uint64_t *)offs, flat);
    HIPCHK(hipMemcpyAsync(n_flat, offs + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

KPAL_API int kpal_count_feed_fasta(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_feed_fasta before kpal_count_begin");
    if (nbytes == 0) return KPAL_OK;
    if (!host_buf) return set_err(KPAL_E_INVALID, "host_buf is NULL");
    uint64_t n_flat = 0;
    CHK(fasta_flatten_to_device(ctx, host_buf, nbytes, &n_flat));
    if (n_flat == 0) return KPAL_OK;
    return count_device_range(ctx, (const uint8_t *)ctx->fa_flat.p, (size_t)n_flat, 0);
}

KPAL_API int kpal_fasta_flatten(kpal_ctx *ctx, const uint8_t *host_buf, size_t nbytes, uint8_t *host_out, uint64_t *n_out)
{
    CTX_ENTER(ctx);
    if (!n_out || (nbytes && (!host_buf || !host_out))) return set_err(KPAL_E_INVALID, "NULL pointer");
    *n_out = 0;
    if (nbytes == 0) return KPAL_OK;
    uint64_t n_flat = 0;
    CHK(fasta_flatten_to_device(ctx, host_buf, nbytes, &n_flat));
    if (n_flat) {
        HIPCHK(hipMemcpyAsync(host_out, ctx->fa_flat.p, n_flat, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    *n_out = n_flat;
    return KPAL_OK;
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

KPAL_API int kpal_count_finish(kpal_ctx *ctx, int64_t *host_out)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_count_finish before kpal_count_begin");
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

KPAL_API int kpal_count_table(kpal_ctx *ctx, void **dev_table, uint64_t *n_bins)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (!ctx->counting) return set_err(KPAL_E_STATE, "no count table (call kpal_count_begin)");
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

// ----------------------------------------------------------------------------------------------
// vector operations
// ----------------------------------------------------------------------------------------------
static unsigned stream_grid(kpal_ctx *ctx, uint64_t n_items, unsigned block = 256)
{
    const uint64_t want = (n_items + block - 1) / block;
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cu * 8));
}

KPAL_API uint64_t kpal_reverse_complement(uint64_t number, int k)
{
    if (k < 1 || k > 32) return 0;
    return revcomp(number, k);
}

// out[i] = in[i] + in[rc(i)]; in == out allowed.  LDS-tiled for k >= 6, pairwise kernels below that.
static int launch_balance(kpal_ctx *ctx, int k, const int64_t *in, int64_t *out)
{
    const uint64_t n = 1ULL << (2 * k);
    if (k >= 6) {
        const unsigned tiles = 1u << (2 * (k - 6));
        LAUNCH(ctx, "balance_tiled", balance_tiled_kernel, dim3(std::min<unsigned>(tiles, (unsigned)ctx->num_cu * 2)), dim3(1024), in, out, k);
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
static int finish_partials(kpal_ctx *ctx, uint32_t nq, uint32_t nblocks, std::vector<Partial> &out)
{
    CHK(ensure(ctx, ctx->result, (size_t)nq * sizeof(Partial)));
    LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(nq), dim3(256), (const Partial *)ctx->partials.p,
           nblocks, (Partial *)ctx->result.p);
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
            const unsigned grid = std::min<unsigned>(1u << (2 * (k - 6)), (unsigned)ctx->num_cu * 2);   // persistent
            CHK(ensure(ctx, ctx->partials, (size_t)grid * sizeof(Partial)));
            Partial *pp = (Partial *)ctx->partials.p;
            if (metric == KPAL_PAIRWISE_PROD) LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<0>), dim3(grid), dim3(1024), l, r, k, pp);
            else if (metric == KPAL_PAIRWISE_SUM) LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<1>), dim3(grid), dim3(1024), l, r, k, pp);
            else LAUNCH(ctx, "pair_distance_balanced", (pair_distance_balanced_kernel<2>), dim3(grid), dim3(1024), l, r, k, pp);
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
static int gram_euclidean(kpal_ctx *ctx, int P, uint64_t n, const int64_t *prof, double *out_lower, bool *exact)
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
    LAUNCH(ctx, "gram_mfma", (gram_mfma_kernel<true>), dim3(gx_d, nd), dim3(256), prof, P, n, dt, pp);
    LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(nd * 4096), dim3(256), (const Partial *)pp, gx_d, res_d);
    if (no) {
        LAUNCH(ctx, "gram_mfma", (gram_mfma_kernel<false>), dim3(gx_o, no), dim3(256), prof, P, n, dt + nd, pp + part_d);
        LAUNCH(ctx, "reduce_partials", reduce_partials_kernel, dim3(no * 4096), dim3(256), (const Partial *)(pp + part_d), gx_o,
               res_d + (size_t)nd * 4096);
    }
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
    // euclidean with enough profiles and bins: fp64 Gram matrix on the matrix cores (gram_kernels.hpp), exact
    // while every |x|^2 < 2^53 (checked on the result); KPAL_MATRIX_MFMA=0 forces the int64 kernels
    static const bool allow_mfma = [] { const char *e = getenv("KPAL_MATRIX_MFMA"); return !e || atoi(e) != 0; }();
    if (metric == KPAL_EUCLIDEAN && allow_mfma && P > 8 && k >= 6) {
        bool exact = false;
        CHK(gram_euclidean(ctx, P, n, prof, out_lower, &exact));
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
    const bool super = allow_super && P > 8 && k >= 6;
    unsigned gx;
    if (super) {
        const int sside = (P + 15) / 16;
        std::vector<int2> supers;
        for (int si = 0; si < sside; ++si)
            for (int sj = 0; sj <= si; ++sj) supers.push_back(make_int2(si, sj));
        const uint32_t nsuper = (uint32_t)supers.size();
        gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n / kSuperBins, std::max<uint64_t>(1, (uint64_t)ctx->num_cu * 8 / nsuper)));
        CHK(ensure(ctx, ctx->scratch[3], (size_t)nsuper * sizeof(int2)));
        HIPCHK(hipMemcpyAsync(ctx->scratch[3].p, supers.data(), (size_t)nsuper * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
        CHK(ensure(ctx, ctx->partials, (size_t)ntiles * TILE * TILE * gx * sizeof(Partial)));
        Partial *pp = (Partial *)ctx->partials.p;
        const int2 *dt = (const int2 *)ctx->scratch[3].p;
        if (metric == 0) LAUNCH(ctx, "matrix_super", (matrix_super_kernel<0>), dim3(gx, nsuper), dim3(256), prof, P, n, dt, pp);
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
    CHK(finish_partials(ctx, ntiles * TILE * TILE, gx, res));
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

// ----------------------------------------------------------------------------------------------
// profiling
// ----------------------------------------------------------------------------------------------
KPAL_API int kpal_prof_enable(kpal_ctx *ctx, int on)
{
    CTX_ENTER(ctx);
    CHK(prof_collect(ctx));
    ctx->prof = on != 0;
    return KPAL_OK;
}

KPAL_API int kpal_prof_reset(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    CHK(prof_collect(ctx));
    std::fill(ctx->prof_ms.begin(), ctx->prof_ms.end(), 0.0);
    std::fill(ctx->prof_launches.begin(), ctx->prof_launches.end(), 0);
    ctx->prof_dropped = 0;
    return KPAL_OK;
}

KPAL_API int kpal_prof_count(kpal_ctx *ctx, int *n_kernels)
{
    CTX_ENTER(ctx);
    CHK(prof_collect(ctx));
    if (n_kernels) *n_kernels = (int)ctx->prof_names.size();
    if (ctx->prof_dropped)   // totals would silently miss launches: say so instead
        return set_err(KPAL_E_HIP, "%llu launches could not be timed (hipEventRecord failed)", (unsigned long long)ctx->prof_dropped);
    return KPAL_OK;
}

KPAL_API int kpal_prof_get(kpal_ctx *ctx, int index, char *name_out, size_t name_cap, double *total_ms,
                           uint64_t *launches)
{
    CTX_ENTER(ctx);
    CHK(prof_collect(ctx));
    if (index < 0 || index >= (int)ctx->prof_names.size()) return set_err(KPAL_E_INVALID, "index out of range");
    if (name_out && name_cap) {
        strncpy(name_out, ctx->prof_names[index].c_str(), name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = ctx->prof_ms[index];
    if (launches) *launches = ctx->prof_launches[index];
    return KPAL_OK;
}

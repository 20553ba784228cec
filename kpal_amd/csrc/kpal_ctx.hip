// kpal_ctx.hip -- errors, context, device memory helpers and the per-kernel timing API of the C-ABI.
#include "kpal_host.hpp"

#include <unistd.h>

// ----------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

KPAL_API const char *kpal_last_error(void) { return g_err; }
KPAL_API const char *kpal_version(void) { return "kpal_amd 0.4 (gfx950)"; }

int ensure(kpal_ctx *ctx, DevBuf &b, size_t bytes)
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

int prof_name_id(kpal_ctx *ctx, const char *name)
{
    for (size_t i = 0; i < ctx->prof_names.size(); ++i)
        if (ctx->prof_names[i] == name) return (int)i;
    ctx->prof_names.push_back(name);
    ctx->prof_ms.push_back(0.0);
    ctx->prof_launches.push_back(0);
    return (int)ctx->prof_names.size() - 1;
}

hipEvent_t prof_event(kpal_ctx *ctx)
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

int prof_collect(kpal_ctx *ctx)
{
    if (ctx->prof_pending.empty()) return KPAL_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIPCHK(hipStreamSynchronize(ctx->comm_stream));
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
    {   // the host NUMA node the GPU hangs on: /sys/bus/pci/devices/<domain:bus:device.function>/numa_node
        char bus[32] = "", path[128];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) == hipSuccess && bus[0]) {
            for (char *q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
            snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
            if (FILE *f = fopen(path, "r")) {
                int node = -1;
                if (fscanf(f, "%d", &node) == 1 && node >= 0) ctx->numa_node = node;
                fclose(f);
            }
        }
        (void)hipGetLastError();
        HostPool::set_preferred_node(ctx->numa_node);   // (the first context of the process decides where the copy threads run)
    }
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
    if (const char *e = getenv("KPAL_FASTA_CHUNK")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 16 && v <= kpal_ctx::kStage) ctx->fa_chunk = (size_t)v;
    }
    if (const char *e = getenv("KPAL_QUAD_STEPS")) ctx->quad_steps_forced = atoi(e);
    if (const char *e = getenv("KPAL_QUAD_STEPS2")) ctx->quad_steps2_forced = atoi(e);
    if (const char *e = getenv("KPAL_QUAD_VERBOSE")) ctx->quad_verbose = atoi(e) != 0;
    if (const char *e = getenv("KPAL_QUAD_REPEAT")) ctx->quad_repeat_forced = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("KPAL_HIST_PACKED")) ctx->quad_hist_unpacked = atoi(e) == 0;
    if (const char *e = getenv("KPAL_DIRECT_SEG")) {   // tests: tiny TableSink segments force the overflow fallback of the FRESH mode
        const long v = atol(e);
        if (v >= 1 && v <= (1 << 20)) {
            ctx->direct_seg = (uint32_t)v;
            ctx->direct_seg_forced = true;
        }
    }
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

KPAL_API int kpal_comm_destroy(kpal_ctx *ctx);

KPAL_API void kpal_ctx_destroy(kpal_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)kpal_comm_destroy(ctx);
    DevBuf *bufs[] = {&ctx->table, &ctx->keys, &ctx->cntmat, &ctx->offs, &ctx->bucket_start, &ctx->slice_start, &ctx->chunk_meta, &ctx->chunk_table, &ctx->chunk_ovf, &ctx->chunk_sorted, &ctx->quad_meta, &ctx->quad_meta2, &ctx->direct_list, &ctx->direct_meta, &ctx->residuals, &ctx->cnt1, &ctx->offs1, &ctx->start1, &ctx->fa_raw[0], &ctx->fa_raw[1], &ctx->fa_flat[0], &ctx->fa_flat[1], &ctx->fa_meta[0], &ctx->fa_meta[1], &ctx->fa_tail, &ctx->rec_raw, &ctx->rec_flat, &ctx->rec_meta, &ctx->rec_starts, &ctx->rec_hdr, &ctx->xsend, &ctx->xrecv, &ctx->dstage[0],
                      &ctx->dstage[1], &ctx->scratch[0], &ctx->scratch[1], &ctx->scratch[2], &ctx->scratch[3],
                      &ctx->partials, &ctx->result, &ctx->canon, &ctx->opt_l, &ctx->opt_r, &ctx->opt_levels, &ctx->opt_profiles};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (ctx->fa_nflat_host) (void)hipHostFree(ctx->fa_nflat_host);
    if (ctx->rec_fd >= 0) (void)close(ctx->rec_fd);
    for (void *p : ctx->host_allocs) (void)hipHostFree(p);
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

KPAL_API int kpal_sync(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    if (ctx->counting) CHK(table_ready(ctx));   // (a pending finalisation of the count table belongs to "everything queued so far")
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIPCHK(hipStreamSynchronize(ctx->comm_stream));   // ... and so does a pipelined reduce
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
    if (ctx->counting) CHK(table_ready(ctx));   // (a pending finalisation may still want to read a fed buffer: kpal_quads2.hip, FRESH)
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

KPAL_API int kpal_memcpy_d2d(kpal_ctx *ctx, void *dev_dst, const void *dev_src, size_t nbytes)
{
    CTX_ENTER(ctx);
    if (nbytes == 0) return KPAL_OK;
    if (ctx->counting) CHK(table_ready(ctx));   // (the source may be the count table)
    HIPCHK(hipMemcpyAsync(dev_dst, dev_src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));   // asynchronous, stream-ordered
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

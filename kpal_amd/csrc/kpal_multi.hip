// kpal_multi.hip -- multi-GPU entry points of the C-ABI: one process per GPU, the per-rank count tables merged by ONE
// RCCL reduce (int64 sum: bit-exact for any reduction order) over xGMI, issued on the context's own HIP streams -- no host
// synchronisation between count, reduce and balance.
//
// The reference has no parallelism of any kind; what is mirrored is Profile.merge with the 'sum' merger
// (kpal/klib.py:269-283, kpal/metrics.py:175): "merging ... is equivalent to first concatenating both fasta files"
// (doc/tutorial.rst:94-95) -- so every rank counts its shard of the reads into a private table and the tables add.
//
// RCCL is bound at run time (dlopen), not at link time: libkpal_hip.so stays loadable on a box without RCCL, and inside a
// PyTorch process the SAME librccl.so PyTorch uses is taken (the Python side passes its path), next to the same HIP runtime.
#include "kpal_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "comm_schedule.hpp"
#include "range_index.hpp"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // the bin-range merge (optional: a library without them only lacks kpal_comm_reduce_scatter_table / _gather_table)
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

RcclApi g_rccl;

int rccl_load(const char *path)
{
    if (g_rccl.handle) return KPAL_OK;
    const char *candidates[] = {path, getenv("KPAL_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    std::string tried;
    for (const char *c : candidates) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : "; ") + c + ": " + (dlerror() ? "cannot be loaded" : "?");
    }
    if (!h) return set_err(KPAL_E_HIP, "RCCL not found (%s)", tried.c_str());
    RcclApi a;
    a.handle = h;
#define KPAL_RCCL_SYM(field, name)                                                              \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                              \
    if (!a.field) {                                                                             \
        dlclose(h);                                                                             \
        return set_err(KPAL_E_HIP, "RCCL library lacks %s", name);                              \
    }
    KPAL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    KPAL_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    KPAL_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    KPAL_RCCL_SYM(Reduce, "ncclReduce")
    KPAL_RCCL_SYM(AllReduce, "ncclAllReduce")
    KPAL_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef KPAL_RCCL_SYM
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    g_rccl = a;
    return KPAL_OK;
}

}  // namespace

#define NCCLCHK(expr)                                                                                                    \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) return set_err(KPAL_E_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static_assert(KPAL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kpal_comm_unique_id hands out an ncclUniqueId");

// Can this process bind RCCL at all (dlopen + every symbol)?  No communicator, no unique id, no bootstrap listener: what every
// rank other than the one that creates the id calls before the collective kpal_comm_init.
KPAL_API int kpal_comm_probe(const char *rccl_library) { return rccl_load(rccl_library); }

KPAL_API int kpal_comm_unique_id(const char *rccl_library, uint8_t *id_out)
{
    if (!id_out) return set_err(KPAL_E_INVALID, "id_out is NULL");
    CHK(rccl_load(rccl_library));
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return KPAL_OK;
}

KPAL_API int kpal_comm_destroy(kpal_ctx *ctx);

KPAL_API int kpal_comm_init(kpal_ctx *ctx, const char *rccl_library, int rank, int world, const uint8_t *id)
{
    CTX_ENTER(ctx);
    if (world < 1 || rank < 0 || rank >= world) return set_err(KPAL_E_INVALID, "bad rank / world size %d / %d", rank, world);
    if (!id) return set_err(KPAL_E_INVALID, "id is NULL");
    if (ctx->comm) return set_err(KPAL_E_STATE, "the context already has a communicator");
    CHK(rccl_load(rccl_library));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    // the pipelined reduce runs on its own stream, ahead of the counting kernels in the hardware queues
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    hipError_t e = hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_table_copied, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->ev_side_free[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        kpal_comm_destroy(ctx);
        return set_err(KPAL_E_HIP, "communicator streams / events: %s", hipGetErrorString(e));
    }
    return KPAL_OK;
}

KPAL_API int kpal_comm_destroy(kpal_ctx *ctx)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    (void)hipSetDevice(ctx->device);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
    if (ctx->ev_table_copied) (void)hipEventDestroy(ctx->ev_table_copied);
    ctx->ev_table_copied = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_side_free[i]) (void)hipEventDestroy(ctx->ev_side_free[i]);
        ctx->ev_side_free[i] = nullptr;
        ctx->side_used[i] = false;
        if (ctx->side[i].p) (void)hipFree(ctx->side[i].p);
        ctx->side[i] = DevBuf();
    }
    ctx->merged = nullptr;
    ctx->merged_bins = 0;
    return KPAL_OK;
}

// count -> reduce -> [balance on root], all on the context's stream: the merged table replaces the count table on root.
KPAL_API int kpal_comm_reduce_table(kpal_ctx *ctx, int root, int balance)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_comm_reduce_table before kpal_count_begin");
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    if (root < 0 || root >= ctx->comm_world) return set_err(KPAL_E_INVALID, "root %d not in 0..%d", root, ctx->comm_world - 1);
    CHK(table_ready(ctx));   // the complete, unbalanced table goes on the wire
    {
        ProfScope ps_(ctx, "rccl_reduce");
        NCCLCHK(g_rccl.Reduce(ctx->table.p, ctx->table.p, (size_t)ctx->bins, ncclInt64, ncclSum, root, (ncclComm_t)ctx->comm, ctx->stream));
    }
    ctx->merged = ctx->table.p;
    ctx->merged_bins = ctx->bins;
    ctx->merged_first = 0;
    if (balance && ctx->comm_rank == root) CHK(launch_balance(ctx, ctx->k, (const int64_t *)ctx->table.p, (int64_t *)ctx->table.p));
    return KPAL_OK;
}

// Pipelined: the table is copied to one of two side buffers (0.05 ms at k = 12) and reduced + balanced THERE, on the
// communicator's stream, while the context's stream goes on with the next count.  The side buffers alternate: the merged
// table of step i stays valid until the reduce of step i+2 is issued.  The schedule itself is comm_schedule.hpp (shared with a
// CPU harness that runs it on a fake runtime under ThreadSanitizer); this is its HIP + RCCL runtime.
namespace {
struct HipCommRuntime {
    kpal_ctx *ctx;
    int rc = KPAL_OK;   // the first library error (its message is in kpal_last_error)
    size_t side_capacity(int t) { return ctx->side[t].p ? ctx->side[t].cap : 0; }
    void *side_ptr(int t) { return ctx->side[t].p; }
    int hip(hipError_t e, const char *what)
    {
        if (e == hipSuccess) return 0;
        rc = set_err(e == hipErrorOutOfMemory ? KPAL_E_NOMEM : KPAL_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return rc;
    }
    int side_grow(int t, size_t bytes) { return rc = ensure(ctx, ctx->side[t], bytes); }
    int host_wait_side_free(int t) { return hip(hipEventSynchronize(ctx->ev_side_free[t]), "hipEventSynchronize"); }
    int main_wait_side_free(int t) { return hip(hipStreamWaitEvent(ctx->stream, ctx->ev_side_free[t], 0), "hipStreamWaitEvent"); }
    int main_copy_table_to_side(int t, size_t bytes) { return hip(hipMemcpyAsync(ctx->side[t].p, ctx->table.p, bytes, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync"); }
    int main_record_copied() { return hip(hipEventRecord(ctx->ev_table_copied, ctx->stream), "hipEventRecord"); }
    int comm_wait_copied() { return hip(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_table_copied, 0), "hipStreamWaitEvent"); }
    int comm_reduce_side(int t, int root)
    {
        std::swap(ctx->stream, ctx->comm_stream);   // (ProfScope works on ctx->stream)
        {
            ProfScope ps_(ctx, "rccl_reduce");
            const ncclResult_t r = g_rccl.Reduce(ctx->side[t].p, ctx->side[t].p, (size_t)ctx->bins, ncclInt64, ncclSum, root, (ncclComm_t)ctx->comm, ctx->stream);
            if (r != ncclSuccess) rc = set_err(KPAL_E_HIP, "ncclReduce failed: %s", g_rccl.GetErrorString(r));
        }
        std::swap(ctx->stream, ctx->comm_stream);
        return rc;
    }
    int comm_balance_side(int t)
    {
        std::swap(ctx->stream, ctx->comm_stream);   // (LAUNCH works on ctx->stream)
        rc = launch_balance(ctx, ctx->k, (const int64_t *)ctx->side[t].p, (int64_t *)ctx->side[t].p);
        std::swap(ctx->stream, ctx->comm_stream);
        return rc;
    }
    int comm_record_side_free(int t) { return hip(hipEventRecord(ctx->ev_side_free[t], ctx->comm_stream), "hipEventRecord"); }
};
}  // namespace

KPAL_API int kpal_comm_reduce_table_async(kpal_ctx *ctx, int root, int balance)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_comm_reduce_table_async before kpal_count_begin");
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    if (root < 0 || root >= ctx->comm_world) return set_err(KPAL_E_INVALID, "root %d not in 0..%d", root, ctx->comm_world - 1);
    CHK(table_ready(ctx));
    HipCommRuntime rt{ctx};
    CommPipeState st;
    st.side_turn = ctx->side_turn;
    st.side_used[0] = ctx->side_used[0];
    st.side_used[1] = ctx->side_used[1];
    st.merged = ctx->merged;
    st.merged_bins = ctx->merged_bins;
    st.merged_first = ctx->merged_first;
    const int rc = comm_reduce_async_schedule(rt, st, ctx->bins, ctx->comm_rank, root, balance != 0);
    ctx->side_turn = st.side_turn;
    ctx->side_used[0] = st.side_used[0];
    ctx->side_used[1] = st.side_used[1];
    ctx->merged = st.merged;
    ctx->merged_bins = st.merged_bins;
    ctx->merged_first = st.merged_first;
    return rc;
}

// ---- bin-range merge (k >= 13: the whole-table reduce to one rank moves and then balances 8 GiB per rank at k = 15) -------------
// ncclReduceScatter leaves rank r the merged bins [r * 4^k / W, (r + 1) * 4^k / W); Profile.balance of that range needs the
// entries rc(i), of which every rank holds 1 / W (range_index.hpp): one all-to-all of 4^k / W^2 entries per pair of ranks,
// packed and unpacked by two permutation kernels.  Nothing leaves the ranks unless a caller wants the whole vector on one
// (kpal_comm_gather_table).
__global__ __launch_bounds__(256) void range_pack_kernel(RangeIndex R, uint32_t rank, const unsigned long long *__restrict__ table, unsigned long long *__restrict__ send)
{
    const uint64_t n1 = R.range_bins(), n2 = R.pair_bins();
    for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n1; l += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = (uint64_t)rank * n1 + l;
        const uint32_t q = R.owner(RangeIndex::revcomp(j, R.k));
        send[(uint64_t)q * n2 + R.pos(j)] = table[j];
    }
}

__global__ __launch_bounds__(256) void range_unpack_kernel(RangeIndex R, uint32_t rank, unsigned long long *__restrict__ table, const unsigned long long *__restrict__ recv)
{
    const uint64_t n1 = R.range_bins(), n2 = R.pair_bins();
    for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n1; l += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (uint64_t)rank * n1 + l;
        const uint64_t j = RangeIndex::revcomp(i, R.k);
        table[i] += recv[(uint64_t)R.owner(j) * n2 + R.pos(j)];     // (klib.py:285-298: a palindrome meets itself -- doubled)
    }
}

static int comm_range_geometry(kpal_ctx *ctx, RangeIndex &R)
{
    const int W = ctx->comm_world;
    if (W < 1 || (W & (W - 1))) return set_err(KPAL_E_INVALID, "the bin-range merge needs a power-of-two number of ranks (%d): use kpal_comm_reduce_table", W);
    int w = 0;
    while ((1 << w) < W) ++w;
    R = RangeIndex{ctx->k, w};
    if (!R.valid()) return set_err(KPAL_E_INVALID, "the bin-range merge needs 2 ceil(log2(W) / 2) <= k (k=%d, %d ranks)", ctx->k, W);
    return KPAL_OK;
}

// The packed range a rank's mirror entries travel in: rank_pack / rank_unpack of the library for callers that move the blocks
// themselves (kpal_amd.dist over torch.distributed): send/recv hold 4^k / W entries, block q = what goes to / came from rank q.
KPAL_API int kpal_range_pack_device(kpal_ctx *ctx, int k, int rank, int world, const int64_t *dev_table, int64_t *dev_send)
{
    CTX_ENTER(ctx);
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world) return set_err(KPAL_E_INVALID, "bad rank / power-of-two world %d / %d", rank, world);
    int w = 0;
    while ((1 << w) < world) ++w;
    const RangeIndex R{k, w};
    if (k < 1 || k > KPAL_MAX_K || !R.valid()) return set_err(KPAL_E_INVALID, "k=%d with %d ranks", k, world);
    if (!dev_table || !dev_send) return set_err(KPAL_E_INVALID, "NULL pointer");
    const unsigned grid = (unsigned)std::min<uint64_t>((R.range_bins() + 255) / 256, (uint64_t)ctx->num_cu * 16);
    LAUNCH(ctx, "range_pack", range_pack_kernel, dim3(grid), dim3(256), R, (uint32_t)rank, (const unsigned long long *)dev_table, (unsigned long long *)dev_send);
    return KPAL_OK;
}

KPAL_API int kpal_range_unpack_device(kpal_ctx *ctx, int k, int rank, int world, int64_t *dev_table, const int64_t *dev_recv)
{
    CTX_ENTER(ctx);
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world) return set_err(KPAL_E_INVALID, "bad rank / power-of-two world %d / %d", rank, world);
    int w = 0;
    while ((1 << w) < world) ++w;
    const RangeIndex R{k, w};
    if (k < 1 || k > KPAL_MAX_K || !R.valid()) return set_err(KPAL_E_INVALID, "k=%d with %d ranks", k, world);
    if (!dev_table || !dev_recv) return set_err(KPAL_E_INVALID, "NULL pointer");
    const unsigned grid = (unsigned)std::min<uint64_t>((R.range_bins() + 255) / 256, (uint64_t)ctx->num_cu * 16);
    LAUNCH(ctx, "range_unpack", range_unpack_kernel, dim3(grid), dim3(256), R, (uint32_t)rank, (unsigned long long *)dev_table, (const unsigned long long *)dev_recv);
    return KPAL_OK;
}

KPAL_API int kpal_comm_reduce_scatter_table(kpal_ctx *ctx, int balance)
{
    CTX_ENTER(ctx);
    if (!ctx->counting) return set_err(KPAL_E_STATE, "kpal_comm_reduce_scatter_table before kpal_count_begin");
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    if (!g_rccl.ReduceScatter || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return set_err(KPAL_E_HIP, "this RCCL library lacks ncclReduceScatter / ncclSend / ncclRecv");
    RangeIndex R{0, 0};
    CHK(comm_range_geometry(ctx, R));
    CHK(table_ready(ctx));
    const uint32_t W = (uint32_t)ctx->comm_world, r = (uint32_t)ctx->comm_rank;
    const uint64_t n1 = R.range_bins(), n2 = R.pair_bins();
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    // Everything of this rank that can fail goes BEFORE the first collective: a rank that returned between the reduce-scatter
    // and the exchange (an allocation of 8 * 4^k / W bytes twice: 1 GiB each at k = 15, W = 8) would leave its peers waiting in
    // the Send / Recv group for ever.
    if (balance) {
        CHK(ensure(ctx, ctx->xsend, (size_t)n1 * 8));
        CHK(ensure(ctx, ctx->xrecv, (size_t)n1 * 8));
    }
    {
        ProfScope ps_(ctx, "rccl_reduce_scatter");
        NCCLCHK(g_rccl.ReduceScatter(table, table + (uint64_t)r * n1, (size_t)n1, ncclInt64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));   // in place
    }
    if (balance) {
        unsigned long long *send = (unsigned long long *)ctx->xsend.p, *recv = (unsigned long long *)ctx->xrecv.p;
        const unsigned grid = (unsigned)std::min<uint64_t>((n1 + 255) / 256, (uint64_t)ctx->num_cu * 16);
        // (a launch or copy that fails here is remembered and reported AFTER the group: the peers' Send / Recv still get their partner)
        int local_rc = KPAL_OK;
        {
            ProfScope ps_(ctx, "range_pack");
            hipLaunchKernelGGL(range_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, R, r, (const unsigned long long *)table, send);
        }
        if (hipGetLastError() != hipSuccess) local_rc = set_err(KPAL_E_HIP, "range_pack did not launch");
        if (hipMemcpyAsync(recv + (uint64_t)r * n2, send + (uint64_t)r * n2, (size_t)n2 * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)   // a rank's own block
            local_rc = set_err(KPAL_E_HIP, "the copy of the rank's own block failed");
        if (W > 1) {
            ProfScope ps_(ctx, "rccl_mirror_exchange");
            NCCLCHK(g_rccl.GroupStart());
            ncclResult_t bad = ncclSuccess;
            for (uint32_t q = 0; q < W; ++q) {
                if (q == r) continue;
                ncclResult_t e = g_rccl.Send(send + (uint64_t)q * n2, (size_t)n2, ncclInt64, (int)q, (ncclComm_t)ctx->comm, ctx->stream);
                if (e == ncclSuccess) e = g_rccl.Recv(recv + (uint64_t)q * n2, (size_t)n2, ncclInt64, (int)q, (ncclComm_t)ctx->comm, ctx->stream);
                if (e != ncclSuccess) bad = e;
            }
            const ncclResult_t ge = g_rccl.GroupEnd();      // (always closed: an open group would swallow every later call)
            if (bad != ncclSuccess || ge != ncclSuccess) return set_err(KPAL_E_HIP, "the mirror exchange failed: %s", g_rccl.GetErrorString(bad != ncclSuccess ? bad : ge));
        }
        if (local_rc != KPAL_OK) return local_rc;
        LAUNCH(ctx, "range_unpack", range_unpack_kernel, dim3(grid), dim3(256), R, r, table, (const unsigned long long *)recv);
    }
    ctx->merged = table + (uint64_t)r * n1;
    ctx->merged_bins = n1;
    ctx->merged_first = (uint64_t)r * n1;
    return KPAL_OK;
}

// the ranges of all ranks into every rank's table (in place): the whole merged vector where a caller wants it
KPAL_API int kpal_comm_gather_table(kpal_ctx *ctx)
{
    CTX_ENTER(ctx);
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    if (!g_rccl.AllGather) return set_err(KPAL_E_HIP, "this RCCL library lacks ncclAllGather");
    RangeIndex R{0, 0};
    CHK(comm_range_geometry(ctx, R));
    unsigned long long *table = (unsigned long long *)ctx->table.p;
    if (!table || ctx->merged != table + (uint64_t)ctx->comm_rank * R.range_bins()) return set_err(KPAL_E_STATE, "kpal_comm_gather_table follows kpal_comm_reduce_scatter_table");
    {
        ProfScope ps_(ctx, "rccl_allgather");
        NCCLCHK(g_rccl.AllGather(table + (uint64_t)ctx->comm_rank * R.range_bins(), table, (size_t)R.range_bins(), ncclInt64, (ncclComm_t)ctx->comm, ctx->stream));
    }
    ctx->merged = table;
    ctx->merged_bins = ctx->bins;
    ctx->merged_first = 0;
    return KPAL_OK;
}

KPAL_API int kpal_comm_merged_range(kpal_ctx *ctx, void **dev_table, uint64_t *first_bin, uint64_t *n_bins)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (!ctx->merged) return set_err(KPAL_E_STATE, "no merged table (kpal_comm_reduce_table[_async] / kpal_comm_reduce_scatter_table)");
    if (dev_table) *dev_table = ctx->merged;
    if (first_bin) *first_bin = ctx->merged_first;
    if (n_bins) *n_bins = ctx->merged_bins;
    return KPAL_OK;
}

KPAL_API int kpal_comm_merged_table(kpal_ctx *ctx, void **dev_table, uint64_t *n_bins)
{
    if (!ctx) return set_err(KPAL_E_INVALID, "ctx is NULL");
    if (!ctx->merged) return set_err(KPAL_E_STATE, "no merged table (kpal_comm_reduce_table[_async]; a kpal_count_begin with another k discards it)");
    if (dev_table) *dev_table = ctx->merged;
    if (n_bins) *n_bins = ctx->merged_bins;
    return KPAL_OK;
}

// ---- distance matrix over bin-range shards --------------------------------------------------------------------
// Partial of vec_kernels.hpp restated (that header's kernels belong to kpal_vec.hip): a double sum and a 64-bit count / wrapping dot
struct PartialPod {
    double s;
    unsigned long long m;
};

__global__ void partials_split_kernel(const PartialPod *__restrict__ p, size_t n, double *__restrict__ s, unsigned long long *__restrict__ m)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        s[i] = p[i].s;
        m[i] = p[i].m;
    }
}

__global__ void partials_join_kernel(PartialPod *__restrict__ p, size_t n, const double *__restrict__ s, const unsigned long long *__restrict__ m)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = PartialPod{s[i], m[i]};
}

int comm_allreduce_partials(kpal_ctx *ctx, void *dev_partials, size_t count)
{
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    if (count == 0) return KPAL_OK;
    CHK(ensure(ctx, ctx->scratch[1], count * 16));
    double *s = (double *)ctx->scratch[1].p;
    unsigned long long *m = (unsigned long long *)(s + count);
    const unsigned grid = (unsigned)std::min<size_t>((count + 255) / 256, 1024);
    LAUNCH(ctx, "partials_split", partials_split_kernel, dim3(grid), dim3(256), (const PartialPod *)dev_partials, count, s, m);
    {
        ProfScope ps_(ctx, "rccl_allreduce");
        NCCLCHK(g_rccl.AllReduce(s, s, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
        NCCLCHK(g_rccl.AllReduce(m, m, count, ncclUint64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    }
    LAUNCH(ctx, "partials_join", partials_join_kernel, dim3(grid), dim3(256), (PartialPod *)dev_partials, count, (const double *)s,
           (const unsigned long long *)m);
    return KPAL_OK;
}

KPAL_API int kpal_comm_distance_matrix_device(kpal_ctx *ctx, int P, uint64_t bin_count, const int64_t *dev_slices, int metric, double *out_lower)
{
    CTX_ENTER(ctx);
    if (!ctx->comm) return set_err(KPAL_E_STATE, "no communicator (kpal_comm_init)");
    // The ranks must take the SAME kernel family: the Gram path all-reduces dot products, the others per-pair sums and counts.
    // Which one a rank would take depends on its own slice (>= 4096 bins, a multiple of 64), so the choice is agreed first:
    // the LDS-staged kernels only if EVERY rank's slice allows them.  A rank whose own arguments are bad must not return before
    // that collective (the others would wait in it for ever): validity rides in the same all-reduce(MIN) -- agreed values
    // 1 = all valid, LDS-staged kernels; 0 = all valid, plain kernels; -1 = some rank's arguments are invalid: every rank returns
    // KPAL_E_INVALID.  (P and metric are the same on all ranks by contract: checked before, they cannot differ.)
    if (P < 1) return set_err(KPAL_E_INVALID, "P must be >= 1");
    if (metric < 0 || metric > 2) return set_err(KPAL_E_INVALID, "unknown metric %d", metric);
    if (P == 1) return KPAL_OK;
    const bool valid = dev_slices && out_lower && bin_count != 0 && ((uintptr_t)dev_slices & 15) == 0;
    int tiled = !valid ? -1 : ((bin_count >= 4096 && bin_count % 64 == 0) ? 1 : 0);
    CHK(ensure(ctx, ctx->result, 64));
    HIPCHK(hipMemcpyAsync(ctx->result.p, &tiled, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(g_rccl.AllReduce(ctx->result.p, ctx->result.p, 1, ncclInt32, ncclMin, (ncclComm_t)ctx->comm, ctx->stream));
    HIPCHK(hipMemcpyAsync(&tiled, ctx->result.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (tiled < 0)
        return set_err(KPAL_E_INVALID, valid ? "another rank's slice is empty, misaligned or NULL" : "every rank needs a non-empty, 16-byte aligned slice and an output array");
    return distance_matrix_core(ctx, P, bin_count, dev_slices, metric, out_lower, true, tiled);
}

// A scalar agreed on by all ranks (bench: the slowest rank's time; tests): max over the ranks, through the device.
KPAL_API int kpal_comm_max_f64(kpal_ctx *ctx, double *inout)
{
    CTX_ENTER(ctx);
    if (!inout) return set_err(KPAL_E_INVALID, "inout is NULL");
    if (!ctx->comm) return KPAL_OK;
    CHK(ensure(ctx, ctx->result, 64));
    HIPCHK(hipMemcpyAsync(ctx->result.p, inout, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(g_rccl.AllReduce(ctx->result.p, ctx->result.p, 1, ncclDouble, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    HIPCHK(hipMemcpyAsync(inout, ctx->result.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KPAL_OK;
}

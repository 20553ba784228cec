// fake_rccl.cpp -- TEST STAND-IN for librccl.so: the dozen entry points libkpal_hip.so binds (kpal_multi.hip: RcclApi), implemented
// between PROCESSES THAT SHARE ONE GPU through a POSIX shared-memory file.  Test infrastructure only: nothing in the product
// names it; a test points KPAL_RCCL_LIBRARY at the built .so.
//
// Why.  The boxes of this pool have one GPU and RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the
// library's kpal_comm_* protocol -- who sends what to whom at which offset, the order of the collectives on every rank, the
// events between the counting stream and the communicator's stream -- had only ever run with a world of one.  With this
// stand-in a world of 2 or 4 real processes, each with its own context, streams and kernels on device 0, runs the real
// library code; only the transport differs.
//
// Semantics kept: every call is a collective in program order; data is read from the device buffer after everything queued
// on `stream` before the call has completed and the result is in place before anything queued on `stream` after it starts.
// Two modes.  Default: the call itself waits for the stream (hipStreamSynchronize), moves the data and returns -- a legal,
// maximally synchronous execution.  KPAL_FAKE_RCCL_ASYNC=1: the call returns at once, as RCCL's do -- it records an event on
// the stream, queues a host function there that holds the stream until the operation is done, and hands the operation to the
// communicator's worker thread, which waits for the event, moves the data on a private stream (KPAL_FAKE_RCCL_DELAY_MS:
// after sleeping (rank + 1) x that long, so that the collective is in flight while the caller races ahead) and releases the
// stream.  A missing event between the caller's streams around a collective then shows as wrong data, as it would with RCCL.  ncclSend / ncclRecv are only valid inside a group and run
// at ncclGroupEnd (one send and one receive per peer and group).  Every wait is bounded (KPAL_FAKE_RCCL_TIMEOUT_S, default
// 120): a rank that never arrives turns into ncclSystemError on the others, not a hang.  KPAL_FAKE_RCCL_FAULT = reduce | recv | early | stall | init
// makes the stand-in lose a contribution / deliver the wrong block / (async) release the stream before the data has moved / hang in a
// collective / hang in ncclCommInitRank: the tests that use it must then FAIL (they are run that way once).
//
//   hipcc -O2 -shared -fPIC -o libfake_rccl.so tests/native/fake_rccl.cpp
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxWorld = 16;

struct Shared {
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;
    uint64_t rounds[kMaxWorld];
};

double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

double timeout_s()
{
    const char *e = getenv("KPAL_FAKE_RCCL_TIMEOUT_S");
    return e && *e ? atof(e) : 120.0;
}

}  // namespace

struct AsyncOp {
    std::vector<hipEvent_t> events;            // recorded on the caller's streams at the call
    std::function<ncclResult_t()> body;
    uint64_t seq = 0;
};

struct ncclComm {
    int rank = 0, world = 1;
    Shared *sh = nullptr;
    uint8_t *slots = nullptr;      // world slots of slot_bytes
    size_t slot_bytes = 0, map_bytes = 0;
    char name[64] = {0};
    uint8_t *slot(int r) const { return slots + (size_t)r * slot_bytes; }
    // KPAL_FAKE_RCCL_ASYNC=1
    bool async = false;
    int device = 0;
    long delay_ms = 0;
    hipStream_t priv = nullptr;    // the worker's copies (non-blocking: never ordered against the caller's streams)
    std::thread worker;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::deque<AsyncOp> q;
    bool stop = false;
    uint64_t issued = 0, done = 0;
    ncclResult_t async_error = ncclSuccess;
};

namespace {

struct P2P {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    ncclComm *comm;
    hipStream_t stream;
};
thread_local int g_group_depth = 0;
thread_local std::vector<P2P> g_group_ops;

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

bool barrier(ncclComm *c)
{
    Shared *sh = c->sh;
    if (sh->failed.load()) return false;
    const uint32_t gen = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
        sh->arrived.store(0);
        sh->generation.fetch_add(1);
        return true;
    }
    const double t0 = now_s(), limit = timeout_s();
    while (sh->generation.load() == gen) {
        if (sh->failed.load()) return false;
        if (now_s() - t0 > limit) {
            sh->failed.store(1);
            fprintf(stderr, "fake_rccl: rank %d waited %.0f s at a barrier of %d ranks\n", c->rank, limit, c->world);
            return false;
        }
        sched_yield();
    }
    return true;
}

template <typename T>
void combine(T *acc, const T *x, size_t n, ncclRedOp_t op)
{
    switch (op) {
    case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = (T)(acc[i] + x[i]); break;
    case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
    case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
    default: break;
    }
}

bool combine_any(void *acc, const void *x, size_t n, ncclDataType_t t, ncclRedOp_t op)
{
    if (op != ncclSum && op != ncclMax && op != ncclMin) return false;
    switch (t) {
    case ncclInt32: combine((int32_t *)acc, (const int32_t *)x, n, op); return true;
    case ncclUint32: combine((uint32_t *)acc, (const uint32_t *)x, n, op); return true;
    case ncclInt64: combine((int64_t *)acc, (const int64_t *)x, n, op); return true;
    case ncclUint64: combine((uint64_t *)acc, (const uint64_t *)x, n, op); return true;
    case ncclFloat64: combine((double *)acc, (const double *)x, n, op); return true;
    case ncclFloat32: combine((float *)acc, (const float *)x, n, op); return true;
    default: return false;
    }
}

// device <-> shared memory: the caller's thread (synchronous mode) or the worker on its private stream
hipError_t dev_copy(ncclComm *c, void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    if (!c->priv) return hipMemcpy(dst, src, n, kind);
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, c->priv);
    return e != hipSuccess ? e : hipStreamSynchronize(c->priv);
}

#define HIPOK(expr)                                                                          \
    do {                                                                                     \
        if ((expr) != hipSuccess) {                                                          \
            fprintf(stderr, "fake_rccl: %s failed\n", #expr);                                \
            return ncclUnhandledCudaError;                                                   \
        }                                                                                    \
    } while (0)

// every rank's `count` elements at send -> their reduction over the ranks at recv of rank `root` (root < 0: of every rank)
ncclResult_t reduce_impl(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm *c)
{
    const size_t es = dtype_size(t);
    const size_t per = c->slot_bytes / es;
    std::vector<uint8_t> acc;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (n) HIPOK(dev_copy(c, c->slot(c->rank), (const uint8_t *)send + off * es, n * es, hipMemcpyDeviceToHost));
        if (!barrier(c)) return ncclSystemError;
        if (n && (root < 0 || root == c->rank)) {
            // (KPAL_FAKE_RCCL_FAULT=reduce: the last rank's contribution is dropped)
            static const bool fault = getenv("KPAL_FAKE_RCCL_FAULT") && !strcmp(getenv("KPAL_FAKE_RCCL_FAULT"), "reduce");
            acc.assign(c->slot(0), c->slot(0) + n * es);
            for (int r = 1; r < c->world - (fault ? 1 : 0); ++r)
                if (!combine_any(acc.data(), c->slot(r), n, t, op)) return ncclInvalidArgument;
            HIPOK(dev_copy(c, (uint8_t *)recv + off * es, acc.data(), n * es, hipMemcpyHostToDevice));
        }
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t reduce_scatter_impl(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm *c)
{
    const size_t es = dtype_size(t);
    const size_t per = c->slot_bytes / es / (size_t)c->world;      // elements per destination and pass
    std::vector<uint8_t> acc;
    for (size_t off = 0; off < recvcount; off += per) {
        const size_t n = recvcount - off < per ? recvcount - off : per;
        for (int d = 0; d < c->world; ++d)
            HIPOK(dev_copy(c, c->slot(c->rank) + (size_t)d * per * es, (const uint8_t *)send + ((size_t)d * recvcount + off) * es, n * es, hipMemcpyDeviceToHost));
        if (!barrier(c)) return ncclSystemError;
        const size_t mine = (size_t)c->rank * per * es;
        acc.assign(c->slot(0) + mine, c->slot(0) + mine + n * es);
        for (int r = 1; r < c->world; ++r)
            if (!combine_any(acc.data(), c->slot(r) + mine, n, t, op)) return ncclInvalidArgument;
        HIPOK(dev_copy(c, (uint8_t *)recv + off * es, acc.data(), n * es, hipMemcpyHostToDevice));
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t all_gather_impl(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm *c)
{
    const size_t es = dtype_size(t);
    const size_t per = c->slot_bytes / es;
    for (size_t off = 0; off < sendcount; off += per) {
        const size_t n = sendcount - off < per ? sendcount - off : per;
        HIPOK(dev_copy(c, c->slot(c->rank), (const uint8_t *)send + off * es, n * es, hipMemcpyDeviceToHost));
        if (!barrier(c)) return ncclSystemError;
        for (int r = 0; r < c->world; ++r)
            HIPOK(dev_copy(c, (uint8_t *)recv + ((size_t)r * sendcount + off) * es, c->slot(r), n * es, hipMemcpyHostToDevice));
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t group_impl(const std::vector<P2P> &ops, ncclComm *c)
{
    // a rank's slot is cut into one region per destination; the passes go on until the longest message of ANY rank is through
    const size_t region = c->slot_bytes / (size_t)c->world;
    uint64_t rounds = 0;
    for (const P2P &o : ops) rounds = std::max<uint64_t>(rounds, (o.bytes + region - 1) / region);
    c->sh->rounds[c->rank] = rounds;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) rounds = std::max<uint64_t>(rounds, c->sh->rounds[r]);
    if (!barrier(c)) return ncclSystemError;
    for (uint64_t p = 0; p < rounds; ++p) {
        for (const P2P &o : ops) {
            const size_t off = (size_t)p * region;
            if (!o.send || off >= o.bytes) continue;
            const size_t n = o.bytes - off < region ? o.bytes - off : region;
            HIPOK(dev_copy(c, c->slot(c->rank) + (size_t)o.peer * region, (const uint8_t *)o.buf + off, n, hipMemcpyDeviceToHost));
        }
        if (!barrier(c)) return ncclSystemError;
        for (const P2P &o : ops) {
            const size_t off = (size_t)p * region;
            if (o.send || off >= o.bytes) continue;
            const size_t n = o.bytes - off < region ? o.bytes - off : region;
            // (KPAL_FAKE_RCCL_FAULT=recv: the data of the wrong region arrives -- the test of the tests, see test_gpu_dist.py)
            static const bool fault = getenv("KPAL_FAKE_RCCL_FAULT") && !strcmp(getenv("KPAL_FAKE_RCCL_FAULT"), "recv");
            const int from_region = fault ? (c->rank + 1) % c->world : c->rank;
            HIPOK(dev_copy(c, (uint8_t *)o.buf + off, c->slot(o.peer) + (size_t)from_region * region, n, hipMemcpyHostToDevice));
        }
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

// (KPAL_FAKE_RCCL_FAULT=stall: from its fourth operation on the LAST rank never moves again -- a hung collective: the other ranks
// wait at their barriers (for KPAL_FAKE_RCCL_TIMEOUT_S), the caller's stream of every rank stands still)
void maybe_stall(ncclComm *c)
{
    static const bool stall = getenv("KPAL_FAKE_RCCL_FAULT") && !strcmp(getenv("KPAL_FAKE_RCCL_FAULT"), "stall");
    static std::atomic<uint32_t> ops{0};
    if (stall && c->rank == c->world - 1 && ops.fetch_add(1) >= 3)
        for (;;) pause();
}

// ---- when an operation runs
struct WaitArg {
    ncclComm *c;
    uint64_t seq;
};

void hold_stream(void *p)      // host function on the caller's stream: returns when operation `seq` is done (no HIP call in here)
{
    WaitArg *a = (WaitArg *)p;
    {
        std::unique_lock<std::mutex> l(a->c->m);
        a->c->cv_done.wait(l, [&] { return a->c->done >= a->seq; });
    }
    delete a;
}

void worker_main(ncclComm *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        AsyncOp op;
        {
            std::unique_lock<std::mutex> l(c->m);
            c->cv.wait(l, [&] { return c->stop || !c->q.empty(); });
            if (c->q.empty()) return;
            op = std::move(c->q.front());
            c->q.pop_front();
        }
        ncclResult_t r = ncclSuccess;
        for (hipEvent_t e : op.events) {
            if (hipEventSynchronize(e) != hipSuccess) r = ncclUnhandledCudaError;
            (void)hipEventDestroy(e);
        }
        if (c->delay_ms > 0) usleep((useconds_t)(c->delay_ms * 1000 * (c->rank + 1)));
        {
            std::unique_lock<std::mutex> l(c->m);
            if (r == ncclSuccess) r = c->async_error;
        }
        // (KPAL_FAKE_RCCL_FAULT=early: the stream is released BEFORE the data has moved -- what a missing stream dependency looks like)
        static const bool early = getenv("KPAL_FAKE_RCCL_FAULT") && !strcmp(getenv("KPAL_FAKE_RCCL_FAULT"), "early");
        if (early) {
            {
                std::unique_lock<std::mutex> l(c->m);
                c->done = op.seq;
            }
            c->cv_done.notify_all();
        }
        maybe_stall(c);
        if (r == ncclSuccess) r = op.body();
        {
            std::unique_lock<std::mutex> l(c->m);
            if (r != ncclSuccess && c->async_error == ncclSuccess) {
                c->async_error = r;
                c->sh->failed.store(1);             // the other ranks must not wait for this one
                fprintf(stderr, "fake_rccl: rank %d: operation %llu failed (%d)\n", c->rank, (unsigned long long)op.seq, (int)r);
            }
            c->done = op.seq;
        }
        c->cv_done.notify_all();
    }
}

ncclResult_t submit(ncclComm *c, const std::vector<hipStream_t> &streams, std::function<ncclResult_t()> body)
{
    if (!c->async) {
        for (hipStream_t s : streams) HIPOK(hipStreamSynchronize(s));
        maybe_stall(c);
        return body();
    }
    AsyncOp op;
    op.body = std::move(body);
    {
        std::unique_lock<std::mutex> l(c->m);
        if (c->async_error != ncclSuccess) return c->async_error;     // (what ncclCommGetAsyncError would report)
        op.seq = ++c->issued;
    }
    for (hipStream_t s : streams) {
        hipEvent_t e;
        HIPOK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPOK(hipEventRecord(e, s));
        op.events.push_back(e);
        HIPOK(hipLaunchHostFunc(s, hold_stream, new WaitArg{c, op.seq}));
    }
    {
        std::unique_lock<std::mutex> l(c->m);
        c->q.push_back(std::move(op));
    }
    c->cv.notify_one();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<uint32_t> serial{0};
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/kpal_fake_rccl_%d_%u_%llx", (int)getpid(), serial.fetch_add(1), (unsigned long long)(now_s() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    // (KPAL_FAKE_RCCL_FAULT=init: the last rank never joins -- a bootstrap that hangs: the others wait in here for KPAL_FAKE_RCCL_TIMEOUT_S)
    if (getenv("KPAL_FAKE_RCCL_FAULT") && !strcmp(getenv("KPAL_FAKE_RCCL_FAULT"), "init") && nranks > 1 && rank == nranks - 1)
        for (;;) pause();
    if (memchr(id.internal, 0, sizeof(id.internal)) == nullptr || strncmp(id.internal, "/kpal_fake_rccl_", 16) != 0) return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->rank = rank;
    c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const char *mb = getenv("KPAL_FAKE_RCCL_SLOT_MB");
    c->slot_bytes = (size_t)(mb && *mb ? atol(mb) : 32) << 20;
    c->map_bytes = 4096 + (size_t)nranks * c->slot_bytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
        perror("fake_rccl: shm_open / ftruncate");
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    static_assert(sizeof(Shared) <= 4096, "header page");
    c->sh = (Shared *)m;                       // (a new shared-memory file reads as zeros: every counter starts at 0)
    c->slots = (uint8_t *)m + 4096;
    c->sh->attached.fetch_add(1);
    const double t0 = now_s(), limit = timeout_s();
    while (c->sh->attached.load() < (uint32_t)nranks) {
        if (now_s() - t0 > limit) {
            fprintf(stderr, "fake_rccl: rank %d: only %u of %d ranks attached after %.0f s\n", rank, c->sh->attached.load(), nranks, limit);
            c->sh->failed.store(1);
            munmap(m, c->map_bytes);
            delete c;
            return ncclSystemError;
        }
        sched_yield();
    }
    if (!barrier(c)) return ncclSystemError;
    if (rank == 0) shm_unlink(c->name);        // everybody has it mapped: the name can go
    const char *as = getenv("KPAL_FAKE_RCCL_ASYNC");
    if (as && *as == '1') {
        const char *dl = getenv("KPAL_FAKE_RCCL_DELAY_MS");
        c->delay_ms = dl && *dl ? atol(dl) : 0;
        if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->priv, hipStreamNonBlocking) != hipSuccess) return ncclUnhandledCudaError;
        c->async = true;
        c->worker = std::thread(worker_main, c);
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    if (c->async) {                            // what is queued runs first
        {
            std::unique_lock<std::mutex> l(c->m);
            c->stop = true;
        }
        c->cv.notify_one();
        c->worker.join();
        (void)hipStreamDestroy(c->priv);
        r = c->async_error;
    }
    munmap((void *)c->sh, c->map_bytes);
    delete c;
    return r;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
    case ncclSystemError: return "fake_rccl: a rank did not arrive (timeout) or shared memory failed";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: error";
    }
}

ncclResult_t ncclReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t stream)
{
    if (!c || root < 0 || root >= c->world || g_group_depth) return ncclInvalidUsage;
    if (!dtype_size(t)) return ncclInvalidArgument;
    return submit(c, {stream}, [=] { return reduce_impl(send, recv, count, t, op, root, c); });
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (!c || g_group_depth) return ncclInvalidUsage;
    if (!dtype_size(t)) return ncclInvalidArgument;
    return submit(c, {stream}, [=] { return reduce_impl(send, recv, count, t, op, -1, c); });
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (!c || g_group_depth) return ncclInvalidUsage;
    if (!dtype_size(t)) return ncclInvalidArgument;
    return submit(c, {stream}, [=] { return reduce_scatter_impl(send, recv, recvcount, t, op, c); });
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t stream)
{
    if (!c || g_group_depth) return ncclInvalidUsage;
    if (!dtype_size(t)) return ncclInvalidArgument;
    return submit(c, {stream}, [=] { return all_gather_impl(send, recv, sendcount, t, c); });
}

ncclResult_t ncclGroupStart()
{
    ++g_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream)
{
    if (!c || !g_group_depth || peer < 0 || peer >= c->world || peer == c->rank || !dtype_size(t)) return ncclInvalidUsage;
    g_group_ops.push_back(P2P{true, const_cast<void *>(buf), count * dtype_size(t), peer, c, stream});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream)
{
    if (!c || !g_group_depth || peer < 0 || peer >= c->world || peer == c->rank || !dtype_size(t)) return ncclInvalidUsage;
    g_group_ops.push_back(P2P{false, buf, count * dtype_size(t), peer, c, stream});
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth) return ncclSuccess;
    std::vector<P2P> ops;
    ops.swap(g_group_ops);
    if (ops.empty()) return ncclSuccess;
    ncclComm *c = ops[0].comm;
    bool sent[kMaxWorld] = {false}, received[kMaxWorld] = {false};
    std::vector<hipStream_t> streams;
    for (const P2P &o : ops) {
        if (o.comm != c) return ncclInvalidUsage;
        bool *seen = o.send ? sent : received;
        if (seen[o.peer]) return ncclInvalidUsage;       // (one send and one receive per peer and group)
        seen[o.peer] = true;
        if (std::find(streams.begin(), streams.end(), o.stream) == streams.end()) streams.push_back(o.stream);
    }
    return submit(c, streams, [=] { return group_impl(ops, c); });
}

}  // extern "C"

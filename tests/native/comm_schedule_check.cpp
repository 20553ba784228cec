// comm_schedule_check.cpp -- the stream / event schedule of the pipelined table reduce (kpal_amd/csrc/comm_schedule.hpp: the code
// kpal_comm_reduce_table_async runs on HIP streams and RCCL) on a FAKE runtime, under ThreadSanitizer: W ranks as threads; a
// stream is a worker thread executing queued tasks in order; an event is a flag a stream sets and streams / the host wait for;
// the reduce is a rendezvous of the ranks' communicator streams that adds their side buffers onto the root's; the balance is an
// in-place transform on the root; the "count" of the next step is a main-stream task that overwrites the table while the
// communicator's stream still works on the side buffers.  Checked per step: the merged table (read after both streams caught up,
// like kpal_sync + kpal_comm_merged_table) is balance(sum of the ranks' tables of THAT step), the merged table of step i is still
// intact while step i + 1 is in flight (two reduces in flight at most), buffers that grow mid-run are re-allocated only after
// their last reader.  A missing wait in the schedule is a data race ThreadSanitizer reports (or a wrong value).
// Test infrastructure; run by tests/test_native_sanitized.py.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../kpal_amd/csrc/comm_schedule.hpp"

using namespace kpal;

struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void push(std::function<void()> f)
    {
        {
            std::unique_lock<std::mutex> lk(m);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void sync()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.empty() && !busy; });
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::unique_lock<std::mutex> lk(m);
                busy = false;
            }
            cv.notify_all();
        }
    }
};

// an event: record = "a task on a stream bumps `done` to the ticket taken at record time"; wait = block until done >= that ticket
struct Event {
    std::mutex m;
    std::condition_variable cv;
    uint64_t issued = 0, done = 0;
    uint64_t record_on(Stream &s)
    {
        uint64_t ticket;
        {
            std::unique_lock<std::mutex> lk(m);
            ticket = ++issued;
        }
        s.push([this, ticket] {
            {
                std::unique_lock<std::mutex> lk(m);
                if (done < ticket) done = ticket;
            }
            cv.notify_all();
        });
        return ticket;
    }
    uint64_t last()
    {
        std::unique_lock<std::mutex> lk(m);
        return issued;
    }
    void wait(uint64_t ticket)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= ticket; });
    }
    void stream_wait(Stream &s)     // the stream waits for the event as recorded so far
    {
        const uint64_t ticket = last();
        s.push([this, ticket] { wait(ticket); });
    }
};

// rendezvous of W communicator streams: everyone contributes a buffer, the root's receives the sum
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int W, arrived = 0;
    uint64_t round = 0;
    std::vector<std::vector<int64_t> *> bufs;
    explicit Rendezvous(int w) : W(w), bufs((size_t)w, nullptr) {}
    void reduce(int rank, int root, std::vector<int64_t> *buf)
    {
        std::unique_lock<std::mutex> lk(m);
        bufs[(size_t)rank] = buf;
        const uint64_t my_round = round;
        if (++arrived == W) {
            std::vector<int64_t> &dst = *bufs[(size_t)root];
            for (int r = 0; r < W; ++r)
                if (r != root)
                    for (size_t i = 0; i < dst.size(); ++i) dst[i] += (*bufs[(size_t)r])[i];
            arrived = 0;
            ++round;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return round != my_round; });
        }
    }
};

static int64_t balanced(int64_t v) { return 2 * v + 1; }

struct FakeRuntime {
    int rank;
    Rendezvous *rv;
    Stream main, comm;
    Event copied, side_free[2];
    std::vector<int64_t> table;
    std::vector<int64_t> *side[2] = {nullptr, nullptr};
    ~FakeRuntime()
    {
        main.sync();
        comm.sync();
        delete side[0];
        delete side[1];
    }
    size_t side_capacity(int t) { return side[t] ? side[t]->size() * sizeof(int64_t) : 0; }
    void *side_ptr(int t) { return side[t]; }
    int side_grow(int t, size_t bytes)
    {
        delete side[t];                                   // (the schedule has waited for the buffer's last reader)
        side[t] = new std::vector<int64_t>(bytes / sizeof(int64_t));
        return 0;
    }
    int host_wait_side_free(int t)
    {
        side_free[t].wait(side_free[t].last());
        return 0;
    }
    int main_wait_side_free(int t)
    {
        side_free[t].stream_wait(main);
        return 0;
    }
    int main_copy_table_to_side(int t, size_t bytes)
    {
        std::vector<int64_t> *dst = side[t];
        main.push([this, dst, bytes] {
            for (size_t i = 0; i < bytes / sizeof(int64_t); ++i) (*dst)[i] = table[i];
        });
        return 0;
    }
    int main_record_copied()
    {
        copied.record_on(main);
        return 0;
    }
    int comm_wait_copied()
    {
        copied.stream_wait(comm);
        return 0;
    }
    int comm_reduce_side(int t, int root)
    {
        std::vector<int64_t> *buf = side[t];
        comm.push([this, buf, root] {
            std::this_thread::sleep_for(std::chrono::microseconds(300));   // (the wire is slower than the count)
            rv->reduce(rank, root, buf);
        });
        return 0;
    }
    int comm_balance_side(int t)
    {
        std::vector<int64_t> *buf = side[t];
        comm.push([buf] {
            for (auto &v : *buf) v = balanced(v);
        });
        return 0;
    }
    int comm_record_side_free(int t)
    {
        side_free[t].record_on(comm);
        return 0;
    }
};

static std::atomic<int> failures{0};

static int64_t value_of(int rank, int step, size_t i) { return (int64_t)(rank + 1) * 1000003 + (int64_t)step * 7919 + (int64_t)(i % 97); }

static void verify(int W, int step, size_t bins, const std::vector<int64_t> &m)
{
    if (m.size() != bins) ++failures;
    for (size_t i = 0; i < bins && i < m.size(); ++i) {
        int64_t sum = 0;
        for (int r = 0; r < W; ++r) sum += value_of(r, step, i);
        if (m[i] != balanced(sum)) {
            ++failures;
            break;
        }
    }
}

// The steps are PIPELINED as bench.py issues them: count i + 1 and its reduce are queued while the reduce of step i may still
// run; the host never waits for a stream inside the loop.  Step i's merged table is verified after step i + 1 has been issued --
// the host waits for step i's side_free event only -- i.e. while step i + 1 is in flight: it must be intact until step i + 2.
static void rank_main(int rank, int W, Rendezvous *rv, int steps, int root)
{
    FakeRuntime rt;
    rt.rank = rank;
    rt.rv = rv;
    CommPipeState st;
    size_t bins = 1 << 12;
    rt.table.assign(bins, 0);
    struct Pending {
        int step = -1, t = 0;
        uint64_t ticket = 0;
        size_t bins = 0;
        const std::vector<int64_t> *buf = nullptr;
    } pending;
    auto check_pending = [&] {
        if (pending.step < 0) return;
        rt.side_free[pending.t].wait(pending.ticket);     // the reduce + balance of that step are done; nothing else is waited for
        if (rank == root) verify(W, pending.step, pending.bins, *pending.buf);
        pending.step = -1;
    };
    for (int step = 0; step < steps; ++step) {
        if (step == steps / 2) {                          // "kpal_count_begin with a larger k": the table and the side buffers grow
            pending.step = -1;                            // (not looked at: nothing may order the communicator's stream here but the schedule)
            rt.main.sync();                               // (the host resizes the table: like ensure(), after the stream is idle)
            bins *= 4;
            rt.table.assign(bins, 0);
        }
        // the count of this step: a main-stream task that overwrites the table (side buffers of earlier steps may still be in use)
        rt.main.push([&rt, rank, step, bins] {
            for (size_t i = 0; i < bins; ++i) rt.table[i] = value_of(rank, step, i);
        });
        if (comm_reduce_async_schedule(rt, st, bins, rank, root, true) != 0) ++failures;
        Pending now;
        now.step = step;
        now.t = st.side_turn;
        now.ticket = rt.side_free[st.side_turn].last();
        now.bins = bins;
        now.buf = static_cast<const std::vector<int64_t> *>(st.merged);
        if (st.merged_bins != bins) ++failures;
        // every third step the host looks at the step before (while this one is in flight); otherwise it does not wait for
        // anything: the main stream runs ahead of the communicator's (whose reduce is slow: a rendezvous of all ranks + a nap), and
        // only the schedule's own waits keep the copy of step i + 2 off the buffer the reduce of step i still works on
        if (step % 3 == 0) check_pending();
        else pending.step = -1;
        pending = now;
    }
    check_pending();
    rt.main.sync();
    rt.comm.sync();
}

int main()
{
    for (int W : {1, 2, 3}) {
        for (int root = 0; root < W; ++root) {
            Rendezvous rv(W);
            std::vector<std::thread> ranks;
            for (int r = 0; r < W; ++r) ranks.emplace_back(rank_main, r, W, &rv, 12, root);
            for (auto &t : ranks) t.join();
        }
    }
    if (failures.load()) {
        std::printf("comm_schedule_check: %d failure(s)\n", failures.load());
        return 1;
    }
    std::printf("comm_schedule_check: worlds 1..3, every root, 12 pipelined steps each\nSANITIZE_OK\n");
    return 0;
}

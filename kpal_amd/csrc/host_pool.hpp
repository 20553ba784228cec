// host_pool.hpp -- a small pool of host threads for the copies that feed the GPU: page cache -> pinned staging (pread) and
// pageable memory -> pinned staging (memcpy).  One thread moves 7-9 GB/s, the link takes 56.  The MI355X host has two sockets
// (256 hardware threads, two NUMA nodes) and a GPU hangs on ONE of them: the pool's threads are bound to the CPUs of the GPU's
// node (set_preferred_node: the first context says which; /sys/devices/system/node/nodeN/cpulist) and the pinned staging
// buffers are allocated there (kpal_count.hip: ensure_pinned) -- left to the scheduler the same 16 threads gave 25 to 42 GB/s
// end to end from one run to the next, depending on where they and the buffers happened to land; spread over the L3 domains of
// BOTH sockets 28.  KPAL_READ_THREADS (default 16) sizes the pool, KPAL_READ_PIN=0 leaves its threads unbound.
// The pool lives for the process (its threads sleep on a condition variable between jobs).
#pragma once
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace kpal {

class HostPool {
public:
    // the NUMA node the pool's threads should run on (-1: unknown / no binding); takes effect if called before the first use
    static void set_preferred_node(int node) { preferred_node() = node; }
    static HostPool &instance()
    {
        static HostPool *pool = new HostPool();   // never destroyed: its threads may outlive static destructors
        return *pool;
    }
    int size() const { return (int)workers_.size() + 1; }   // the caller takes part

    // fn(0) .. fn(ntasks - 1) on the workers and the calling thread; returns when all are done.  No exceptions may leave fn.
    void run(int ntasks, const std::function<void(int)> &fn)
    {
        start(ntasks, fn);
        help(generation_of_user_);
        wait();
    }
    // The same without the calling thread (it has other work): wait() later.  One job at a time: start() blocks while another
    // thread's job runs (contexts are single-threaded, but two threads may each own one).
    void start(int ntasks, const std::function<void(int)> &fn)
    {
        user_.lock();
        {
            std::unique_lock<std::mutex> lk(m_);
            fn_ = fn;
            ntasks_ = ntasks;
            next_ = 0;
            done_ = 0;
            generation_of_user_ = ++generation_;
        }
        cv_work_.notify_all();
    }
    void wait()
    {
        help(generation_of_user_);   // (also the whole job when no worker thread could be created)
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return done_ >= ntasks_; });
        }
        user_.unlock();
    }

private:
    static int &preferred_node()
    {
        static int node = -1;
        return node;
    }
    HostPool()
    {
        const char *e = getenv("KPAL_READ_THREADS");
        int n = e ? atoi(e) : 16;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n > hw) n = (int)hw;
        n = n < 1 ? 1 : (n > 64 ? 64 : n);
        const char *p = getenv("KPAL_READ_PIN");
        const bool pin = !p || atoi(p) != 0;
        cpu_set_t node_cpus;
        const bool have = pin && node_cpu_set(preferred_node(), &node_cpus);
        for (int w = 0; w + 1 < n; ++w) {
            try {
                workers_.emplace_back([this] { loop(); });
            } catch (...) {
                break;
            }
            if (have) (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof(node_cpus), &node_cpus);   // (a refusal changes nothing)
        }
        for (auto &t : workers_) t.detach();
    }

    // the CPUs of NUMA node `node` this process may run on (/sys/devices/system/node/nodeN/cpulist: "0-63,128-191")
    static bool node_cpu_set(int node, cpu_set_t *out)
    {
        if (node < 0) return false;
        char path[96], line[4096];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = fopen(path, "r");
        if (!f) return false;
        const bool ok = fgets(line, sizeof(line), f) != nullptr;
        fclose(f);
        if (!ok) return false;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
        CPU_ZERO(out);
        int count = 0;
        for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int lo = 0, hi = 0;
            const int got = sscanf(tok, "%d-%d", &lo, &hi);
            if (got < 1) continue;
            if (got == 1) hi = lo;
            for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
                if (c >= 0 && CPU_ISSET(c, &allowed)) {
                    CPU_SET(c, out);
                    ++count;
                }
        }
        return count > 0;
    }

    // Tasks are claimed and retired under the lock, tagged with the job they belong to: a worker that wakes late can never take
    // a task index of the job before (tasks are megabytes of copying each; the lock is noise).
    void help(uint64_t gen)
    {
        for (;;) {
            std::function<void(int)> fn;
            int t;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (gen != generation_ || next_ >= ntasks_) return;
                t = next_++;
                fn = fn_;
            }
            fn(t);
            {
                std::unique_lock<std::mutex> lk(m_);
                if (gen == generation_ && ++done_ >= ntasks_) cv_done_.notify_all();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            help(seen);
        }
    }

    std::vector<std::thread> workers_;
    std::mutex m_, user_;
    std::condition_variable cv_work_, cv_done_;
    std::function<void(int)> fn_;
    int next_ = 0, ntasks_ = 0, done_ = 0;
    uint64_t generation_ = 0, generation_of_user_ = 0;
};

}  // namespace kpal

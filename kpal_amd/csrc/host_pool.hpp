// host_pool.hpp -- a small pool of host threads for the copies that feed the GPU: page cache -> pinned staging (pread) and
// pageable memory -> pinned staging (memcpy).  One thread moves 7-9 GB/s, the link takes 56.  Measured on the MI355X host (two
// sockets, 256 hardware threads; tools/clibench.py, 8 GB FASTA in tmpfs): 16 threads left to the scheduler 42 GB/s end to end, 8
// threads 29-34; spreading the workers over the L3 domains of BOTH sockets (KPAL_READ_PIN=1) 28 GB/s -- half of them then sit
// on the socket the pinned buffer and the GPU are not attached to.  So: 16 workers (KPAL_READ_THREADS), not pinned.
// The pool lives for the process (its threads sleep on a condition variable between jobs).
#pragma once
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace kpal {

class HostPool {
public:
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
    HostPool()
    {
        const char *e = getenv("KPAL_READ_THREADS");
        int n = e ? atoi(e) : 16;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n > hw) n = (int)hw;
        n = n < 1 ? 1 : (n > 64 ? 64 : n);
        const char *p = getenv("KPAL_READ_PIN");
        const bool pin = p && atoi(p) != 0;
        std::vector<std::vector<int>> domains;
        if (pin) domains = l3_domains();
        for (int w = 0; w + 1 < n; ++w) {
            try {
                workers_.emplace_back([this] { loop(); });
            } catch (...) {
                break;
            }
            if (domains.size() > 1) {
                const std::vector<int> &d = domains[(size_t)(w + 1) % domains.size()];
                cpu_set_t set;
                CPU_ZERO(&set);
                for (int c : d)
                    if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
                (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof(set), &set);   // (a refusal changes nothing)
            }
        }
        for (auto &t : workers_) t.detach();
    }

    // the CPUs this process may run on, grouped by the L3 cache they share
    static std::vector<std::vector<int>> l3_domains()
    {
        std::vector<std::vector<int>> out;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
        std::map<std::string, std::vector<int>> by_l3;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &allowed)) continue;
            char path[128], line[256];
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
            FILE *f = fopen(path, "r");
            if (!f) return std::vector<std::vector<int>>();
            const bool ok = fgets(line, sizeof(line), f) != nullptr;
            fclose(f);
            if (!ok) return std::vector<std::vector<int>>();
            by_l3[line].push_back(c);
        }
        for (auto &kv : by_l3) out.push_back(kv.second);
        return out;
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

// host_pool_check.cpp -- ThreadSanitizer / AddressSanitizer harness of kpal_amd/csrc/host_pool.hpp (the copy threads behind
// the host feeds and the FASTA reader): many jobs, from two user threads at once (start() serialises them), run() and the
// start() ... wait() form, jobs with fewer and with many more tasks than threads, tasks of uneven length (workers that wake
// late must never take a task of the job before: every task index of every job is executed exactly once), jobs of zero
// tasks.  KPAL_READ_THREADS sizes the pool (1: no workers at all -- the caller does everything); the test runs the binary
// with 1, 2 and 16.  Test infrastructure; run by tests/test_native_sanitized.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../kpal_amd/csrc/host_pool.hpp"

using kpal::HostPool;

static std::atomic<long> failures{0};

static void user(int id, int jobs)
{
    HostPool &pool = HostPool::instance();
    uint64_t rng = 0x9E3779B97F4A7C15ull * (uint64_t)(id + 1);
    auto rnd = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    for (int j = 0; j < jobs; ++j) {
        const int ntasks = (int)(rnd() % 40);                       // 0 .. 39 (more and fewer than the pool has threads)
        std::vector<int> hits((size_t)ntasks, 0);                   // plain ints: two executions of one task are a data race TSan sees
        std::vector<long> sums((size_t)ntasks, 0);
        const bool slow = (rnd() & 7) == 0;
        auto fn = [&](int t) {
            hits[(size_t)t] += 1;
            long s = 0;
            const int spin = slow && (t & 1) ? 20000 : 50;          // uneven tasks: some workers are still busy when the job ends
            for (int i = 0; i < spin; ++i) s += (long)i * (t + 1);
            sums[(size_t)t] = s;
            if (slow && t == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        };
        if (j & 1) {
            pool.run(ntasks, fn);
        } else {
            pool.start(ntasks, fn);
            long other = 0;                                          // the caller's own work between start() and wait()
            for (int i = 0; i < 1000; ++i) other += i;
            if (other < 0) ++failures;
            pool.wait();
        }
        for (int t = 0; t < ntasks; ++t)
            if (hits[(size_t)t] != 1) ++failures;
    }
}

int main()
{
    HostPool &pool = HostPool::instance();
    std::printf("host_pool_check: pool of %d (KPAL_READ_THREADS)\n", pool.size());
    user(0, 200);                                                    // one user
    std::thread a(user, 1, 300), b(user, 2, 300);                    // two users at once
    a.join();
    b.join();
    user(3, 50);
    if (failures.load()) {
        std::printf("host_pool_check: %ld task(s) executed a wrong number of times\n", failures.load());
        return 1;
    }
    std::printf("SANITIZE_OK\n");
    return 0;
}

// Stress of the library's host-side thread helpers (csrc/host_threads.hpp) for ThreadSanitizer: tests/test_host_threads_cpu.py builds this
// with g++ -fsanitize=thread and runs it.  No GPU, no HIP: the helpers are plain C++.
//   HostPool: run() from several caller threads (serialised inside), with and without a preceding warm(), warm() without a run() after it
//   Uploader: submit() / wait() pairs whose job spins on a flag the submitter sets later (the shape of msm_partial_batch's use), a job that
//             throws, one uploader per slot from concurrent caller threads
#include <cstdio>
#include <cstdlib>
#include "host_threads.hpp"

static std::atomic<long> g_sum{0};

int main() {
    int bad = 0;
    // ---- HostPool
    for (int round = 0; round < 300; round++) {
        if (round % 3 == 0) host_pool().warm(std::chrono::microseconds(200));
        if (round % 7 == 6) { host_pool().warm(std::chrono::microseconds(50)); continue; }  // a warm() that no run() follows
        const size_t n = 1 + (size_t)(round % 5);
        std::vector<long> out(n, 0);
        const std::function<void(size_t)> job = [&](size_t i) {
            long s = 0;
            for (int k = 0; k < 2000; k++) s += (long)(i + 1) * k;
            out[i] = s;
        };
        host_pool().run(job, n);
        for (size_t i = 0; i < n; i++)
            if (out[i] != (long)(i + 1) * (2000L * 1999 / 2)) bad++;
    }
    {   // several caller threads: run() serialises them
        std::vector<std::thread> callers;
        for (int t = 0; t < 4; t++)
            callers.emplace_back([&, t] {
                for (int r = 0; r < 50; r++) {
                    if ((r + t) & 1) host_pool().warm(std::chrono::microseconds(100));
                    const std::function<void(size_t)> job = [&](size_t i) { g_sum.fetch_add((long)i + 1, std::memory_order_relaxed); };
                    host_pool().run(job, 3);
                }
            });
        for (auto& c : callers) c.join();
        if (g_sum.load() != 4L * 50 * 6) bad++;
    }
    // ---- Uploader
    for (int round = 0; round < 200; round++) {
        std::atomic<int> go{0};
        std::atomic<uint64_t> issued{0};
        long payload[4] = {0, 0, 0, 0};
        uploader(0).submit([&]() -> int {
            int g;
            while ((g = go.load(std::memory_order_acquire)) == 0) { }
            if (g < 0) return ZKP_HOST_THREADS_OK;
            for (int k = 0; k < 4; k++) {
                payload[k] = round + k;
                issued.store((uint64_t)k + 1, std::memory_order_release);
            }
            return round % 11 == 10 ? ZKP_HOST_THREADS_E_DEVICE : ZKP_HOST_THREADS_OK;
        });
        go.store(round % 13 == 12 ? -1 : 1, std::memory_order_release);
        if (round % 13 != 12)
            for (uint64_t k = 1; k <= 4; k++) {
                while (issued.load(std::memory_order_acquire) < k) { }
                if (payload[k - 1] != round + (long)k - 1) bad++;
            }
        const int rc = uploader(0).wait();
        const int want = (round % 13 != 12 && round % 11 == 10) ? ZKP_HOST_THREADS_E_DEVICE : ZKP_HOST_THREADS_OK;
        if (rc != want) bad++;
    }
    uploader(1).submit([]() -> int { throw 1; });
    if (uploader(1).wait() != ZKP_HOST_THREADS_E_DEVICE) bad++;
    {   // one uploader per slot, concurrent callers
        std::vector<std::thread> callers;
        std::atomic<int> errs{0};
        for (int slot = 0; slot < 4; slot++)
            callers.emplace_back([&, slot] {
                for (int r = 0; r < 50; r++) {
                    std::atomic<int> done{0};
                    uploader(slot).submit([&]() -> int { done.store(r + 1, std::memory_order_release); return ZKP_HOST_THREADS_OK; });
                    if (uploader(slot).wait() != ZKP_HOST_THREADS_OK || done.load(std::memory_order_acquire) != r + 1) errs++;
                }
            });
        for (auto& c : callers) c.join();
        bad += errs.load();
    }
    std::printf("host threads stress: %d failures\n", bad);
    return bad ? 1 : 0;
}

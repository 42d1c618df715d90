// Host-side thread helpers of the library (pure C++17, no HIP): the resident pool for the serial tails of MSM batches and the uploader
// thread of host-fed MSMs.  In a header of their own so that tests/abi/host_threads_stress.cpp can run them under ThreadSanitizer
// (tests/test_host_threads_cpu.py); api.hip includes this file inside its anonymous namespace.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#ifndef ZKP_HOST_THREADS_E_DEVICE
#define ZKP_HOST_THREADS_E_DEVICE (-3)  // = ZKP_E_DEVICE (include/zkp_hip.h); api.hip asserts the equality
#endif
#ifndef ZKP_HOST_THREADS_OK
#define ZKP_HOST_THREADS_OK 0
#endif

// ----------------------------------------------------------------------------------------------------
// A few resident host threads for the short serial chains that follow a batch of MSMs (one Horner chain of ~35 group
// operations, ~30 us, per MSM).  Creating threads per call cost as much as the chains themselves (three chains: ~100 us with
// std::thread per call, the same as running them one after the other).  The workers are detached and the pool is never
// destroyed: they sleep on a condition variable between calls and end with the process.  run() serialises its callers; the jobs
// are pure host arithmetic and take no other lock.
// ----------------------------------------------------------------------------------------------------
class HostPool {
    std::mutex run_mu;  // one run() at a time
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void(size_t)>* fn = nullptr;
    size_t next = 0, total = 0, finished = 0;
    bool started = false;
    // warm(): a caller that knows a run() is coming within the next few hundred microseconds (the MSM is waiting for its last kernel)
    // wakes the workers early; they spin on `posted` until the job arrives or the deadline passes.  A sleeping thread takes 20-50 us
    // (at times hundreds) to come back, as long as the chains it is woken for (profiles/r05_q_host_pool_warm.md).
    std::atomic<bool> posted{false};
    std::chrono::steady_clock::time_point warm_until{};
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return (fn != nullptr && next < total) || std::chrono::steady_clock::now() < warm_until; });
            if (!(fn != nullptr && next < total)) {  // woken early: spin outside the lock until a job is posted or the deadline passes
                const auto until = warm_until;
                lk.unlock();
                while (!posted.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
                lk.lock();
                if (!(fn != nullptr && next < total)) {
                    if (std::chrono::steady_clock::now() >= warm_until) warm_until = {};  // back to sleep
                    continue;
                }
            }
            const size_t i = next++;
            const std::function<void(size_t)>* f = fn;
            lk.unlock();
            (*f)(i);
            lk.lock();
            if (++finished == total) cv_done.notify_one();
        }
    }

public:
    void run(const std::function<void(size_t)>& f, size_t n) {
        std::lock_guard<std::mutex> one(run_mu);
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            started = true;
            for (int i = 0; i < 3; i++) std::thread([this] { worker(); }).detach();
        }
        fn = &f;
        next = 0;
        total = n;
        finished = 0;
        warm_until = {};  // (a worker that finds no job left sleeps until the next run or warm)
        posted.store(true, std::memory_order_release);
        cv.notify_all();
        while (next < total) {  // the caller works too
            const size_t i = next++;
            lk.unlock();
            f(i);
            lk.lock();
            ++finished;
        }
        cv_done.wait(lk, [&] { return finished == total; });
        fn = nullptr;
        posted.store(false, std::memory_order_release);
        warm_until = {};
    }
    void warm(std::chrono::microseconds how_long) {
        std::unique_lock<std::mutex> lk(mu, std::try_to_lock);  // never wait for it: a run() in progress needs no warming
        if (!lk.owns_lock() || !started) return;
        warm_until = std::chrono::steady_clock::now() + how_long;
        cv.notify_all();
    }
};
HostPool& host_pool() {
    static HostPool* pool = new HostPool;  // intentionally leaked, see above
    return *pool;
}

// One resident host thread that issues the uploads of the later scalar ranges of a host-fed MSM (zkp_msm_g1) while the caller's thread
// enqueues the kernels of the first range: hipMemcpyAsync from pageable memory holds its caller for most of the transfer, and issued in
// line -- after the dozen launches of the first range -- the second upload started 100 us late and ended after the first range's kernels
// (profiles/r05_o_range_handover.md).  Same life cycle as the pool above: created on first use, detached, ends with the process.
// submit() hands over one job; wait() returns its result once it has run (every submit is followed by exactly one wait).
class Uploader {
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::function<int()> job;
    bool pending = false, running = false, started = false;
    int rc = ZKP_HOST_THREADS_OK;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return pending; });
            pending = false;
            running = true;
            std::function<int()> f = std::move(job);
            lk.unlock();
            int r;
            try {
                r = f();
            } catch (...) {
                r = ZKP_HOST_THREADS_E_DEVICE;
            }
            lk.lock();
            rc = r;
            running = false;
            cv_done.notify_all();
        }
    }

public:
    void submit(std::function<int()> f) {
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            started = true;
            std::thread([this] { worker(); }).detach();
        }
        job = std::move(f);
        pending = true;
        cv.notify_one();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return !pending && !running; });
        return rc;
    }
};
Uploader& uploader(int slot) {  // one per device slot (the chunk MSMs of sharded bases run concurrently, one caller thread per slot)
    static std::mutex mu;
    static std::vector<Uploader*> all;  // intentionally leaked, as the pool
    std::lock_guard<std::mutex> lk(mu);
    if (slot < 0) slot = 0;
    while (all.size() <= (size_t)slot) all.push_back(new Uploader);
    return *all[(size_t)slot];
}


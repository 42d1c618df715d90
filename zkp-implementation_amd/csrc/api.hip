// api.hip -- host side of libzkp_hip.so: the C ABI of include/zkp_hip.h over the gfx950 kernels in
// ntt.hpp / msm.hpp.  No CPU implementation of an MSM or NTT exists here: every compute entry launches HIP
// kernels and fails with ZKP_E_DEVICE when no gfx950 device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zkp_hip.h"
#include "host_ff.hpp"
#include "kzg_host.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "plonk.hpp"
#include "fri.hpp"
#include "transcript_host.hpp"
#include "pairing_host.hpp"

using namespace zkp;
using namespace zkp::host;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// Every int-returning entry is a function-try-block: a C, Go or Rust caller must never see a C++ exception unwind through
// the boundary (include/zkp_hip.h: "never aborts or unwinds").  Host containers and std::thread are the only sources.
int on_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try { g_err = "host allocation failed"; } catch (...) {}
        return ZKP_E_NOMEM;
    } catch (const std::exception& e) {
        try { g_err = std::string("internal error: ") + e.what(); } catch (...) {}
        return ZKP_E_DEVICE;
    } catch (...) {
        try { g_err = "internal error: unknown exception"; } catch (...) {}
        return ZKP_E_DEVICE;
    }
}
#define ZKP_CATCH_INT catch (...) { return on_exception(); }

#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? ZKP_E_NOMEM : ZKP_E_DEVICE,                             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
    } while (0)
#define ZCHK(expr)             \
    do {                       \
        int r_ = (expr);       \
        if (r_ != ZKP_OK) return r_; \
    } while (0)

#include "host_threads.hpp"  // HostPool (host_pool()), Uploader (uploader(slot))
static_assert(ZKP_HOST_THREADS_E_DEVICE == ZKP_E_DEVICE && ZKP_HOST_THREADS_OK == ZKP_OK, "host_threads.hpp error codes");

// ----------------------------------------------------------------------------------------------------
// One resident host thread per device slot for the multi-device entries (zkp_init_devices): job i of a batch runs on thread i,
// enters its slot's context there and launches on that slot's stream, so the per-device pieces of one call (the chunk MSMs
// over sharded bases, the per-chunk SRS expansion) run concurrently.  Threads are created on first use and joined by
// zkp_shutdown.  A job reports through its own (rc, message) pair: the thread-local error string of a worker is not the
// caller's.
// ----------------------------------------------------------------------------------------------------
class DeviceWorkers {
    struct W {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        const std::function<void()>* job = nullptr;
        bool busy = false, quit = false;
    };
    std::mutex run_mu;
    std::vector<W*> ws;
    static void loop(W* w) {
        std::unique_lock<std::mutex> lk(w->mu);
        for (;;) {
            w->cv.wait(lk, [&] { return w->job != nullptr || w->quit; });
            if (w->quit) return;
            const std::function<void()>* j = w->job;
            lk.unlock();
            (*j)();
            lk.lock();
            w->job = nullptr;
            w->busy = false;
            w->cv.notify_all();
        }
    }

public:
    void run(const std::vector<std::function<void()>>& jobs) {
        std::lock_guard<std::mutex> one(run_mu);
        while (ws.size() < jobs.size()) {
            W* w = new W;
            w->th = std::thread(loop, w);
            ws.push_back(w);
        }
        for (size_t i = 0; i < jobs.size(); i++) {
            std::lock_guard<std::mutex> lk(ws[i]->mu);
            ws[i]->job = &jobs[i];
            ws[i]->busy = true;
            ws[i]->cv.notify_all();
        }
        for (size_t i = 0; i < jobs.size(); i++) {
            std::unique_lock<std::mutex> lk(ws[i]->mu);
            ws[i]->cv.wait(lk, [&] { return !ws[i]->busy; });
        }
    }
    void stop() {
        std::lock_guard<std::mutex> one(run_mu);
        for (W* w : ws) {
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->quit = true;
                w->cv.notify_all();
            }
            w->th.join();
            delete w;
        }
        ws.clear();
    }
};
DeviceWorkers g_workers;
void device_workers_stop() { g_workers.stop(); }

// ----------------------------------------------------------------------------------------------------
// optional per-phase timing: HIP events on the launch stream (zkp_profile_* in include/zkp_hip.h)
// ----------------------------------------------------------------------------------------------------
struct ProfRec {
    const char* name;
    hipEvent_t a, b;  // device phases
    double host_ms;   // host phases (a == nullptr)
    bool owns_a;      // false: `a` is the end event of the previous record (chained scope)
};
std::atomic<int> g_prof_on{0};  // 0 off, 1 every phase, 2 the dominant kernel only (zkp_profile_enable)
struct Ctx;
Ctx& ctx();
std::vector<ProfRec>& prof_records();
hipStream_t& prof_last_stream();

struct ProfScope {
    ProfRec rec;
    hipStream_t st;
    bool on;
    // chain = true: this phase starts where the previous recorded scope on the same stream ended and NOTHING was enqueued in
    // between, so its start is that scope's end event -- one marker per phase boundary instead of two (each marker is a
    // ~5 us bubble on the stream, inside the region bench.py times)
    ProfScope(const char* name, hipStream_t s, bool chain = false) : st(s), on(false) {
        const int level = g_prof_on.load(std::memory_order_relaxed);
        on = level == 1 || (level == 2 && std::strcmp(name, "msm_accumulate") == 0);
        if (!on) return;
        if (level == 2) chain = false;  // the scope before it was not recorded
        rec.name = name;
        rec.host_ms = 0;
        rec.owns_a = true;
        std::vector<ProfRec>& recs = prof_records();
        if (chain && !recs.empty() && recs.back().b && prof_last_stream() == s) {
            rec.a = recs.back().b;
            rec.owns_a = false;
            if (hipEventCreate(&rec.b) != hipSuccess) on = false;
            return;
        }
        if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(rec.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(rec.b, st);
        prof_records().push_back(rec);
        prof_last_stream() = st;
    }
};
void prof_host(const char* name, double ms) {
    if (g_prof_on.load(std::memory_order_relaxed) != 1) return;
    ProfRec r;
    r.name = name;
    r.a = nullptr;
    r.b = nullptr;
    r.owns_a = false;
    r.host_ms = ms;
    prof_records().push_back(r);
}

struct DevBuf {  // grow-only device allocation
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return ZKP_OK;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(&p, bytes));
        cap = bytes;
        return ZKP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// ----------------------------------------------------------------------------------------------------
// host <-> device field views
// ----------------------------------------------------------------------------------------------------
template <class F> struct HostField;
template <> struct HostField<Fr> {
    typedef HFr H;
    static Fr dev(const HFr& x) { Fr r; std::memcpy(r.l, x.l, 32); return r; }  // Montgomery on both sides
    // twiddle form of the kernels (fr29.hpp): w * 2^261 mod r sliced into 29-bit limbs
    static Fr29 tw(const HFr& x) {
        HFr y = x;
        for (int i = 0; i < 5; i++) y = y.dbl();
        Fr29 r;
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
            unsigned __int128 v = y.l[w];
            if (w + 1 < 4) v |= (unsigned __int128)y.l[w + 1] << 64;
            r.l[i] = (uint32_t)(v >> sh) & MASK29;
        }
        return r;
    }
    static HFr root(unsigned log_n) { return fr_root_of_unity(log_n); }
    static constexpr int ID = 0;
};
template <> struct HostField<Gl> {
    typedef HGl H;
    static Gl dev(const HGl& x) { return Gl{x.from_mont().l[0]}; }  // device twiddles are canonical (ff.hpp)
    static Gl tw(const HGl& x) { return dev(x); }
    static HGl root(unsigned log_n) { return gl_root_of_unity(log_n); }
    static constexpr int ID = 1;
};

template <class F>
struct NttPlan {
    unsigned log_n = 0;
    int passes = 0;
    int r[4] = {0, 0, 0, 0};
    const typename NttOps<F>::W* tw[4] = {nullptr, nullptr, nullptr, nullptr};
    typename NttOps<F>::W* inter_lo = nullptr;
    typename NttOps<F>::W* inter_hi = nullptr;
    // inverse plans: inter_lo times 1/n.  Pass 0 multiplies every element by one inter-pass twiddle anyway, so reading it from
    // this table applies the 1/n of the inverse transform for free (no scaling product at the store of the last pass)
    typename NttOps<F>::W* inter_lo_ninv = nullptr;
    uint32_t h = 0;
    // passes 1 .. P-2 work on sub-problems of size M_p = n >> (r_0 + .. + r_{p-1}); up to 2^17 their inter-pass twiddles
    // omega_{M_p}^e come from a direct table (one load, no product of a low and a high factor)
    typename NttOps<F>::W* direct[4] = {nullptr, nullptr, nullptr, nullptr};
    // pass 0's inter-pass twiddles as a matrix shaped like the data (NttOps::PASS0_MATRIX); [1] = times 1/n.  Built on first use.
    F* tw_matrix[2] = {nullptr, nullptr};
    typename HostField<F>::H n_inv;
};

template <class F>
struct CosetCache {
    bool valid = false;
    unsigned log_n = 0;
    int inverse = 0;
    uint64_t key[4] = {0, 0, 0, 0};
    uint64_t ckey[4] = {0, 0, 0, 0};  // the constant factor c of the table c * g^e
    typename NttOps<F>::W* lo = nullptr;
    typename NttOps<F>::W* hi = nullptr;
    size_t lo_cap = 0, hi_cap = 0;
    uint32_t h = 0;
};

// One context per device SLOT.  zkp_init(device) makes a single slot; zkp_init_devices() one per listed HIP device (the same
// device may be listed more than once: two slots on one GPU have separate workspaces and streams, which is how a 1-GPU box
// rehearses the multi-device entries).  Every entry runs inside ONE context, chosen by the handle it is given (bases, prover) or by
// the calling thread's zkp_set_device(); the context's mutex serialises the entries of that slot only -- entries on different
// slots run concurrently.
struct Ctx {
    int slot = 0;
    int device = -1;
    std::mutex mu;
    hipStream_t stream = nullptr;  // non-blocking; the per-slot workers of the multi-device entries launch on it
    // Workspaces and cached tables are shared by every call on this slot, whatever stream the caller passes.  The `*_dev`
    // entries return without synchronising, so a later call on ANOTHER stream must not touch them before the earlier work
    // is done: each entry records ws_event on its stream when it has enqueued everything, and an entry that arrives with a
    // different stream makes it wait for that event first (WsOrder).
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    std::vector<ProfRec> prof;
    hipStream_t prof_last = nullptr;
    // NTT
    std::map<std::pair<int, int>, void*> radix_tw[2];  // [field] (log_r, inverse) -> table
    std::map<std::pair<unsigned, int>, NttPlan<Fr>> plans_fr;
    std::map<std::pair<unsigned, int>, NttPlan<Gl>> plans_gl;
    // a few (size, direction, coset) tables per field: a PLONK proof alternates forward and inverse transforms on the 4n
    // coset, and a single entry was rebuilt four times per proof (115 us of pow_table launches)
    static constexpr int COSET_WAYS = 8;
    CosetCache<Fr> coset_fr[COSET_WAYS];
    CosetCache<Gl> coset_gl[COSET_WAYS];
    unsigned coset_victim[2] = {0, 0};
    DevBuf ntt_scratch;
    std::map<std::pair<unsigned, int>, void*> axis0_tw[2];  // [field] (log_len, inverse) -> omega_len^e, e < len (run_ntt_axis0)
    // MSM
    DevBuf scalars, digits, sorted, entries, counts, start, perm, over, pieces, buckets, parts, pyr1, odd0, odd1, result;
    void* host_result = nullptr;  // pinned
    size_t host_result_cap = 0;
    void* fri_small = nullptr;    // pinned: the few dozen words zkp_fri_prove reads back after the folding phase
    size_t fri_small_cap = 0;
    DevBuf fb_table;              // fixed-base table (32 x 255 affine points)
    bool fb_ready = false;
    DevBuf tmp;                   // staging for host-pointer entry points
    hipStream_t copy_stream = nullptr;  // zkp_msm_g1: upload of the next scalar range
    hipEvent_t copy_event = nullptr;
    std::vector<hipEvent_t> copy_events;  // one per scalar range of a host-fed MSM beyond the first (created on demand)
    hipStream_t sort_stream = nullptr;  // shared-bucket MSM in several scalar ranges: digits + sort of range r+1 under accumulate r
    hipEvent_t ev_sort[2] = {nullptr, nullptr}, ev_acc[2] = {nullptr, nullptr}, ev_begin = nullptr;
    DevBuf fri_arena, fri_meta;   // zkp_fri_prove: layers (evaluations + Merkle nodes) and the gather descriptors
    DevBuf clk;                   // in-kernel clock stamps (ClkRec per instrumented kernel family, msm.hpp), zkp_profile_clock_read
    // in-process multi-GPU transform (ntt_sharded.inc): two exchange buffers of one slab each, the stream the peer copies run on
    // (under the transforms of the launch stream) and the events that order both against the other slots
    uint32_t tail_max_waves = 2048;  // co-residency bound of msm_pyramid_tail_kernel on this device (create_slot_locked)
    DevBuf xchg_a, xchg_b;
    hipStream_t xstream = nullptr;
    std::vector<hipEvent_t> xev;
    bool peers_enabled = false;
};
enum { CLK_MSM_ACCUMULATE = 0, CLK_MAD_PROBE = 1, CLK_NTT_FR = 2, CLK_NTT_GL = 3, CLK_COUNT = 4 };
static const char* const kClkNames[CLK_COUNT] = {"msm_accumulate", "mad_probe", "ntt_fr_pass", "ntt_gl_pass"};

struct Runtime {
    std::mutex mu;              // guards `slots` (creation / shutdown); never held while a context works
    std::vector<Ctx*> slots;
    bool multi = false;         // zkp_init_devices() with more than one slot
};
Runtime g_rt;

// The record the stamps of kernel family `which` go to while profiling is on (nullptr otherwise: the kernels then execute no stamp)
ClkRec* clk_record(int which) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return nullptr;
    Ctx& c = ctx();
    if (!c.clk.p) {
        if (c.clk.ensure(sizeof(ClkRec) * CLK_COUNT) != ZKP_OK) return nullptr;
        if (hipMemset(c.clk.p, 0, sizeof(ClkRec) * CLK_COUNT) != hipSuccess) return nullptr;
    }
    return reinterpret_cast<ClkRec*>(c.clk.p) + which;
}
thread_local int t_slot = -1;       // zkp_set_device(): the slot of handle-less entries on this thread; -1 = default (slot 0, and
                                    // zkp_g1_bases_create shards over ALL slots)
thread_local Ctx* t_cur = nullptr;  // the context of the entry this thread is inside
Ctx& ctx() { return *t_cur; }
std::vector<ProfRec>& prof_records() { return ctx().prof; }
hipStream_t& prof_last_stream() { return ctx().prof_last; }

int create_slot_locked(int device);  // below zkp_init

// Resolve the slot of an entry (`slot` < 0: the thread's zkp_set_device() choice, slot 0 by default; no runtime yet: one slot
// on the current HIP device, as zkp_init(-1)), enter its context and make its device current.
struct CtxScope {
    Ctx* prev;
    std::unique_lock<std::mutex> lk;
    int rc = ZKP_OK;
    int restore_device = -1;  // the caller's current HIP device when it differs from the slot's: put back on exit (a caller such as
                              // torch keeps its own idea of the current device)
    explicit CtxScope(int slot) : prev(t_cur) {
        Ctx* c = nullptr;
        {
            std::lock_guard<std::mutex> g(g_rt.mu);
            if (g_rt.slots.empty()) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess) dev = 0;
                rc = create_slot_locked(dev);
                if (rc != ZKP_OK) return;
            }
            const int s = slot >= 0 ? slot : (t_slot >= 0 ? t_slot : 0);
            if (s >= (int)g_rt.slots.size()) {
                rc = fail(ZKP_E_ARG, "device slot out of range (zkp_init_devices / zkp_set_device)");
                return;
            }
            c = g_rt.slots[(size_t)s];
        }
        lk = std::unique_lock<std::mutex>(c->mu);
        t_cur = c;
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != c->device) restore_device = cur;
        if (cur != c->device && hipSetDevice(c->device) != hipSuccess) rc = fail(ZKP_E_DEVICE, "hipSetDevice failed");
    }
    ~CtxScope() {
        t_cur = prev;
        if (restore_device >= 0 && !prev) (void)hipSetDevice(restore_device);  // (a nested scope leaves the outer one's device alone)
    }
};
#define CTX_ENTER(slot)           \
    CtxScope ctx_scope_(slot);    \
    if (ctx_scope_.rc != ZKP_OK) return ctx_scope_.rc

// See Ctx::ws_event.  Constructed by an entry after CTX_ENTER with the stream it is about to launch on.
struct WsOrder {
    hipStream_t st;
    explicit WsOrder(hipStream_t s) : st(s) {
        Ctx& c = ctx();
        if (c.ws_pending && c.ws_stream != st) (void)hipStreamWaitEvent(st, c.ws_event, 0);
    }
    ~WsOrder() {
        Ctx& c = ctx();
        if (!c.ws_event && hipEventCreateWithFlags(&c.ws_event, hipEventDisableTiming) != hipSuccess) {
            c.ws_event = nullptr;
            (void)hipStreamSynchronize(st);  // no event to order later callers with: be done before returning
            c.ws_pending = false;
            return;
        }
        (void)hipEventRecord(c.ws_event, st);
        c.ws_stream = st;
        c.ws_pending = true;
    }
};

template <class K>
int allow_big_lds(K kernel) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    return ZKP_OK;
}

// ----------------------------------------------------------------------------------------------------
// NTT driver
// ----------------------------------------------------------------------------------------------------
template <class F>
std::map<std::pair<unsigned, int>, NttPlan<F>>& plan_map();
template <> std::map<std::pair<unsigned, int>, NttPlan<Fr>>& plan_map<Fr>() { return ctx().plans_fr; }
template <> std::map<std::pair<unsigned, int>, NttPlan<Gl>>& plan_map<Gl>() { return ctx().plans_gl; }
template <class F> CosetCache<F>* coset_cache();
template <> CosetCache<Fr>* coset_cache<Fr>() { return ctx().coset_fr; }
template <> CosetCache<Gl>* coset_cache<Gl>() { return ctx().coset_gl; }

template <class F>
int make_pow_table(const typename HostField<F>::H& base, const typename HostField<F>::H& c, uint32_t shift,
                   uint32_t count, typename NttOps<F>::W* out, hipStream_t st) {
    hipLaunchKernelGGL(pow_table_kernel<F>, dim3((count + 255) / 256), dim3(256), 0, st, HostField<F>::dev(base),
                       HostField<F>::dev(c), shift, count, out);
    HIPCHK(hipGetLastError());
    return ZKP_OK;
}

template <class F>
int get_radix_table(int log_r, int inverse, const typename NttOps<F>::W** out, hipStream_t st) {
    typedef typename HostField<F>::H H;
    typedef typename NttOps<F>::W W;
    auto& m = ctx().radix_tw[HostField<F>::ID];
    auto key = std::make_pair(log_r, inverse);
    auto it = m.find(key);
    if (it == m.end()) {
        uint32_t count = log_r ? (1u << (log_r - 1)) : 1u;
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, sizeof(W) * count));
        H w = HostField<F>::root((unsigned)log_r);
        if (inverse) w = w.inverse();
        ZCHK(make_pow_table<F>(w, H::one(), 0, count, reinterpret_cast<W*>(p), st));
        HIPCHK(hipStreamSynchronize(st));  // tables are shared by later calls on any stream
        it = m.emplace(key, p).first;
    }
    *out = reinterpret_cast<const W*>(it->second);
    return ZKP_OK;
}

// largest transform whose pass-0 twiddles are kept as a matrix (32 B per element per direction: 512 MiB at 2^24).  Measured with and
// without on one box (profiles/r02_m_ntt_twiddle_matrix.md): 2^18 -2 %, 2^21..2^23 -4 %, 2^24 -2.7 %, fifteen 2^18 -4.5 %; 2^25 and
// 2^26 gain 1 % for 1 and 2 GiB per direction, which is not worth the memory
static unsigned pass0_matrix_max_log() {
    static const unsigned v = [] {
        const char* e = getenv("ZKP_NTT_TW_MATRIX_MAX_LOG");
        return e ? (unsigned)std::min(30, std::max(0, atoi(e))) : 24u;
    }();
    return v;
}

template <class F>
int get_plan(unsigned log_n, int inverse, bool allow_wide, NttPlan<F>** out, hipStream_t st) {
    typedef typename HostField<F>::H H;
    typedef typename NttOps<F>::W W;
    auto& m = plan_map<F>();
    auto key = std::make_pair(log_n, inverse | (allow_wide ? 2 : 0));
    auto it = m.find(key);
    if (it == m.end()) {
        NttPlan<F> pl;
        pl.log_n = log_n;
        pl.passes = (int)log_n <= NttOps<F>::MAX_TILE_LOG ? 1 : (int)((log_n + NttOps<F>::MAX_PASS_LOG - 1) / NttOps<F>::MAX_PASS_LOG);
        // wide (radix-2^9, two-column) passes where they save a whole pass: 2^25 5.84 -> 4.42 ms, 2^26 11.47 -> 9.09 ms, 2^27 22.6 -> 19.3 ms,
        // fifteen 2^18 transforms (PLONK round 3) 2.11 -> 1.98 ms (profiles/r02_l_ntt_wide_pass.md).  Not for a lone small transform:
        // 2^17 would be 128 tiles on 256 CUs (0.047 against 0.042 ms): the caller allows it from 2^19 elements per launch.
        if (pl.passes > 1 && allow_wide && !getenv("ZKP_NTT_NO_WIDE_PASS")) {
            for (int maxr = NttOps<F>::MAX_PASS_LOG + 1; maxr <= NttOps<F>::WIDE_PASS_LOG; maxr++) {  // the narrowest radix that saves a pass
                // radix 2^10 (single-column tiles, 32-byte runs) only while the data is cache-resident: 2^19 0.102 -> 0.092 ms, 2^20
                // 0.173 -> 0.165 ms, but 2^28 45.1 -> 48.3 ms (profiles/r02_l_ntt_wide_pass.md)
                if ((int)log_n > NttOps<F>::wide_max_log_n(maxr)) break;
                const int wide = (int)((log_n + maxr - 1) / maxr);
                if (wide < pl.passes) pl.passes = wide;
            }
        }
        int base = (int)log_n / pl.passes, rem = (int)log_n % pl.passes;
        for (int p = 0; p < pl.passes; p++) pl.r[p] = base + (p < rem ? 1 : 0);
        for (int p = 0; p < pl.passes; p++) ZCHK(get_radix_table<F>(pl.r[p], inverse, &pl.tw[p], st));
        H w = HostField<F>::root(log_n);
        if (inverse) w = w.inverse();
        H two = H::from_u64(2), ninv = H::one(), half = two.inverse();
        for (unsigned i = 0; i < log_n; i++) ninv = ninv * half;
        pl.n_inv = ninv;
        if (pl.passes > 1) {
            pl.h = (log_n + 1) / 2;
            uint32_t nlo = 1u << pl.h, nhi = 1u << (log_n - pl.h);
            // a plan that fails half-way is not cached: give back what it had allocated (the kernels filling the tables may
            // still be queued on st, so drain it first)
            auto build = [&]() -> int {
                HIPCHK(hipMalloc(reinterpret_cast<void**>(&pl.inter_lo), sizeof(W) * nlo));
                HIPCHK(hipMalloc(reinterpret_cast<void**>(&pl.inter_hi), sizeof(W) * nhi));
                ZCHK(make_pow_table<F>(w, H::one(), 0, nlo, pl.inter_lo, st));
                ZCHK(make_pow_table<F>(w, H::one(), pl.h, nhi, pl.inter_hi, st));
                if (inverse) {
                    HIPCHK(hipMalloc(reinterpret_cast<void**>(&pl.inter_lo_ninv), sizeof(W) * nlo));
                    ZCHK(make_pow_table<F>(w, ninv, 0, nlo, pl.inter_lo_ninv, st));
                }
                unsigned outer = pl.r[0];
                for (int p = 1; p + 1 < pl.passes; p++) {
                    const unsigned log_m = log_n - outer;
                    if (log_m <= 17) {
                        HIPCHK(hipMalloc(reinterpret_cast<void**>(&pl.direct[p]), sizeof(W) << log_m));
                        ZCHK(make_pow_table<F>(w, H::one(), outer, 1u << log_m, pl.direct[p], st));  // w^(e << outer) = omega_M^e
                    }
                    outer += pl.r[p];
                }
                HIPCHK(hipStreamSynchronize(st));
                return ZKP_OK;
            };
            const int rc = build();
            if (rc != ZKP_OK) {
                (void)hipStreamSynchronize(st);
                (void)hipFree(pl.inter_lo);
                (void)hipFree(pl.inter_hi);
                (void)hipFree(pl.inter_lo_ninv);
                for (int p = 0; p < 4; p++) (void)hipFree(pl.direct[p]);
                (void)hipGetLastError();
                return rc;
            }
        }
        it = m.emplace(key, pl).first;
    }
    *out = &it->second;
    return ZKP_OK;
}

// two-level table of c * g^idx, idx < 2^log_n
template <class F>
int get_coset_tables(unsigned log_n, int inverse, const uint64_t* coset, const typename HostField<F>::H& c,
                     PowTab<F>* out, hipStream_t st) {
    typedef typename HostField<F>::H H;
    typedef typename NttOps<F>::W W;
    constexpr int NL = sizeof(H) / 8;
    CosetCache<F>* ways = coset_cache<F>();
    uint64_t key[4] = {0, 0, 0, 0}, ckey[4] = {0, 0, 0, 0};
    std::memcpy(key, coset, 8 * NL);
    std::memcpy(ckey, c.l, 8 * NL);
    int way = -1;
    for (int i = 0; i < Ctx::COSET_WAYS && way < 0; i++)
        if (ways[i].valid && ways[i].log_n == log_n && ways[i].inverse == inverse && std::memcmp(ways[i].key, key, sizeof key) == 0 &&
            std::memcmp(ways[i].ckey, ckey, sizeof ckey) == 0)
            way = i;
    const bool hit = way >= 0;
    if (!hit) {  // an unused entry, else round-robin (tables still in use by enqueued kernels are rewritten in stream order)
        for (int i = 0; i < Ctx::COSET_WAYS && way < 0; i++)
            if (!ways[i].valid) way = i;
        if (way < 0) way = (int)(ctx().coset_victim[HostField<F>::ID]++ % Ctx::COSET_WAYS);
    }
    CosetCache<F>& cc = ways[way];
    if (!hit) {
        cc.valid = false;
        uint32_t h = (log_n + 1) / 2;
        uint32_t nlo = 1u << h, nhi = 1u << (log_n - h);
        // grow-only tables: the entry forgets a table BEFORE releasing it, so that a failing hipFree / hipMalloc leaves an empty
        // (invalid, capacity 0) entry behind and never a dangling pointer with a stale capacity
        auto regrow = [&](W*& tab, size_t& cap, uint32_t want) -> int {
            if (cap >= want) return ZKP_OK;
            W* old = tab;
            tab = nullptr;
            cap = 0;
            if (old) HIPCHK(hipFree(old));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&tab), sizeof(W) * want));
            cap = want;
            return ZKP_OK;
        };
        ZCHK(regrow(cc.lo, cc.lo_cap, nlo));
        ZCHK(regrow(cc.hi, cc.hi_cap, nhi));
        H g = H::load(coset);
        if (inverse) g = g.inverse();
        ZCHK(make_pow_table<F>(g, c, 0, nlo, cc.lo, st));
        ZCHK(make_pow_table<F>(g, H::one(), h, nhi, cc.hi, st));
        cc.h = h;
        cc.log_n = log_n;
        cc.inverse = inverse;
        std::memcpy(cc.key, key, sizeof key);
        std::memcpy(cc.ckey, ckey, sizeof ckey);
        cc.valid = true;
    }
    out->lo = cc.lo;
    out->hi = cc.hi;
    out->h = cc.h;
    return ZKP_OK;
}

template <class F>
ScaleSpec<F> no_scale() {
    ScaleSpec<F> s;
    std::memset(&s, 0, sizeof s);
    s.mode = SCALE_NONE;
    return s;
}

// Optional extras of a batched transform (the local pieces of the multi-GPU four-step NTT, zkp_hip/dist.py)
struct NttIo {
    const NttRemap* in_remap = nullptr;   // gathered input: logical element e of transform b at ntt_phys(...)
    const NttRemap* out_remap = nullptr;  // scattered output (same mapping on the natural output index)
    unsigned tw_log_n = 0;                // != 0: output k of transform b is multiplied by omega_{2^tw_log_n}^(+-(tw_row0 + b) k)
    uint64_t tw_row0 = 0;
};

template <class F>
int run_ntt(const F* d_in, F* d_data, unsigned log_n, size_t batch, int inverse, const uint64_t* coset, hipStream_t st,
            const NttIo* io = nullptr) {
    typedef typename HostField<F>::H H;
    if (log_n > 32) return fail(ZKP_E_ARG, "log_n > 32 (two-adicity of the field)");
    if (batch == 0 || log_n == 0) return ZKP_OK;  // size-1 transform is the identity (n^-1 = coset^0 = 1)
    if (batch > 65535) return fail(ZKP_E_ARG, "batch > 65535");
    inverse = inverse ? 1 : 0;
    NttPlan<F>* pl = nullptr;
    ZCHK(get_plan<F>(log_n, inverse, ((uint64_t)batch << log_n) >= (1ull << 19), &pl, st));
    const uint64_t n = 1ull << log_n;
    ScaleSpec<F> pre = no_scale<F>(), post = no_scale<F>();
    const bool four_step_tw = io && io->tw_log_n != 0;
    if (four_step_tw) {
        if (coset) return fail(ZKP_E_ARG, "a coset and a four-step twiddle cannot be combined");
        if (io->tw_log_n > 32 || ((io->tw_row0 + batch - 1) * (n - 1)) >> io->tw_log_n)
            return fail(ZKP_E_ARG, "four-step twiddle exponent (row0 + batch - 1) * (n - 1) must stay below 2^tw_log_n");
        post.mode = SCALE_POW_ROW;
        post.row0 = io->tw_row0;
        const H w = HostField<F>::root(io->tw_log_n);  // get_coset_tables inverts the base itself when inverse != 0
        ZCHK(get_coset_tables<F>(io->tw_log_n, inverse, w.l, inverse ? pl->n_inv : H::one(), &post.t, st));  // 1/n rides along
    } else if (coset) {
        if (!inverse) {
            pre.mode = SCALE_POW;
            ZCHK(get_coset_tables<F>(log_n, 0, coset, H::one(), &pre.t, st));
        } else {
            post.mode = SCALE_POW;
            ZCHK(get_coset_tables<F>(log_n, 1, coset, pl->n_inv, &post.t, st));
        }
    } else if (inverse && pl->passes == 1) {
        post.mode = SCALE_CONST;
        post.c = HostField<F>::tw(pl->n_inv);
    }
    const bool ninv_in_pass0 = inverse && !coset && !four_step_tw && pl->passes > 1;  // 1/n rides on pass 0's inter-pass twiddles
    constexpr int LOG_T = NttOps<F>::LOG_T;
    typedef typename NttOps<F>::E E;
    typedef typename NttOps<F>::W W;
    const int P = pl->passes;
    const F* cur_in = d_in;
    F* work = d_data;
    NttRemap no_remap;
    std::memset(&no_remap, 0, sizeof no_remap);
    if (P > 1) {
        ZCHK(ctx().ntt_scratch.ensure(sizeof(F) * n * batch));
        work = reinterpret_cast<F*>(ctx().ntt_scratch.p);
    }
    unsigned log_outer = 0;
    for (int p = 0; p + 1 < P; p++) {
        NttStridedParams<F> sp;
        std::memset(&sp, 0, sizeof sp);
        sp.in = cur_in;
        sp.out = work;
        sp.tw = pl->tw[p];
        sp.n = n;
        sp.inner = n >> (log_outer + pl->r[p]);
        sp.log_r = pl->r[p];
        if (pl->direct[p]) {
            sp.tw_stride_log = 0;
            sp.inter.lo = pl->direct[p];
            sp.inter.hi = pl->direct[p];  // never read: every exponent is below 2^h
            sp.inter.h = log_n - log_outer;
        } else {
            sp.tw_stride_log = log_outer;
            sp.inter.lo = (p == 0 && ninv_in_pass0) ? pl->inter_lo_ninv : pl->inter_lo;
            sp.inter.hi = pl->inter_hi;
            sp.inter.h = pl->h;
        }
        if (p == 0 && NttOps<F>::PASS0_MATRIX && log_n <= pass0_matrix_max_log()) {
            F*& mat = pl->tw_matrix[ninv_in_pass0 ? 1 : 0];
            if (!mat) {  // (the two-level tables set above are what the matrix is made from)
                // An optimisation, 32 B per element held until zkp_shutdown (512 MiB per direction at 2^24): when the device has
                // no room for it the transform keeps the two-level tables (one more product per element) instead of failing
                // The plan is cached: the matrix is published in it only once it is filled -- a failed fill must not leave a
                // non-null table of garbage behind for every later transform of this size
                F* fresh = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&fresh), sizeof(F) * n) != hipSuccess) {
                    (void)hipGetLastError();
                } else {
                    hipLaunchKernelGGL(twiddle_matrix_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sp.inter, (uint64_t)n,
                                       (uint64_t)sp.inner, fresh);
                    hipError_t fe = hipGetLastError();
                    if (fe == hipSuccess) fe = hipStreamSynchronize(st);  // shared by later calls on any stream
                    if (fe != hipSuccess) {
                        (void)hipFree(fresh);
                        return fail(ZKP_E_DEVICE, std::string("twiddle matrix: ") + hipGetErrorString(fe));
                    }
                    mat = fresh;
                }
            }
            sp.tw_matrix = mat;
        }
        sp.pre = p == 0 ? pre : no_scale<F>();
        sp.clk = clk_record(HostField<F>::ID == 0 ? CLK_NTT_FR : CLK_NTT_GL);
        sp.remap = (p == 0 && io && io->in_remap) ? *io->in_remap : no_remap;
        const size_t R = 1ull << pl->r[p];
        const int log_t = NttOps<F>::log_t_of(pl->r[p]);
        const size_t lds = sizeof(E) * (R << log_t) + sizeof(W) * (R / 2);
        const uint64_t tiles = (n >> pl->r[p]) >> log_t;
        {
            ProfScope ps(HostField<F>::ID == 0 ? "ntt_fr_pass" : "ntt_gl_pass", st, p > 0);  // passes of one transform are adjacent
            const dim3 grid((unsigned)tiles, (unsigned)batch), block(NttOps<F>::THREADS);
            constexpr int T0 = NttOps<F>::LOG_T;  // tile widths: T0 (radix <= MAX_PASS_LOG) and, where the field has wide passes, T0 - 1, T0 - 2
            constexpr bool WIDE = NttOps<F>::WIDE_PASS_LOG > NttOps<F>::MAX_PASS_LOG;
            if (!WIDE || log_t == T0)
                hipLaunchKernelGGL((ntt_pass_strided<F, T0>), grid, block, lds, st, sp);
            else if (log_t == T0 - 1)
                hipLaunchKernelGGL((ntt_pass_strided<F, (WIDE ? T0 - 1 : T0)>), grid, block, lds, st, sp);
            else
                hipLaunchKernelGGL((ntt_pass_strided<F, (WIDE ? T0 - 2 : T0)>), grid, block, lds, st, sp);
        }
        HIPCHK(hipGetLastError());
        cur_in = work;
        log_outer += pl->r[p];
    }
    NttLastParams<F> lp;
    std::memset(&lp, 0, sizeof lp);
    lp.in = cur_in;
    lp.out = d_data;
    lp.tw = pl->tw[P - 1];
    lp.n = n;
    lp.log_r = pl->r[P - 1];
    lp.log_r0 = P > 1 ? pl->r[0] : 0;
    lp.log_m = 0;
    for (int p = 1; p + 1 < P; p++) lp.log_m += pl->r[p];
    lp.log_r1 = P == 4 ? pl->r[1] : lp.log_m;
    lp.t_log = std::min<uint32_t>((uint32_t)NttOps<F>::log_t_of((int)lp.log_r), lp.log_r0);
    lp.pre = P == 1 ? pre : no_scale<F>();
    lp.post = post;
    lp.remap = (P == 1 && io && io->in_remap) ? *io->in_remap : no_remap;
    lp.out_remap = (io && io->out_remap) ? *io->out_remap : no_remap;
    lp.clk = clk_record(HostField<F>::ID == 0 ? CLK_NTT_FR : CLK_NTT_GL);
    {
        const size_t R = 1ull << lp.log_r, T = 1ull << lp.t_log;
        const size_t stride = T > 1 ? T + NttOps<F>::PAD : 1;
        const size_t lds = sizeof(E) * (R * stride) + sizeof(W) * (R / 2);
        const uint64_t tiles = (1ull << (lp.log_r0 - lp.t_log)) << lp.log_m;
        ProfScope ps(HostField<F>::ID == 0 ? "ntt_fr_pass" : "ntt_gl_pass", st, P > 1);
        hipLaunchKernelGGL(ntt_pass_last<F>, dim3((unsigned)tiles, (unsigned)batch), dim3(NttOps<F>::THREADS), lds, st, lp);
        HIPCHK(hipGetLastError());
    }
    return ZKP_OK;
}

template <class F>
int run_ntt(F* d_data, unsigned log_n, size_t batch, int inverse, const uint64_t* coset, hipStream_t st) {
    return run_ntt<F>(d_data, d_data, log_n, batch, inverse, coset, st, nullptr);
}

// Transforms of length 2^log_len along axis 0 of a row-major matrix [2^log_len][cols] (the columns are the contiguous
// direction), natural order in and out, every output (k, b) multiplied by omega_{2^tw_log_n}^(+-(col0 + b) k) when
// tw_log_n != 0 and by 1/2^log_len when inverse.  One or two strided passes (ntt_pass_strided): the second one stores the rows
// in natural order and applies the twiddle, so the matrix is read and written exactly once per pass and never transposed.
// This is the column half of the multi-GPU four-step transform: the all-to-all delivers [all rows][my columns].
template <class F>
int run_ntt_axis0(const F* d_in, F* d_out, unsigned log_len, size_t cols, int inverse, unsigned tw_log_n, uint64_t col0,
                  hipStream_t st) {
    typedef typename HostField<F>::H H;
    typedef typename NttOps<F>::E E;
    typedef typename NttOps<F>::W W;
    constexpr int LOG_T = NttOps<F>::LOG_T;
    constexpr int MAXR = NttOps<F>::MAX_PASS_LOG;
    if (log_len == 0 || log_len > 2 * (unsigned)MAXR) return fail(ZKP_E_ARG, "axis-0 transform length out of range");
    if (cols == 0 || (cols & (cols - 1)) || cols < (1u << LOG_T)) return fail(ZKP_E_ARG, "cols must be a power of two >= 4");
    inverse = inverse ? 1 : 0;
    const uint64_t L = 1ull << log_len, total = L * cols;
    unsigned col_bits = 0;
    while ((1ull << col_bits) < cols) col_bits++;
    if (tw_log_n > 32 || (tw_log_n && (((col0 + cols - 1) * (L - 1)) >> tw_log_n)))
        return fail(ZKP_E_ARG, "four-step twiddle exponent (col0 + cols - 1) * (len - 1) must stay below 2^tw_log_n");
    const int P = log_len <= (unsigned)MAXR ? 1 : 2;
    const int r0 = P == 1 ? (int)log_len : (int)(log_len + 1) / 2, r1 = (int)log_len - r0;
    // final factor table: c * base^e with c = 1/len for the inverse; base = the N-th root (or 1: a constant table)
    H ninv = H::one();
    if (inverse) {
        const H half = H::from_u64(2).inverse();
        for (unsigned i = 0; i < log_len; i++) ninv = ninv * half;
    }
    PowTab<F> fin;
    {
        const H base = tw_log_n ? HostField<F>::root(tw_log_n) : H::one();
        ZCHK(get_coset_tables<F>(tw_log_n ? tw_log_n : log_len, inverse, base.l, ninv, &fin, st));
    }
    NttStridedParams<F> sp;
    std::memset(&sp, 0, sizeof sp);
    sp.n = total;
    sp.pre = no_scale<F>();
    sp.col_bits = col_bits;
    sp.clk = clk_record(HostField<F>::ID == 0 ? CLK_NTT_FR : CLK_NTT_GL);
    auto launch = [&](int log_r) {
        const size_t R = 1ull << log_r;
        const size_t lds = sizeof(E) * (R << LOG_T) + sizeof(W) * (R / 2);
        const uint64_t tiles = (total >> log_r) >> LOG_T;
        ProfScope ps(HostField<F>::ID == 0 ? "ntt_fr_pass" : "ntt_gl_pass", st);
        hipLaunchKernelGGL((ntt_pass_strided<F, NttOps<F>::LOG_T>), dim3((unsigned)tiles, 1), dim3(NttOps<F>::THREADS), lds, st, sp);
    };
    if (P == 1) {
        ZCHK(get_radix_table<F>(r0, inverse, &sp.tw, st));
        sp.in = d_in;
        sp.out = d_out;
        sp.inner = cols;
        sp.log_r = (uint32_t)r0;
        sp.axis0_last = 1;
        sp.tw_on = tw_log_n ? 1u : 0u;
        sp.outer_count = 1;
        sp.col0 = col0;
        sp.inter = fin;
        launch(r0);
    } else {
        // inter-pass twiddles omega_len^(k0 * d1): a direct table of `len` entries per (length, direction)
        auto key = std::make_pair(log_len, inverse);
        auto& cache = ctx().axis0_tw[HostField<F>::ID];
        auto it = cache.find(key);
        if (it == cache.end()) {
            void* p = nullptr;
            HIPCHK(hipMalloc(&p, sizeof(W) << log_len));
            H w = HostField<F>::root(log_len);
            if (inverse) w = w.inverse();
            ZCHK(make_pow_table<F>(w, H::one(), 0, (uint32_t)L, reinterpret_cast<W*>(p), st));
            HIPCHK(hipStreamSynchronize(st));
            it = cache.emplace(key, p).first;
        }
        ZCHK(ctx().ntt_scratch.ensure(sizeof(F) * total));
        F* work = reinterpret_cast<F*>(ctx().ntt_scratch.p);
        ZCHK(get_radix_table<F>(r0, inverse, &sp.tw, st));
        sp.in = d_in;
        sp.out = work;
        sp.inner = (uint64_t)cols << r1;
        sp.log_r = (uint32_t)r0;
        sp.tw_stride_log = 0;
        sp.inter.lo = reinterpret_cast<const W*>(it->second);
        sp.inter.hi = sp.inter.lo;  // never read: every exponent is below 2^h
        sp.inter.h = log_len;
        launch(r0);
        ZCHK(get_radix_table<F>(r1, inverse, &sp.tw, st));
        sp.in = work;
        sp.out = d_out;
        sp.inner = cols;
        sp.log_r = (uint32_t)r1;
        sp.axis0_last = 1;
        sp.tw_on = tw_log_n ? 1u : 0u;
        sp.outer_count = 1ull << r0;
        sp.col0 = col0;
        sp.inter = fin;
        launch(r1);
    }
    HIPCHK(hipGetLastError());
    return ZKP_OK;
}

template <class F>
int ntt_host_entry(uint64_t* data, unsigned log_n, int inverse, const uint64_t* coset) {
    if (!data) return fail(ZKP_E_ARG, "data is null");
    if (log_n > 32) return fail(ZKP_E_ARG, "log_n > 32");
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    const size_t bytes = sizeof(F) << log_n;
    ZCHK(ctx().tmp.ensure(bytes));
    HIPCHK(hipMemcpyAsync(ctx().tmp.p, data, bytes, hipMemcpyHostToDevice, nullptr));
    ZCHK(run_ntt<F>(reinterpret_cast<F*>(ctx().tmp.p), log_n, 1, inverse, coset, nullptr));
    HIPCHK(hipMemcpyAsync(data, ctx().tmp.p, bytes, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return ZKP_OK;
}

// ----------------------------------------------------------------------------------------------------
// MSM driver
// ----------------------------------------------------------------------------------------------------
}  // namespace

struct zkp_bases {
    void* d_xy = nullptr;      // n x 128 B: device-internal affine form (28-bit limbs, fq28.hpp / g1_28.hpp)
    uint8_t* d_inf = nullptr;  // nullable
    size_t n = 0;
    int device = 0;
    int slot = 0;              // device slot that owns d_xy (Ctx::slot)
    // zkp_init_devices with more than one slot: the handle is a CONTAINER (d_xy == nullptr) over per-slot chunk handles,
    // chunk i = points [shard_off[i], shard_off[i] + shards[i]->n) resident on slot shards[i]->slot (contiguous
    // point/scalar chunk per GPU, sharded once at creation)
    std::vector<zkp_bases*> shards;
    std::vector<size_t> shard_off;
    uint32_t pre_c = 0;        // != 0: d_xy holds pre_planes planes of n points, plane s = 2^pre_off[s] * P (shared-bucket MSM);
                               // pre_c = widest slice in bits (2^(pre_c-1) buckets)
    uint32_t pre_planes = 0;
    uint32_t pre_req = 0;      // window_bits the expansion was requested with
    uint16_t pre_off[36] = {0};
};

namespace {

// Window width.  Scalars have 255 bits and the signed-digit recoding needs one spare bit, so only widths dividing 256
// give a top window that is as densely populated as the others (any other width leaves a few-bit top window whose
// handful of buckets collect n / 8 points each: measured 4-9x slower at 2^14..2^18, profiles/r01_window_sweep.txt).
unsigned pick_window_bits(size_t n) {
    int c = n >= 2048 ? 16 : 8;
    if (const char* e = getenv("ZKP_MSM_C")) {
        int v = atoi(e);
        if (v >= 8 && v <= 16) c = v;  // below 8 bits a scalar has more than 32 windows (MsmGeom::off holds 36 offsets)
    }
    return (unsigned)c;
}

// out[m] = sum_i scalars[m][i] * bases[i] as extended-Jacobian points (host), for `count` scalar vectors of the same
// length over the same bases.  The vectors are stacked as extra windows of ONE pass through the kernels, so that a
// batch of small MSMs (the 3 + 1 + 3 + 2 commitments of a PLONK proof) fills the GPU and pays the latency-bound
// bucket reduction once.  With expanded bases (zkp_g1_bases_precompute) all windows of a scalar share one bucket set.
// Host scalars fed range by range: the upload of range k+1 (second stream) overlaps the kernels of range k (shared-bucket
// mode only: later ranges add into the same buckets).  Used by the host-pointer entry zkp_msm_g1.
struct MsmFeed {
    const uint64_t* h_scalars;  // n x 4 limbs on the host
    hipStream_t copy_stream;    // non-blocking
    hipEvent_t ev;
    uint64_t range_log;         // log2 of the scalars per range
    uint64_t first_len;         // != 0: the first range is this short (its upload is the exposed one), the rest follows in one piece
                                // per 2^range_log scalars
    uint64_t second_len = 0;    // != 0 (with first_len): a second range of this length before the rest
};
// bucket lanes (lane-per-bucket kernel: three waves on each of the 1024 SIMDs) / bucket quads below which a bucket's run is split
// (round 5: lanes for TWO generations of workgroups, not one.  With exactly one resident generation the launch lasts as long as its
// longest lanes -- the largest buckets, 1.8x the average run at 32 entries per bucket -- while the SIMDs that drew short runs idle; a
// second generation lets the dispatcher even that out.  PLONK 2^16: the batches of three go from two to four parts per bucket, accumulate +
// fold 1.82 -> 1.76 ms per proof, the proof 3.83 -> 3.75 ms on one box, tools/job_r05m.sh)
static constexpr uint64_t SPLIT_FILL_LANES = 6 * 1024 * 64, SPLIT_FILL_QUADS = 2 * 1024 * 64 / 4 * 2;
static constexpr uint64_t FOLD_LANE_MIN_ADDS = 1ull << 15;  // adds in one fold launch from which one lane per add is used (profiles/r05_m_fold_lane.md)
static constexpr uint32_t MSM_MAX_SPLIT_LOG = 2;  // eight parts measured no better than four (2^16 single 0.535 against 0.529 ms, batches worse)
#ifdef ZKP_MSM_CHECK  // diagnosis builds: wait for every kernel of the walk and say which one was reached
#define MSM_TRACE(stream, what) do { hipError_t e_ = hipStreamSynchronize(stream); fprintf(stderr, "ZKP_MSM_CHECK range %llu: %s done (%d)\n", (unsigned long long)ridx, what, (int)e_); } while (0)
#else
#define MSM_TRACE(stream, what) do { } while (0)
#endif
int msm_partial_batch(const zkp_bases* bases, const Fr* const* d_scalars, size_t count, size_t n, hipStream_t st, HXyzz* out,
                      const MsmFeed* feed = nullptr) {
    if (n > bases->n) return fail(ZKP_E_SIZE, "more scalars than bases (kzg/src/scheme.rs:86)");
    if (count == 0) return ZKP_OK;
    if (n == 0) {
        for (size_t m = 0; m < count; m++) out[m] = HXyzz::infinity();
        return ZKP_OK;
    }
    if (n >= (1ull << 31)) return fail(ZKP_E_ARG, "n >= 2^31");
    if (count > (size_t)MSM_MAX_BATCH) return fail(ZKP_E_ARG, "batch of more than 64 MSMs");
    // expanded bases: always the shared bucket set.  Even a 2^8-term vector over 20-bit windows (2^19 mostly empty buckets)
    // beats the per-window path, whose host tail alone (256 doublings) costs 0.4 ms: 0.35 vs 0.85 ms at 2^10 terms.
    const bool shared = bases->pre_c != 0;
    MsmGeom g;
    g.c = shared ? bases->pre_c : pick_window_bits(n);
    const uint32_t nwin1 = shared ? bases->pre_planes : 256 / g.c + (256 % g.c ? 1 : 0);
    g.nslice = nwin1;
    static_assert(sizeof(MsmGeom::off) / sizeof(uint16_t) == 36, "MsmGeom::off");
    if (nwin1 + 1 > 36) return fail(ZKP_E_ARG, "more than 35 windows per scalar");
    for (uint32_t s = 0; s <= nwin1 && s < 36; s++) g.off[s] = shared ? bases->pre_off[s] : (uint16_t)(s * g.c);
    g.shared = shared ? 1u : 0u;
    // Shared mode walks the scalars in ranges of at most 2^23: the expanded bases of a range are 13 x 2^23 x 128 B = 14 GB,
    // and random 128-byte reads over a larger footprint fall off a translation cliff (accumulate: 6.3 G adds/s up to 2^23,
    // 4.5 G/s at 2^24 in one range, profiles/r01_f_shared_buckets.md).  Later ranges add into the same buckets.
    uint64_t range = n;            // the longest range: what the workspaces and the sort geometry are sized for
    std::vector<uint64_t> lens;    // the scalar ranges in order (host-fed scalars: one or two short ranges first, see msm_host_scalars)
    if (shared) {
        // ... measured again in round 2 with 12 planes (22-bit windows): ranges of 2^24 (25.8 GB of planes) are still fine -- 2^24 32.73 ->
        // 32.31 ms in one range, 2^26 128.4 -> 126.6 ms -- and 2^25 (51.5 GB) is over the cliff (2^25 in one range 80.0 against 63.8 ms):
        // profiles/r02_j_sort_under_accumulate.md
        const uint64_t max_range = bases->pre_planes <= 12 ? 1ull << 24 : 1ull << 23;
        uint64_t cap = feed ? std::min<uint64_t>(max_range, 1ull << feed->range_log) : max_range;
        if (const char* e = getenv("ZKP_MSM_RANGE_LOG")) {
            int v = atoi(e);
            if (v >= 10 && v <= 30) cap = 1ull << v;
        }
        const uint64_t npass = (n + cap - 1) / cap;
        range = (n + npass - 1) / npass;
        uint64_t want_first = feed ? feed->first_len : 0;
        if (!feed && count == 1)
            if (const char* e = getenv("ZKP_MSM_FIRST_PCT")) {  // tuning aid (resident scalars): a short first range whose sort is the exposed one
                const int v = atoi(e);
                if (v >= 1 && v <= 90) want_first = std::max<uint64_t>(1024, ((uint64_t)n * v / 100) & ~(uint64_t)1023);
            }
        uint64_t done = 0;  // short ranges first (they obey the range limit like the others), then the rest in equal ranges of at most `cap`
        for (uint64_t want : {want_first, feed && want_first ? feed->second_len : (uint64_t)0})
            if (want && done + want < n) {
                lens.push_back(std::min<uint64_t>(want, cap));
                done += lens.back();
            }
        const uint64_t rest = n - done, rpass = (rest + cap - 1) / cap, rr = (rest + rpass - 1) / rpass;
        for (; done < n; done += lens.back()) lens.push_back(std::min<uint64_t>(rr, n - done));
        range = *std::max_element(lens.begin(), lens.end());
    } else {
        lens.push_back(n);
    }
    g.resume = 0;
    g.more = 0;
    g.interleave = 0;  // set below once the geometry is known
    g.ns = range;
    g.plane_stride = bases->n;
    g.nwin = shared ? (uint32_t)count : nwin1 * (uint32_t)count;  // sort windows = bucket sets
    g.n = shared ? (uint64_t)nwin1 * range : n;                        // entries per sort window
    if (g.n >= (1ull << 31)) return fail(ZKP_E_ARG, "windows x scalars >= 2^31 with expanded bases");
    g.nb = 1u << (g.c - 1);
    g.interleave = (g.nwin > 1 && g.n <= (1ull << 22)) ? 1u : 0u;  // measured: +5 % at 2^22, 0 at 2^23, -5 % at 2^24
    const uint64_t entries = g.n;
    uint32_t want = std::max<uint32_t>(1, (512 + g.nwin - 1) / g.nwin);
    uint64_t maxchunks = (entries + 4095) / 4096;
    g.nchunk = (uint32_t)std::min<uint64_t>(want, maxchunks);
    if (const char* e = getenv("ZKP_MSM_NCHUNK")) {
        int v = atoi(e);
        if (v >= 1 && v <= 4096) g.nchunk = (uint32_t)std::min<uint64_t>((uint64_t)v, entries);
    }
    g.chunk = (entries + g.nchunk - 1) / g.nchunk;
    // a bucket is oversized above 4x the average run; its pieces are no longer than an average run (they execute next to
    // the ordinary lanes, so a longer piece would become the critical path)
    g.run_limit = (uint32_t)std::max<uint64_t>(128, 4 * (entries / g.nb));
    g.piece = (uint32_t)std::max<uint64_t>(32, entries / g.nb);
    SortGeom sg;
    // 2^19 buckets: 1024 partitions x 512 bins (first-pass runs of 4 entries per partition and tile instead of 2; measured
    // sort 0.245 -> 0.225 ms at 2^20, 3.41 -> 3.31 ms at 2^24; 512 x 1024 is slower again)
    // wider windows (21..24 bits over an expanded SRS: 12 or 11 slices instead of 13): 1024 bins per partition
    sg.lo_bits = std::min<uint32_t>(g.c >= 21 ? 10 : g.c >= 20 ? 9 : 8, g.c - 1);
    if (const char* e = getenv("ZKP_SORT_LO_BITS")) {  // tuning aid
        const int v = atoi(e);
        if (v >= 6 && v <= 10 && (uint32_t)v < g.c) sg.lo_bits = (uint32_t)v;
    }
    sg.nhi = g.nb >> sg.lo_bits;
    if (sg.nhi > SORT_MAX_PART) return fail(ZKP_E_ARG, "window width above 24 bits is not supported by the sort");
    const size_t W = g.nwin, nb = g.nb, c = g.c;
    // Several scalar ranges: the digits + sort of range r+1 run on a second stream while range r is being accumulated (the sort is
    // memory-bound, the accumulate issue-bound); what the accumulate reads (sorted indices, bucket starts, size order, piece
    // descriptors) is double-buffered for it.
    const bool overlap = shared && range < n && !getenv("ZKP_MSM_NO_OVERLAP");
    const size_t nbuf = overlap ? 2 : 1;
    ZCHK(ctx().digits.ensure(4 * W * entries));
    ZCHK(ctx().sorted.ensure(nbuf * 4 * W * entries));
    ZCHK(ctx().counts.ensure(4 * W * ((size_t)g.nchunk * sg.nhi + 2 * sg.nhi + 1 + 512)));
    ZCHK(ctx().entries.ensure(8 * W * entries));
    ZCHK(ctx().start.ensure(nbuf * 4 * W * (nb + 2)));
    ZCHK(ctx().perm.ensure(nbuf * 4 * W * nb));
    // oversized-bucket bookkeeping (msm_order): at most n / LIMIT oversized buckets and n / PIECE + that many pieces
    const uint32_t over_cap = (uint32_t)std::min<uint64_t>(entries / 128 + 1, (uint64_t)nb);  // also bounds the saturated bin
    const uint32_t desc_cap = (uint32_t)(entries / g.piece + entries / g.run_limit + 2);
    const size_t over_bytes = ((4 * W * (2 + (size_t)over_cap + over_cap + 1) + 16 * W * (size_t)desc_cap) + 255) & ~(size_t)255;
    ZCHK(ctx().over.ensure(nbuf * over_bytes));
    ZCHK(ctx().pieces.ensure(256 * W * (size_t)desc_cap));
    ZCHK(ctx().buckets.ensure(256 * W * nb));
    // Few buckets for the machine (a small MSM over narrow windows, single pass): 2 or 4 lanes / quads share a bucket's run
    // (msm.hpp, split_run) so that narrow windows -- a short bucket reduction -- still fill the SIMDs.
    g.split_log = 0;
    {
        const bool quad_kernel = (uint64_t)g.n * g.nwin <= (1ull << 20);  // (the choice made at the launch below)
        const uint64_t units = (uint64_t)nb * W, want_units = quad_kernel ? SPLIT_FILL_QUADS : SPLIT_FILL_LANES;
        if (range >= n)
            while (g.split_log < MSM_MAX_SPLIT_LOG && (units << g.split_log) < want_units) g.split_log++;
        if (const char* e = getenv("ZKP_MSM_SPLIT_LOG")) {  // tuning aid
            const int v = atoi(e);
            if (v >= 0 && v <= (int)MSM_MAX_SPLIT_LOG && range >= n) g.split_log = (uint32_t)v;
        }
    }
    if (g.split_log) ZCHK(ctx().parts.ensure(256 * W * nb * ((1u << g.split_log) - 1)));
    ZCHK(ctx().pyr1.ensure(256 * W * nb));
    ZCHK(ctx().odd0.ensure(256 * W * nb));
    ZCHK(ctx().odd1.ensure(256 * W * nb));
    // The c result points of every bucket set are written by the last kernel straight into pinned host memory (device-accessible:
    // 5 KB over PCIe), followed by one flag word per bucket set; a device-to-host copy would be one more (blit) kernel launch per
    // MSM.  The device buffer only holds the barrier counters of the last-levels launch, one 128-byte line each.
    ZCHK(ctx().result.ensure(4 * PYR_BAR_STRIDE * W));
    if (ctx().host_result_cap < 256 * W * c + 4 * W) {
        if (ctx().host_result) HIPCHK(hipHostFree(ctx().host_result));
        ctx().host_result = nullptr;
        ctx().host_result_cap = 0;
        HIPCHK(hipHostMalloc(&ctx().host_result, 256 * W * c + 4 * W, hipHostMallocPortable | hipHostMallocMapped));  // written by this slot's device
        ctx().host_result_cap = 256 * W * c + 4 * W;
    }
    uint32_t* digits = reinterpret_cast<uint32_t*>(ctx().digits.p);
    uint32_t* const sorted0 = reinterpret_cast<uint32_t*>(ctx().sorted.p);
    uint32_t* counts = reinterpret_cast<uint32_t*>(ctx().counts.p);
    uint32_t* ptot = counts + W * (size_t)g.nchunk * sg.nhi;   // W x nhi
    uint32_t* pstart = ptot + W * (size_t)sg.nhi;               // W x (nhi + 1)
    uint32_t* ghist = pstart + W * (size_t)(sg.nhi + 1);        // W x 256 size histogram, then W x 256 rank cursors
    uint32_t* gcur = ghist + W * 256;
    uint2* entries_buf = reinterpret_cast<uint2*>(ctx().entries.p);
    uint32_t* const start0 = reinterpret_cast<uint32_t*>(ctx().start.p);
    uint32_t* const perm0 = reinterpret_cast<uint32_t*>(ctx().perm.p);
    uint4* pieces = reinterpret_cast<uint4*>(ctx().pieces.p);
    uint4* buckets = reinterpret_cast<uint4*>(ctx().buckets.p);
    uint4* parts = reinterpret_cast<uint4*>(ctx().parts.p);
    uint4* carry = reinterpret_cast<uint4*>(ctx().pyr1.p);  // hand-over array between scalar ranges: the reduction's second buffer, idle until then
    uint32_t* tail_bar = reinterpret_cast<uint32_t*>(ctx().result.p);
    uint4* const result_out = reinterpret_cast<uint4*>(ctx().host_result);                                       // W x c points, then
    uint32_t* const result_flags = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx().host_result) + 256 * W * c);  // W flag words

#ifdef ZKP_MSM_CHECK  // diagnosis builds: where every workspace lives, to place a faulting address
    {
        static bool once = false;
        if (!once) {
            once = true;
            auto show = [](const char* name, const void* q, size_t bytes) {
                fprintf(stderr, "ZKP_MSM_CHECK %-10s %p .. %p (%zu bytes)\n", name, q, static_cast<const char*>(q) + bytes, bytes);
            };
            show("bases", bases->d_xy, (size_t)bases->n * 128 * (bases->pre_planes ? bases->pre_planes : 1));
            show("scalars", d_scalars[0], 32 * n);
            show("digits", ctx().digits.p, ctx().digits.cap);
            show("sorted", ctx().sorted.p, ctx().sorted.cap);
            show("counts", ctx().counts.p, ctx().counts.cap);
            show("entries", ctx().entries.p, ctx().entries.cap);
            show("start", ctx().start.p, ctx().start.cap);
            show("perm", ctx().perm.p, ctx().perm.cap);
            show("over", ctx().over.p, ctx().over.cap);
            show("pieces", ctx().pieces.p, ctx().pieces.cap);
            show("buckets", ctx().buckets.p, ctx().buckets.cap);
            show("pyr1", ctx().pyr1.p, ctx().pyr1.cap);
            show("odd0", ctx().odd0.p, ctx().odd0.cap);
            show("odd1", ctx().odd1.p, ctx().odd1.cap);
            show("result", ctx().result.p, ctx().result.cap);
            show("host_res", ctx().host_result, ctx().host_result_cap);
            fprintf(stderr, "ZKP_MSM_CHECK geometry: n %zu range %llu first %llu rest %llu entries %llu nb %u nchunk %u over_cap %u desc_cap %u run_limit %u piece %u\n",
                    n, (unsigned long long)range, (unsigned long long)lens[0], (unsigned long long)lens.back(), (unsigned long long)entries, g.nb, g.nchunk,
                    over_cap, desc_cap, g.run_limit, g.piece);
        }
    }
#endif
    hipStream_t sst = st;  // stream of the digits + sort kernels
    if (overlap) {
        Ctx& cx = ctx();
        if (!cx.sort_stream) {
            HIPCHK(hipStreamCreateWithFlags(&cx.sort_stream, hipStreamNonBlocking));
            for (hipEvent_t* e : {&cx.ev_sort[0], &cx.ev_sort[1], &cx.ev_acc[0], &cx.ev_acc[1], &cx.ev_begin})
                HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        sst = cx.sort_stream;
        HIPCHK(hipEventRecord(cx.ev_begin, st));  // whatever the caller enqueued before (the scalars) comes first
        HIPCHK(hipStreamWaitEvent(sst, cx.ev_begin, 0));
    }
    uint64_t ridx = 0;
    // Host-fed scalars: the first range is uploaded by this thread, all later ones by the slot's uploader thread, started before the first
    // range's kernels are enqueued (see Uploader); upload_issued = ranges whose copy and event record have been issued.
    std::atomic<uint64_t> upload_issued{0};
    std::atomic<int> upload_rc{ZKP_OK}, upload_go{0};
    bool upload_submitted = false;
    auto walk_ranges = [&]() -> int {
    for (uint64_t off = 0, len = 0; off < n; off += len, ridx++) {
        len = lens[ridx];
        const size_t par = overlap ? (ridx & 1) : 0;  // buffer set of this range
        uint32_t* sorted = sorted0 + par * W * entries;
        uint32_t* start = start0 + par * W * (nb + 2);
        uint32_t* perm = perm0 + par * W * nb;
        uint4* desc = reinterpret_cast<uint4*>(reinterpret_cast<char*>(ctx().over.p) + par * over_bytes);  // W x desc_cap (16-byte aligned first)
        uint32_t* over = reinterpret_cast<uint32_t*>(desc + W * (size_t)desc_cap);                        // W x 2
        uint32_t* over_b = over + 2 * W;                                                                   // W x over_cap
        uint32_t* over_off = over_b + W * (size_t)over_cap;                                                // W x (over_cap + 1)
        if (overlap && ridx >= 2) HIPCHK(hipStreamWaitEvent(sst, ctx().ev_acc[par], 0));  // range r-2 is done with this buffer set
        if (len != g.ns) {  // a range shorter than the longest (the last one, or the first of a host-fed walk): same buffers, smaller geometry
            g.ns = len;
            g.n = (uint64_t)nwin1 * len;
            g.chunk = (g.n + g.nchunk - 1) / g.nchunk;
        }
        g.resume = off ? 1u : 0u;
        g.more = off + len < n ? 1u : 0u;
        if (feed) {  // this range's scalars: host -> device on the copy stream, the kernels below wait for them
            hipEvent_t ev = feed->ev;
            if (ridx == 0) {
                // the uploader is woken FIRST and spins on upload_go while this thread is held by the first copy: a sleeping thread
                // takes 50-300 us to come back, as long as the first upload itself
                if (len < n) {
                    Ctx& cx = ctx();
                    const size_t later = lens.size() - 1;
                    while (cx.copy_events.size() < later) {
                        hipEvent_t e = nullptr;
                        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        cx.copy_events.push_back(e);
                    }
                    const int device = cx.device;
                    Fr* d_dst = const_cast<Fr*>(d_scalars[0]);
                    const uint64_t* h_src = feed->h_scalars;
                    hipStream_t cs = feed->copy_stream;
                    const hipEvent_t* evs = cx.copy_events.data();
                    const uint64_t first = len;
                    uploader(cx.slot).submit([=, &upload_issued, &upload_rc, &upload_go, &lens]() -> int {
                        int rc = ZKP_OK;
                        if (hipSetDevice(device) != hipSuccess) rc = ZKP_E_DEVICE;
                        int go;
                        while ((go = upload_go.load(std::memory_order_acquire)) == 0) __builtin_ia32_pause();  // the first copy is in the stream
                        if (go < 0) return ZKP_OK;  // the caller gave up
                        size_t k = 0;
                        for (uint64_t o = first; rc == ZKP_OK && o < n; k++) {
                            const uint64_t l = lens[k + 1];
                            if (hipMemcpyAsync(d_dst + o, h_src + 4 * o, 32 * l, hipMemcpyHostToDevice, cs) != hipSuccess ||
                                hipEventRecord(evs[k], cs) != hipSuccess)
                                rc = ZKP_E_DEVICE;
                            else
                                upload_issued.store(k + 1, std::memory_order_release);
                            o += l;
                        }
                        if (rc != ZKP_OK) upload_rc.store(rc, std::memory_order_release);
                        return rc;
                    });
                    upload_submitted = true;
                }
                hipError_t e1 = hipMemcpyAsync(const_cast<Fr*>(d_scalars[0]), feed->h_scalars, 32 * len, hipMemcpyHostToDevice, feed->copy_stream);
                if (e1 == hipSuccess) e1 = hipEventRecord(feed->ev, feed->copy_stream);
                upload_go.store(e1 == hipSuccess ? 1 : -1, std::memory_order_release);
                HIPCHK(e1);
            } else {
                while (upload_issued.load(std::memory_order_acquire) < ridx) {  // (host only: the GPU is busy with the ranges before)
                    if (upload_rc.load(std::memory_order_acquire) != ZKP_OK) return fail(ZKP_E_DEVICE, "upload of a scalar range failed");
                    __builtin_ia32_pause();
                }
                ev = ctx().copy_events[ridx - 1];
            }
            HIPCHK(hipStreamWaitEvent(sst, ev, 0));
        }
        {
            ProfScope ps("msm_digits", sst);
            DigitSources ds;  // digits laid out [msm][slice][scalar]: a shared-mode sort window is one msm
            for (size_t m = 0; m < count; m++) ds.scalars[m] = d_scalars[m] + off;
            hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((len + MSM_THREADS - 1) / MSM_THREADS), (unsigned)count),
                               dim3(MSM_THREADS), 0, sst, ds, bases->d_inf ? bases->d_inf + off : nullptr, g, nwin1, digits);
            MSM_TRACE(sst, "digits");
        }
        {
            ProfScope ps("msm_sort", sst, true);
            hipLaunchKernelGGL(msm_parthist_kernel, dim3(g.nchunk, g.nwin), dim3(1024), 0, sst, digits, g, sg, counts);
            MSM_TRACE(sst, "parthist");
            hipLaunchKernelGGL(msm_partprefix_kernel, dim3((sg.nhi + 63) / 64, g.nwin), dim3(1024), 0, sst, counts, g, sg, ptot);
            hipLaunchKernelGGL(msm_partstart_kernel, dim3(g.nwin), dim3(64), 0, sst, ptot, sg, pstart, ghist, tail_bar);
            MSM_TRACE(sst, "partprefix + partstart");
            const int ps_tile = partscatter_tile(sg.nhi);  // the largest tile whose staging fits the LDS next to 12 bytes per partition
            if (ps_tile == PS_TILE_SMALL)
                hipLaunchKernelGGL(msm_partscatter_kernel<PS_TILE_SMALL>, dim3(g.nchunk, g.nwin), dim3(1024),
                                   partscatter_lds_bytes(sg.nhi, PS_TILE_SMALL), sst, digits, g, sg, counts, pstart, entries_buf);
            else if (ps_tile == PS_TILE_MID)
                hipLaunchKernelGGL(msm_partscatter_kernel<PS_TILE_MID>, dim3(g.nchunk, g.nwin), dim3(1024),
                                   partscatter_lds_bytes(sg.nhi, PS_TILE_MID), sst, digits, g, sg, counts, pstart, entries_buf);
            else
                hipLaunchKernelGGL(msm_partscatter_kernel<PS_TILE_BIG>, dim3(g.nchunk, g.nwin), dim3(1024),
                                   partscatter_lds_bytes(sg.nhi, PS_TILE_BIG), sst, digits, g, sg, counts, pstart, entries_buf);
            MSM_TRACE(sst, "partscatter");
            hipLaunchKernelGGL(msm_binsort_kernel, dim3(sg.nhi, g.nwin), dim3(1024), 0, sst, entries_buf, g, sg, pstart, start,
                               sorted, ghist);
            MSM_TRACE(sst, "binsort");
            const dim3 rank_grid((g.nb + 1023) / 1024, g.nwin);
            hipLaunchKernelGGL(msm_rank_kernel, rank_grid, dim3(1024), 0, sst, start, g, ghist, gcur, perm);
            MSM_TRACE(sst, "rank");
            hipLaunchKernelGGL(msm_order_kernel, dim3(g.nwin), dim3(1024), 0, sst, start, g, ghist, perm, over, over_b, over_off, desc,
                               over_cap, desc_cap);
            MSM_TRACE(sst, "order");
        }
        if (overlap) {
            HIPCHK(hipEventRecord(ctx().ev_sort[par], sst));
            HIPCHK(hipStreamWaitEvent(st, ctx().ev_sort[par], 0));
        }
        {
            ProfScope ps("msm_accumulate", st, true);
            // few entries: the lane-per-bucket kernel would be latency-bound by its longest run -> four lanes per bucket
            // measured (tools/small_msm_bench.py, accumulate us lane -> quad): 2^16 x1 401 -> 293, x2 454 -> 525; 2^14 x3 261 -> 209;
            // 2^12 x1 109 -> 60: four lanes per bucket up to 2^20 entries
            const bool quad = !g.resume && !g.more && (uint64_t)g.n * g.nwin <= (1ull << 20);
            const uint32_t per_block = quad ? ACC_THREADS / 4 : ACC_THREADS;
            const uint32_t bucket_blocks = (uint32_t)((((uint64_t)g.nb << g.split_log) + per_block - 1) / per_block);
            const uint32_t extra_blocks = std::min<uint32_t>((desc_cap + per_block - 1) / per_block, 64);
            if (quad)
                hipLaunchKernelGGL(msm_accumulate_quad_kernel, dim3((bucket_blocks + extra_blocks) * g.nwin), dim3(ACC_THREADS), 0,
                                   st, reinterpret_cast<const uint4*>(bases->d_xy) + off * 8, sorted, start, perm, over, desc,
                                   desc_cap, bucket_blocks, extra_blocks, g, buckets, pieces, parts, clk_record(CLK_MSM_ACCUMULATE));
            else
                hipLaunchKernelGGL(msm_accumulate_kernel, dim3((bucket_blocks + extra_blocks) * g.nwin), dim3(ACC_THREADS), 0, st,
                                   reinterpret_cast<const uint4*>(bases->d_xy) + off * 8, sorted, start, perm, over, desc, desc_cap,
                                   bucket_blocks, extra_blocks, g, buckets, pieces, parts, carry, clk_record(CLK_MSM_ACCUMULATE));
            MSM_TRACE(st, "accumulate");
            hipLaunchKernelGGL(msm_combine_kernel, dim3(std::min<uint32_t>(over_cap, 64), g.nwin), dim3(64), 0, st, over, over_b,
                               over_off, over_cap, desc_cap, g, pieces, buckets, carry);
            MSM_TRACE(st, "combine");
            if (g.split_log) {  // buckets += parts, pairwise: split_log steps
                const uint64_t cap = (uint64_t)g.nwin * g.nb;
                const unsigned fold_x = (unsigned)((cap + MSM_THREADS / 4 - 1) / (MSM_THREADS / 4));
                // a step with many adds runs one lane per add, a small one four lanes per add (latency): msm.hpp
                static const uint64_t lane_from = getenv("ZKP_FOLD_LANE_MIN") ? strtoull(getenv("ZKP_FOLD_LANE_MIN"), nullptr, 10) : FOLD_LANE_MIN_ADDS;
                for (uint32_t t = 0; t < g.split_log; t++) {
                    const unsigned pairs = 1u << (g.split_log - 1 - t);
                    if (cap * pairs >= lane_from)
                        hipLaunchKernelGGL(msm_fold_parts_lane_kernel, dim3((unsigned)((cap + MSM_THREADS - 1) / MSM_THREADS), pairs),
                                           dim3(MSM_THREADS), 0, st, buckets, parts, cap, t);
                    else
                        hipLaunchKernelGGL(msm_fold_parts_kernel, dim3(fold_x, pairs), dim3(MSM_THREADS), 0, st, buckets, parts, cap, t);
                }
            }
        }
        if (overlap) HIPCHK(hipEventRecord(ctx().ev_acc[par], st));
    }
    return ZKP_OK;
    };
    const int walk_rc = walk_ranges();
    int up_rc = ZKP_OK;
    if (upload_submitted) up_rc = uploader(ctx().slot).wait();  // (its job refers to this frame: joined on every path)
    if (walk_rc != ZKP_OK || up_rc != ZKP_OK) {
        // an early return out of the walk can leave digits / sort kernels queued on the second stream that were never joined back
        // into st; the caller's WsOrder event covers st only, so drain them here before the workspaces can be handed to the next entry
        if (overlap) (void)hipStreamSynchronize(sst);
        if (feed) (void)hipStreamSynchronize(feed->copy_stream);
        return walk_rc != ZKP_OK ? walk_rc : fail(ZKP_E_DEVICE, "upload of a scalar range failed");
    }
    HIPCHK(hipGetLastError());
    for (size_t w = 0; w < W; w++) __atomic_store_n(result_flags + w, MSM_FLAG_PENDING, __ATOMIC_RELEASE);  // (the previous MSM's results were read before it returned)
    uint4* pyr[2] = {buckets, reinterpret_cast<uint4*>(ctx().pyr1.p)};
    uint4* odd[2] = {reinterpret_cast<uint4*>(ctx().odd0.p), reinterpret_cast<uint4*>(ctx().odd1.p)};
    ProfScope* ps_red = new ProfScope("msm_bucket_reduce", st, true);
    // tuning aids (A/B runs): workgroup size and count of the last-levels launch, and the per-array pair count from which it takes over
    static const uint32_t tail_threads = getenv("ZKP_PYR_TAIL_THREADS") ? (uint32_t)atoi(getenv("ZKP_PYR_TAIL_THREADS")) : PYR_TAIL_THREADS;
    static const uint32_t tail_blocks = getenv("ZKP_PYR_TAIL_BLOCKS") ? (uint32_t)atoi(getenv("ZKP_PYR_TAIL_BLOCKS")) : PYR_TAIL_BLOCKS;
    static const uint32_t tail_half = getenv("ZKP_PYR_TAIL_HALF") ? (uint32_t)atoi(getenv("ZKP_PYR_TAIL_HALF")) : 64u;
    if (tail_threads < 64 || tail_threads > 512 || (tail_threads & 63) || !tail_blocks || tail_blocks > 256 || !tail_half)
        return fail(ZKP_E_ARG, "ZKP_PYR_TAIL_THREADS must be a multiple of 64 up to 512, ZKP_PYR_TAIL_BLOCKS 1..256, ZKP_PYR_TAIL_HALF >= 1");
    uint32_t level_tail = 0;  // first level whose per-array work is <= 64 pairs: the rest runs in one launch
    while (level_tail + 1 < g.c && (g.nb >> (level_tail + 1)) > tail_half) level_tail++;
    // ... unless even one workgroup per bucket set is more than the device keeps resident (many bucket sets, a partition with few
    // CUs): the barrier of that launch would spin for its whole time-out, so every level runs as its own launch instead
    const uint32_t max_waves = ctx().tail_max_waves;
    if ((uint64_t)g.nwin * (tail_threads / 64) > max_waves) level_tail = g.c - 1;
    for (uint32_t l = 0; l < level_tail; l++) {
        PyrLevel L;
        L.level = l;
        L.half = g.nb >> (l + 1);
        L.nb = g.nb;
        L.nwin = g.nwin;
        // levels with few adds are latency-bound: four lanes per add there (measured: 16 us -> ~6 us per level)
        const uint64_t adds = (uint64_t)L.half * (l + 1) * g.nwin;
        if (adds <= (1u << 16))  // threshold swept 2^14..2^20: 2^16 is the minimum of the reduction time
            hipLaunchKernelGGL(msm_pyramid_quad_kernel, dim3((L.half + MSM_THREADS / 4 - 1) / (MSM_THREADS / 4), l + 1, g.nwin),
                               dim3(MSM_THREADS), 0, st, pyr[l & 1], pyr[(l + 1) & 1], odd[l & 1], odd[(l + 1) & 1], L);
        else
            hipLaunchKernelGGL(msm_pyramid_kernel, dim3((L.half + MSM_THREADS - 1) / MSM_THREADS, l + 1, g.nwin),
                               dim3(MSM_THREADS), 0, st, pyr[l & 1], pyr[(l + 1) & 1], odd[l & 1], odd[(l + 1) & 1], L);
    }
    if (level_tail + 1 < g.c) {
        uint32_t* bar = tail_bar;  // zeroed by msm_partstart
        uint32_t tb = tail_blocks;
        while (tb > 1 && (uint64_t)tb * g.nwin * (tail_threads / 64) > max_waves) tb >>= 1;
        // test hook (tests/test_gpu_parity.py): ask the barrier for one arrival more than there are workgroups, with a short
        // time-out -- the path a workgroup that never became resident would take: MSM_TAIL_TIMEOUT flag, ZKP_E_DEVICE below
        const bool starve = getenv("ZKP_TEST_TAIL_STARVE") != nullptr;
        hipLaunchKernelGGL(msm_pyramid_tail_kernel, dim3(tb, g.nwin), dim3(tail_threads), 0, st, pyr[0], pyr[1], odd[0],
                           odd[1], level_tail, g.c, g.nb, bar, result_out, result_flags, starve ? tb + 1 : tb,
                           starve ? (1u << 12) : PYR_TAIL_SPIN_LIMIT);
    } else {  // every level already ran as its own launch: only the gathering is left
        const uint32_t fin = (g.c - 1) & 1;
        hipLaunchKernelGGL(msm_collect_kernel, dim3(g.nwin), dim3(64), 0, st, pyr[fin], pyr[fin ^ 1], odd[fin], g.nb, g.c,
                           result_out, result_flags);
    }
    delete ps_red;
    HIPCHK(hipGetLastError());
#ifdef ZKP_MSM_CHECK
    {
        uint32_t chk[32];
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpyFromSymbol(chk, HIP_SYMBOL(g_msm_check), sizeof(chk)));
        for (int k = 0; k < 8; k++)
            if (chk[4 * k]) fprintf(stderr, "ZKP_MSM_CHECK class %d: %u violations, first (%u, %u)\n", k, chk[4 * k], chk[4 * k + 1], chk[4 * k + 2]);
    }
#endif
    // The last kernel writes the result points and then one flag word per bucket set straight into pinned host memory.  Up to 2^24
    // entries per bucket set (an MSM of a few milliseconds) the host polls those flags instead of waiting for the stream: the
    // runtime's wait costs 30-60 us of wake-up latency per MSM -- a quarter of the idle time of a 2^16-gate PLONK proof, which
    // makes four of them on its critical path, and 1 % of a 2^20-term MSM (profiles/r05_k).  A kernel that never writes its flag
    // (a fault) is caught by the stream wait the poll falls back to after two seconds.
    if (count > 1 && !getenv("ZKP_POOL_NO_WARM")) host_pool().warm(std::chrono::microseconds(3000));  // the tails below run on the pool: wake it now
    bool seen = false;
    if (g.n <= (1ull << 24) && !getenv("ZKP_MSM_NO_POLL")) {
        const auto t_poll0 = std::chrono::steady_clock::now();
        uint64_t spins = 0;
        for (;;) {
            bool all = true;
            for (size_t w = 0; w < W && all; w++)
                all = (__atomic_load_n(result_flags + w, __ATOMIC_ACQUIRE) & MSM_FLAG_PENDING) == 0;
            if (all) {
                seen = true;
                break;
            }
            if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t_poll0 > std::chrono::seconds(2)) break;
            __builtin_ia32_pause();
        }
    }
    if (!seen) HIPCHK(hipStreamSynchronize(st));  // the kernels' writes to the pinned buffer are visible to the host from here on
    for (size_t w = 0; w < W; w++)
        if (result_flags[w] & MSM_TAIL_TIMEOUT)
            return fail(ZKP_E_DEVICE, "bucket reduction: the workgroups of the last levels did not all become resident (device shared "
                                      "with another job?); no result was produced");
    const auto t_tail0 = std::chrono::steady_clock::now();

    // serial tail on the host.  Per bucket set: V = S + sum_l 2^l U_l.  Per-window mode: total = sum_w 2^(c w) V_w, and
    // every (w, l) lands on its own bit position c w + l, so ONE Horner chain over the positions does it with c W
    // doublings.  Shared mode: the expanded bases already carry the 2^(c w) factors, total = V of the single bucket set.
    const uint32_t wins_per_msm = shared ? 1u : nwin1;
    const uint32_t* host_res = reinterpret_cast<const uint32_t*>(ctx().host_result);  // (the pool's threads are in no context)
    auto tail = [&](size_t m) {
        const uint32_t* res = host_res + m * wins_per_msm * c * 64;  // 64 words / point
        HXyzz total = HXyzz::infinity();
        for (int pos = (int)(wins_per_msm * c) - 1; pos >= 0; pos--) {
            total = total.dbl();
            const int w = pos / (int)c, l = pos % (int)c;
            const uint32_t* rw = res + (size_t)w * c * 64;
            if (l <= (int)c - 2) total = total.add(xyzz_from_internal(rw + (size_t)(1 + l) * 64));
            if (l == 0) total = total.add(xyzz_from_internal(rw));
        }
        out[m] = total;
    };
    if (count == 1) {
        tail(0);
    } else {  // the tails of a batch are independent serial chains: spread over the resident host workers
        const std::function<void(size_t)> job = tail;
        host_pool().run(job, count);
    }
    prof_host("msm_tail_host", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail0).count());
    return ZKP_OK;
}

int msm_partial(const zkp_bases* bases, const Fr* d_scalars, size_t n, hipStream_t st, HXyzz* out) {
    return msm_partial_batch(bases, &d_scalars, 1, n, st, out);
}

int ensure_fixed_base_table(hipStream_t st) {
    if (ctx().fb_ready) return ZKP_OK;
    // table[w * 255 + (d - 1)] = d * 2^(8 w) * G, affine; built on the host once (8160 points)
    static const uint64_t gx[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                   0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
    static const uint64_t gy[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                   0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
    HXyzz base{HFq::load(gx).to_mont(), HFq::load(gy).to_mont(), HFq::one(), HFq::one()};
    const int NW = 32, ND = 255;
    std::vector<HXyzz> pts((size_t)NW * ND);
    for (int w = 0; w < NW; w++) {
        pts[(size_t)w * ND] = base;
        for (int d = 1; d < ND; d++) pts[(size_t)w * ND + d] = pts[(size_t)w * ND + d - 1].add(base);
        base = pts[(size_t)w * ND + ND - 1].add(base);
    }
    // batch-normalise: one inversion for all ZZZ (Montgomery's trick); none of these points is the identity
    std::vector<HFq> pref(pts.size());
    HFq acc = HFq::one();
    for (size_t i = 0; i < pts.size(); i++) {
        pref[i] = acc;
        acc = acc * pts[i].zzz;
    }
    HFq inv = acc.inverse();
    std::vector<uint64_t> tab(pts.size() * 12);
    for (size_t i = pts.size(); i-- > 0;) {
        HFq zi3 = inv * pref[i];
        inv = inv * pts[i].zzz;
        HFq zi2 = (zi3 * pts[i].zz).sqr();
        (pts[i].x * zi2).store(&tab[i * 12]);
        (pts[i].y * zi3).store(&tab[i * 12 + 6]);
    }
    ZCHK(ctx().fb_table.ensure(tab.size() * 8));
    HIPCHK(hipMemcpyAsync(ctx().fb_table.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx().fb_ready = true;
    return ZKP_OK;
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================
extern "C" {

int zkp_abi_version(void) { return 1; }

void zkp_profile_enable(int on) { g_prof_on.store(on == 2 ? 2 : on != 0 ? 1 : 0); }
void zkp_profile_reset(void) {
    std::lock_guard<std::mutex> g(g_rt.mu);
    for (Ctx* c : g_rt.slots) {
        std::lock_guard<std::mutex> lk(c->mu);
        for (ProfRec& r : c->prof) {
            if (r.a && r.owns_a) (void)hipEventDestroy(r.a);
            if (r.b) (void)hipEventDestroy(r.b);
        }
        c->prof.clear();
        if (c->clk.p) {
            int prev = 0;
            const bool restore = hipGetDevice(&prev) == hipSuccess && prev != c->device;
            if (restore) (void)hipSetDevice(c->device);
            (void)hipDeviceSynchronize();
            (void)hipMemset(c->clk.p, 0, sizeof(ClkRec) * CLK_COUNT);
            if (restore) (void)hipSetDevice(prev);
        }
    }
}
// In-kernel clock stamps of the instrumented kernel families, summed over the device slots (msm.hpp, ClkRec): the shader clock held
// under that kernel's load is cycles / ref_ticks x 100 MHz.  Synchronises the devices.
struct DeviceRestore {  // the caller keeps its current HIP device, whichever way an entry leaves
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};
int zkp_profile_clock_read(const char* name, uint64_t* cycles, uint64_t* ref_ticks, uint64_t* waves) try {
    if (!name || !cycles || !ref_ticks || !waves) return fail(ZKP_E_ARG, "null argument");
    int which = -1;
    for (int i = 0; i < CLK_COUNT; i++)
        if (std::strcmp(name, kClkNames[i]) == 0) which = i;
    if (which < 0) return fail(ZKP_E_ARG, "no clock stamps under this name (msm_accumulate, mad_probe, ntt_fr_pass, ntt_gl_pass)");
    std::lock_guard<std::mutex> g(g_rt.mu);
    *cycles = *ref_ticks = *waves = 0;
    DeviceRestore restore;
    for (Ctx* c : g_rt.slots) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->clk.p) continue;
        ClkRec r;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(&r, reinterpret_cast<ClkRec*>(c->clk.p) + which, sizeof r, hipMemcpyDeviceToHost));
        *cycles += r.cycles;
        *ref_ticks += r.ref;
        *waves += r.waves;
    }
    return ZKP_OK;
} ZKP_CATCH_INT

// v_mad_u64_u32 issue rate of this device, now: `launches` back-to-back launches of mad_rate_probe_kernel (8 blocks of 256 lanes per
// CU, ~1 ms each) timed with HIP events on the slot's stream, with the clock its waves saw.  What bench.py divides the multiply-add
// rate of msm_accumulate by, in the same run on the same box (instead of a constant measured once on another one).
int zkp_probe_mad_rate(unsigned launches, double* lane_mads_per_s, double* clock_mhz, double* ms_per_launch) try {
    if (!lane_mads_per_s || !clock_mhz || !ms_per_launch) return fail(ZKP_E_ARG, "null argument");
    if (launches == 0 || launches > 1000) return fail(ZKP_E_ARG, "launches must be in 1..1000");
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ctx().device));
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 8;
    ZCHK(ctx().tmp.ensure((size_t)blocks * 256 * 4 + sizeof(ClkRec)));
    uint32_t* out = reinterpret_cast<uint32_t*>(ctx().tmp.p);
    ClkRec* rec = reinterpret_cast<ClkRec*>(out + (size_t)blocks * 256);
    HIPCHK(hipMemsetAsync(rec, 0, sizeof(ClkRec), nullptr));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mad_rate_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, out, 7u, (ClkRec*)nullptr);  // warm-up
    HIPCHK(hipEventRecord(e0, nullptr));
    for (unsigned i = 0; i < launches; i++)
        hipLaunchKernelGGL(mad_rate_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, out, 9u + i, rec);
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ClkRec r;
    HIPCHK(hipMemcpy(&r, rec, sizeof r, hipMemcpyDeviceToHost));
    const double mads = (double)launches * blocks * 256.0 * MAD_PROBE_ITERS * MAD_PROBE_CHAINS;
    *lane_mads_per_s = mads / (ms * 1e-3);
    *clock_mhz = r.ref ? 100.0 * (double)r.cycles / (double)r.ref : 0.0;
    *ms_per_launch = ms / launches;
    return ZKP_OK;
} ZKP_CATCH_INT
// summed over the device slots (one slot unless zkp_init_devices was used)
int zkp_profile_read(const char* name, double* total_ms, uint64_t* count) try {
    if (!name || !total_ms || !count) return fail(ZKP_E_ARG, "null argument");
    std::lock_guard<std::mutex> g(g_rt.mu);
    double tot = 0;
    uint64_t cnt = 0;
    DeviceRestore restore;
    for (Ctx* c : g_rt.slots) {
        std::lock_guard<std::mutex> lk(c->mu);
        HIPCHK(hipSetDevice(c->device));
        for (ProfRec& r : c->prof) {
            if (std::strcmp(r.name, name) != 0) continue;
            if (r.a) {
                HIPCHK(hipEventSynchronize(r.b));
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
                tot += ms;
            } else {
                tot += r.host_ms;
            }
            cnt++;
        }
    }
    *total_ms = tot;
    *count = cnt;
    return ZKP_OK;
} ZKP_CATCH_INT
const char* zkp_last_error(void) { return g_err.c_str(); }

}  // extern "C"

namespace {

// A new slot on HIP device `device`; g_rt.mu held by the caller.
int create_slot_locked(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(ZKP_E_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(ZKP_E_ARG, "device index out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ZKP_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIPCHK(hipSetDevice(device));
    ZCHK(allow_big_lds((ntt_pass_strided<Fr, NttOps<Fr>::LOG_T>)));
    ZCHK(allow_big_lds((ntt_pass_strided<Fr, NttOps<Fr>::LOG_T - 1>)));
    ZCHK(allow_big_lds((ntt_pass_strided<Fr, NttOps<Fr>::LOG_T - 2>)));
    ZCHK(allow_big_lds((ntt_pass_strided<Gl, NttOps<Gl>::LOG_T>)));
    ZCHK(allow_big_lds(ntt_pass_last<Fr>));
    ZCHK(allow_big_lds(ntt_pass_last<Gl>));
    ZCHK(allow_big_lds(msm_partscatter_kernel<PS_TILE_BIG>));
    ZCHK(allow_big_lds(msm_partscatter_kernel<PS_TILE_MID>));
    ZCHK(allow_big_lds(msm_partscatter_kernel<PS_TILE_SMALL>));
    ZCHK(allow_big_lds(fri_tail_kernel));
    Ctx* c = new Ctx;
    c->device = device;
    c->slot = (int)g_rt.slots.size();
    {   // what the last-levels launch of the bucket reduction may assume resident: two waves per SIMD (four SIMDs per CU), and no
        // more than the kernel's own occupancy (registers, LDS) allows at its default workgroup size
        int per_cu = 0;
        uint32_t cap = (uint32_t)prop.multiProcessorCount * 4 * 2;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msm_pyramid_tail_kernel, (int)PYR_TAIL_THREADS, 0) == hipSuccess && per_cu > 0)
            cap = std::min<uint32_t>(cap, (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount * (PYR_TAIL_THREADS / 64));
        else
            (void)hipGetLastError();
        c->tail_max_waves = std::max<uint32_t>(cap, 4);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(ZKP_E_DEVICE, "hipStreamCreate failed");
    }
    g_rt.slots.push_back(c);
    return ZKP_OK;
}

void destroy_slot(Ctx* c) {
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (ProfRec& r : c->prof) {
        if (r.a && r.owns_a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
    for (int f = 0; f < 2; f++) {
        for (auto& kv : c->radix_tw[f]) (void)hipFree(kv.second);
        for (auto& kv : c->axis0_tw[f]) (void)hipFree(kv.second);
    }
    for (auto& kv : c->plans_fr) {
        (void)hipFree(kv.second.inter_lo); (void)hipFree(kv.second.inter_hi); (void)hipFree(kv.second.inter_lo_ninv);
        for (auto* d : kv.second.direct) (void)hipFree(d);
        for (auto* d : kv.second.tw_matrix) (void)hipFree(d);
    }
    for (auto& kv : c->plans_gl) {
        (void)hipFree(kv.second.inter_lo); (void)hipFree(kv.second.inter_hi); (void)hipFree(kv.second.inter_lo_ninv);
        for (auto* d : kv.second.direct) (void)hipFree(d);
    }
    for (int i = 0; i < Ctx::COSET_WAYS; i++) {
        if (c->coset_fr[i].lo) (void)hipFree(c->coset_fr[i].lo);
        if (c->coset_fr[i].hi) (void)hipFree(c->coset_fr[i].hi);
        if (c->coset_gl[i].lo) (void)hipFree(c->coset_gl[i].lo);
        if (c->coset_gl[i].hi) (void)hipFree(c->coset_gl[i].hi);
    }
    DevBuf* bufs[] = {&c->ntt_scratch, &c->scalars, &c->digits, &c->sorted, &c->entries, &c->counts, &c->start, &c->perm, &c->over,
                      &c->pieces, &c->buckets, &c->parts, &c->pyr1, &c->odd0, &c->odd1, &c->result, &c->fb_table, &c->tmp, &c->fri_arena,
                      &c->fri_meta, &c->clk, &c->xchg_a, &c->xchg_b};
    for (DevBuf* b : bufs) b->release();
    if (c->host_result) (void)hipHostFree(c->host_result);
    if (c->fri_small) (void)hipHostFree(c->fri_small);
    if (c->copy_event) (void)hipEventDestroy(c->copy_event);
    for (hipEvent_t e : c->copy_events) (void)hipEventDestroy(e);
    c->copy_events.clear();
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (hipEvent_t e : {c->ev_sort[0], c->ev_sort[1], c->ev_acc[0], c->ev_acc[1], c->ev_begin})
        if (e) (void)hipEventDestroy(e);
    if (c->sort_stream) (void)hipStreamDestroy(c->sort_stream);
    for (hipEvent_t e : c->xev)
        if (e) (void)hipEventDestroy(e);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->ws_event) (void)hipEventDestroy(c->ws_event);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

}  // namespace

extern "C" {

int zkp_init(int device) try {
    std::lock_guard<std::mutex> g(g_rt.mu);
    if (!g_rt.slots.empty()) return ZKP_OK;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    return create_slot_locked(device);
} ZKP_CATCH_INT

int zkp_init_devices(const int* devices, int n_devices) try {
    if (n_devices < 0 || n_devices > 64) return fail(ZKP_E_ARG, "n_devices out of range");
    std::lock_guard<std::mutex> g(g_rt.mu);
    std::vector<int> want;
    if (n_devices == 0 || !devices) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(ZKP_E_DEVICE, "no HIP device visible");
        const int n = n_devices ? n_devices : count;
        if (n > count) return fail(ZKP_E_ARG, "more devices requested than visible");
        for (int i = 0; i < n; i++) want.push_back(i);
    } else {
        want.assign(devices, devices + n_devices);
    }
    if (!g_rt.slots.empty()) {  // idempotent for the same list only
        bool same = g_rt.slots.size() == want.size();
        for (size_t i = 0; same && i < want.size(); i++) same = g_rt.slots[i]->device == want[i];
        return same ? ZKP_OK : fail(ZKP_E_ARG, "library already initialised with another device list (zkp_shutdown first)");
    }
    for (int d : want) {
        const int rc = create_slot_locked(d);
        if (rc != ZKP_OK) {
            for (Ctx* c : g_rt.slots) { destroy_slot(c); delete c; }
            g_rt.slots.clear();
            return rc;
        }
    }
    g_rt.multi = g_rt.slots.size() > 1;
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_device_count(void) {
    std::lock_guard<std::mutex> g(g_rt.mu);
    return (int)g_rt.slots.size();
}

int zkp_set_device(int slot) try {
    if (slot < -1) return fail(ZKP_E_ARG, "slot must be -1 (default) or a slot index");
    {
        std::lock_guard<std::mutex> g(g_rt.mu);
        if (slot >= 0 && !g_rt.slots.empty() && slot >= (int)g_rt.slots.size()) return fail(ZKP_E_ARG, "device slot out of range");
    }
    t_slot = slot;
    return ZKP_OK;
} ZKP_CATCH_INT

void zkp_shutdown(void) {
    device_workers_stop();
    std::lock_guard<std::mutex> g(g_rt.mu);
    for (Ctx* c : g_rt.slots) {
        destroy_slot(c);
        delete c;
    }
    g_rt.slots.clear();
    g_rt.multi = false;
}

// ---- bases -------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

int bases_alloc(size_t n, bool with_inf, zkp_bases** out) {
    zkp_bases* b = new (std::nothrow) zkp_bases();
    if (!b) return fail(ZKP_E_NOMEM, "host allocation failed");
    b->n = n;
    b->device = ctx().device;
    b->slot = ctx().slot;
    hipError_t e = hipMalloc(&b->d_xy, std::max<size_t>(128 * n, 128));
    if (e == hipSuccess && with_inf) e = hipMalloc(reinterpret_cast<void**>(&b->d_inf), std::max<size_t>(n, 1));
    if (e != hipSuccess) {
        if (b->d_xy) (void)hipFree(b->d_xy);
        delete b;
        return fail(e == hipErrorOutOfMemory ? ZKP_E_NOMEM : ZKP_E_DEVICE, hipGetErrorString(e));
    }
    *out = b;
    return ZKP_OK;
}

// contiguous chunk [lo, hi) of n items owned by shard i of k; sizes differ by at most one (zkp_hip/dist.py: shard_range)
void shard_range(size_t n, size_t i, size_t k, size_t* lo, size_t* hi) {
    const size_t base = n / k, rem = n % k;
    *lo = i * base + std::min(i, rem);
    *hi = *lo + base + (i < rem ? 1 : 0);
}

// n points from host memory onto ONE slot
int bases_create_single(int slot, const uint64_t* xy, const uint8_t* is_inf, size_t n, zkp_bases** out) {
    CTX_ENTER(slot);
    hipStream_t st = g_rt.multi ? ctx().stream : nullptr;
    WsOrder ord(st);
    zkp_bases* b = nullptr;
    ZCHK(bases_alloc(n, is_inf != nullptr, &b));
    hipError_t e = hipSuccess;
    if (n) {
        int rc = ctx().tmp.ensure(96 * n);
        if (rc != ZKP_OK) {
            zkp_g1_bases_destroy(b);
            return rc;
        }
        e = hipMemcpyAsync(ctx().tmp.p, xy, 96 * n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(g1_to_internal_kernel, dim3((unsigned)((n + MSM_THREADS - 1) / MSM_THREADS)),
                               dim3(MSM_THREADS), 0, st, reinterpret_cast<const uint4*>(ctx().tmp.p), (uint64_t)n,
                               reinterpret_cast<uint4*>(b->d_xy));
            e = hipGetLastError();
        }
        if (e == hipSuccess && is_inf) e = hipMemcpyAsync(b->d_inf, is_inf, n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        zkp_g1_bases_destroy(b);
        return fail(ZKP_E_DEVICE, hipGetErrorString(e));
    }
    *out = b;
    return ZKP_OK;
}

// Run fn(i) for every chunk of a sharded handle on the per-slot workers; the first failure (code + message) is the caller's.
int for_each_shard(const zkp_bases* b, const std::function<int(size_t)>& fn) {
    const size_t k = b->shards.size();
    std::vector<int> rc(k, ZKP_OK);
    std::vector<std::string> msg(k);
    std::vector<std::function<void()>> jobs(k);
    for (size_t i = 0; i < k; i++)
        jobs[i] = [&, i] {
            try {
                rc[i] = fn(i);
            } catch (...) {
                rc[i] = on_exception();
            }
            if (rc[i] != ZKP_OK) msg[i] = g_err;
        };
    g_workers.run(jobs);
    for (size_t i = 0; i < k; i++)
        if (rc[i] != ZKP_OK)  // (while a sharded handle is being created its chunk i is slot i and may still be null)
            return fail(rc[i], "device slot " + std::to_string(b->shards[i] ? b->shards[i]->slot : (int)i) + ": " + msg[i]);
    return ZKP_OK;
}

// Automatic window width of an expansion (0 = leave the bases as they are)
unsigned auto_window_bits(size_t n) {
    if (n < 64) return 0;
    return n >= (1u << 22) ? 22 : n >= (1u << 19) ? 20 : n > (1u << 13) ? 16 : n > (1u << 11) ? 14 : 12;
}

// check_only: stop after the argument and budget checks, before anything is allocated (the sharded entry asks every chunk's device
// first, so that a refusal leaves the whole handle unexpanded instead of a mixture)
int precompute_single(zkp_bases* b, unsigned window_bits, bool check_only = false) {
    if (window_bits == 0) {  // automatic
        // Up to 2^18 points the MSM is a chain of latencies, not of throughput: 16-bit windows (16 slices, 2^15 buckets, 14 reduction
        // levels) with the run of a bucket split over 2 or 4 lanes (msm.hpp, split_run) beat the 18..20 bits of round 1, whose 2^17..2^19
        // buckets were needed to fill the lanes and paid for it in the reduction (tools/split_sweep2.sh, one box, single MSM / batch of
        // three): 2^15 0.509 -> 0.463 / 0.870 -> 0.682 ms, 2^16 0.678 -> 0.529 / 1.112 -> 0.933, 2^17 0.922 -> 0.747 / 1.613 -> 1.454,
        // 2^18 1.121 -> 1.053 / 2.836 -> 2.566; 2^19 stays at 20 bits (1.62 ms against 1.80 at 18).  Below 2^13: 14 bits (2^12 0.312
        // against 0.324 ms), below 2^11: 12 bits (2^10 0.278 against 0.319 ms) -- profiles/r02_n_split_runs.md.
        // From 2^22 points 22 bits: 12 slices of 21/22 bits over 2^21 buckets -- one insertion per scalar less, a 4x larger
        // bucket reduction (0.46 -> 1.26 ms): 2^22 9.57 -> 9.10 ms, 2^24 37.5 -> 33.7 ms, 2^26 149.4 -> 132.5 ms
        // (profiles/r02_c_window22.md).
        if (b->pre_c || b->n < 64) return ZKP_OK;
        window_bits = auto_window_bits(b->n);
    }
    if (window_bits < 9 || window_bits > MSM_MAX_WINDOW_BITS) return fail(ZKP_E_ARG, "window_bits must be 0 (automatic) or in 9.." + std::to_string(MSM_MAX_WINDOW_BITS));
    if (b->pre_c) return b->pre_req == window_bits ? ZKP_OK : fail(ZKP_E_ARG, "bases already expanded with another width");
    CTX_ENTER(b->slot);
    hipStream_t st = g_rt.multi ? ctx().stream : nullptr;
    WsOrder ord(st);
    if (!b->n) return ZKP_OK;
    // Slices of a scalar: ceil(256 / window_bits) of them.  When that many windows of window_bits overshoot the 256 bits, the top
    // window is short by that many bits and its 2^-k of the buckets collect 2^k times the points of the others; the 256 bits are
    // then split into slices of floor/ceil(256 / planes) bits instead (18 -> 15 slices of 17/18 bits, 19 -> 14 of 18/19, 20 -> 13
    // of 19/20).  Round 3: from ANY overshoot on (rounds 1-2: from 8 bits) -- at 20 bits the 4-bit overshoot left 2^14 buckets
    // with ~90 entries against 26 on average at 2^20 points, and those 256 waves, dispatched first, were still walking their
    // runs alone when the rest of the machine had finished (profiles/r03_j_balanced_slices.md).
    const uint32_t planes = 256 / window_bits + (256 % window_bits ? 1 : 0);
    SliceOffsets so;
    std::memset(&so, 0, sizeof so);
    uint32_t cmax = window_bits;
    static const uint32_t balance_from = getenv("ZKP_MSM_BALANCE_FROM") ? (uint32_t)atoi(getenv("ZKP_MSM_BALANCE_FROM")) : 1u;  // tuning aid
    if (planes * window_bits - 256 < balance_from) {
        for (uint32_t s = 0; s <= planes; s++) so.off[s] = (uint16_t)(s * window_bits);
    } else {
        const uint32_t base = 256 / planes, rem = 256 % planes;
        cmax = base + (rem ? 1 : 0);
        for (uint32_t s = 0; s < planes; s++) so.off[s + 1] = (uint16_t)(so.off[s] + base + (s < rem ? 1 : 0));
    }
    void* p = nullptr;
    {   // The expansion is `planes` x the SRS (13 x at 20 bits, 12 x at 22: 103 GB for 2^26 points, and a 2^27 SRS no longer fits one
        // device).  Say so with the numbers instead of a bare allocation failure; the handle stays usable unexpanded (per-window MSM).
        const size_t need = 128 * (size_t)planes * b->n;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = total_b = 0, (void)hipGetLastError();
        size_t budget = free_b ? free_b : ~(size_t)0;
        if (const char* env = getenv("ZKP_SRS_EXPAND_MAX_BYTES")) budget = std::min<size_t>(budget, (size_t)strtoull(env, nullptr, 10));
        auto too_large = [&](const char* why) {
            return fail(ZKP_E_NOMEM, "SRS expansion does not fit (" + std::string(why) + "): " + std::to_string(planes) + " planes x " +
                                         std::to_string(b->n) + " points x 128 B = " + std::to_string(need) + " bytes, " +
                                         std::to_string(free_b) + " of " + std::to_string(total_b) +
                                         " bytes free on the device; the bases stay unexpanded (per-window MSM), or shard the SRS over "
                                         "more devices (zkp_init_devices)");
        };
        if (need > budget) return too_large(need > free_b && free_b ? "device memory" : "ZKP_SRS_EXPAND_MAX_BYTES");
        if (check_only) return ZKP_OK;
        if (hipMalloc(&p, need) != hipSuccess) {
            (void)hipGetLastError();
            return too_large("hipMalloc");
        }
    }
    hipError_t e = hipMemcpyAsync(p, b->d_xy, 128 * b->n, hipMemcpyDeviceToDevice, st);
    const uint64_t step = std::min<uint64_t>(b->n, 1ull << 18);  // points per launch: bounds the scratch area (ZZ, ZZZ, products)
    if (e == hipSuccess && ctx().tmp.ensure(192 * (size_t)planes * step) != ZKP_OK) e = hipErrorOutOfMemory;
    for (uint64_t off = 0; e == hipSuccess && off < b->n; off += step) {
        const uint64_t cnt = std::min<uint64_t>(step, b->n - off);
        hipLaunchKernelGGL(g1_expand_planes_kernel, dim3((unsigned)((cnt + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS),
                           0, st, reinterpret_cast<uint4*>(p), reinterpret_cast<uint4*>(ctx().tmp.p), off, cnt,
                           (uint64_t)b->n, planes, so);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(e == hipErrorOutOfMemory ? ZKP_E_NOMEM : ZKP_E_DEVICE, std::string("SRS expansion: ") + hipGetErrorString(e));
    }
    (void)hipFree(b->d_xy);
    b->d_xy = p;
    b->pre_c = cmax;
    b->pre_req = window_bits;
    b->pre_planes = planes;
    std::memcpy(b->pre_off, so.off, sizeof so.off);
    return ZKP_OK;
}

// Back to the plain points (plane 0 of an expansion is the points themselves): roll-back of a sharded expansion that failed half way
int unexpand_single(zkp_bases* b) {
    if (!b->pre_c) return ZKP_OK;
    CTX_ENTER(b->slot);
    hipStream_t st = g_rt.multi ? ctx().stream : nullptr;
    WsOrder ord(st);
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, 128 * std::max<size_t>(b->n, 1)));
    hipError_t e = hipMemcpyAsync(p, b->d_xy, 128 * b->n, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(ZKP_E_DEVICE, hipGetErrorString(e));
    }
    (void)hipFree(b->d_xy);
    b->d_xy = p;
    b->pre_c = b->pre_req = b->pre_planes = 0;
    return ZKP_OK;
}

}  // namespace

extern "C" {

int zkp_g1_bases_create(const uint64_t* xy, const uint8_t* is_inf, size_t n, zkp_bases** out) try {
    if (!out || (n && !xy)) return fail(ZKP_E_ARG, "null argument");
    size_t nslots = 0;
    {
        std::lock_guard<std::mutex> g(g_rt.mu);
        nslots = g_rt.slots.size();
    }
    if (nslots <= 1 || t_slot >= 0) return bases_create_single(t_slot >= 0 ? t_slot : 0, xy, is_inf, n, out);
    // several device slots and no zkp_set_device() choice on this thread: shard by contiguous chunk, one chunk per slot
    zkp_bases* c = new (std::nothrow) zkp_bases();
    if (!c) return fail(ZKP_E_NOMEM, "host allocation failed");
    c->n = n;
    c->slot = -1;
    c->shards.assign(nslots, nullptr);
    c->shard_off.assign(nslots, 0);
    for (size_t i = 0; i < nslots; i++) {
        size_t hi = 0;
        shard_range(n, i, nslots, &c->shard_off[i], &hi);
    }
    const int rc = for_each_shard(c, [&](size_t i) {  // (shards[i] is still null here: for_each_shard only reads its slot on failure)
        size_t lo = 0, hi = 0;
        shard_range(n, i, nslots, &lo, &hi);
        return bases_create_single((int)i, xy + 12 * lo, is_inf ? is_inf + lo : nullptr, hi - lo, &c->shards[i]);
    });
    if (rc != ZKP_OK) {
        zkp_g1_bases_destroy(c);
        return rc;
    }
    *out = c;
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_g1_bases_create_dev(const void* d_xy, const uint8_t* d_is_inf, size_t n, void* stream, zkp_bases** out) try {
    if (!out || (n && !d_xy)) return fail(ZKP_E_ARG, "null argument");
    // device memory belongs to one device: the handle lives on the slot of that device (the thread's zkp_set_device() slot when
    // it matches, else the first slot on the pointer's device)
    int slot = t_slot;
    if (n) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, d_xy) == hipSuccess) {
            std::lock_guard<std::mutex> g(g_rt.mu);
            if (slot >= 0 && slot < (int)g_rt.slots.size() && g_rt.slots[(size_t)slot]->device != attr.device) slot = -1;
            for (size_t i = 0; slot < 0 && i < g_rt.slots.size(); i++)
                if (g_rt.slots[i]->device == attr.device) slot = (int)i;
            if (slot < 0 && !g_rt.slots.empty()) return fail(ZKP_E_ARG, "device pointer belongs to a device the library was not initialised on");
        }
    }
    CTX_ENTER(slot);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WsOrder ord(st);
    zkp_bases* b = nullptr;
    ZCHK(bases_alloc(n, d_is_inf != nullptr, &b));
    hipError_t e = hipSuccess;
    if (n) {
        hipLaunchKernelGGL(g1_to_internal_kernel, dim3((unsigned)((n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS),
                           0, st, reinterpret_cast<const uint4*>(d_xy), (uint64_t)n, reinterpret_cast<uint4*>(b->d_xy));
        e = hipGetLastError();
    }
    if (e == hipSuccess && n && d_is_inf) e = hipMemcpyAsync(b->d_inf, d_is_inf, n, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        zkp_g1_bases_destroy(b);
        return fail(ZKP_E_DEVICE, hipGetErrorString(e));
    }
    *out = b;
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_g1_bases_precompute(zkp_bases* b, unsigned window_bits) try {
    if (!b) return fail(ZKP_E_ARG, "null argument");
    if (b->shards.empty()) return precompute_single(b, window_bits);
    // A sharded handle is expanded all-or-nothing and every chunk alike (zkp_g1_bases_info reports chunk 0 for all of them): the
    // automatic width comes from the largest chunk, every chunk's device is asked for room BEFORE any of them allocates, and a
    // failure half way rolls the finished chunks back to the plain points.
    size_t nmax = 0, expanded = 0;
    for (const zkp_bases* sh : b->shards) nmax = std::max(nmax, sh->n), expanded += sh->pre_c ? 1 : 0;
    if (window_bits == 0) {
        if (expanded == b->shards.size() || !(window_bits = auto_window_bits(nmax))) return ZKP_OK;
        if (expanded) window_bits = b->shards[0]->pre_req ? b->shards[0]->pre_req : window_bits;
    }
    ZCHK(for_each_shard(b, [&](size_t i) { return precompute_single(b->shards[i], window_bits, true); }));
    std::vector<uint8_t> was(b->shards.size());
    for (size_t i = 0; i < b->shards.size(); i++) was[i] = b->shards[i]->pre_c ? 1 : 0;
    const int rc = for_each_shard(b, [&](size_t i) { return precompute_single(b->shards[i], window_bits); });  // every chunk on its own device
    if (rc != ZKP_OK) {
        const std::string why = zkp_last_error();
        bool mixed = false;
        for (size_t i = 0; i < b->shards.size(); i++)
            if (!was[i] && b->shards[i]->pre_c && unexpand_single(b->shards[i]) != ZKP_OK) mixed = true;
        return fail(rc, why + (mixed ? " -- and a finished chunk could not be rolled back: the handle is expanded in part (still usable)"
                                     : " -- every chunk is back to the plain points"));
    }
    return ZKP_OK;
} ZKP_CATCH_INT

size_t zkp_g1_bases_len(const zkp_bases* b) { return b ? b->n : 0; }

int zkp_g1_bases_info(const zkp_bases* b, unsigned* window_bits, unsigned* slices) try {
    if (!b || !window_bits || !slices) return fail(ZKP_E_ARG, "null argument");
    const zkp_bases* s = b->shards.empty() ? b : b->shards[0];  // every chunk of a sharded handle is expanded alike
    if (!s) return fail(ZKP_E_ARG, "empty handle");
    *window_bits = s->pre_req;
    *slices = s->pre_c ? s->pre_planes : 0;
    return ZKP_OK;
} ZKP_CATCH_INT

void zkp_g1_bases_destroy(zkp_bases* b) {
    if (!b) return;
    for (zkp_bases* s : b->shards) zkp_g1_bases_destroy(s);
    if (b->d_xy || b->d_inf) {
        int prev = 0;
        const bool restore = hipGetDevice(&prev) == hipSuccess && prev != b->device;
        if (restore) (void)hipSetDevice(b->device);
        if (b->d_xy) (void)hipFree(b->d_xy);
        if (b->d_inf) (void)hipFree(b->d_inf);
        if (restore) (void)hipSetDevice(prev);
    }
    delete b;
}

// ---- MSM ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

const char* const kShardedDev = "bases are sharded over several devices: device-pointer entries take single-device bases "
                                "(zkp_set_device + zkp_g1_bases_create*); use zkp_msm_g1 with host scalars";

// sum_{i<n} scalars[i] * bases[i] for host scalars over ONE slot's bases, unnormalised
int msm_host_scalars(const zkp_bases* bases, const uint64_t* scalars, size_t n, HXyzz* r) {
    CTX_ENTER(bases->slot);
    hipStream_t st = g_rt.multi ? ctx().stream : nullptr;
    WsOrder ord(st);
    if (n > bases->n) return fail(ZKP_E_SIZE, "more scalars than bases (kzg/src/scheme.rs:86)");
    *r = HXyzz::infinity();
    if (!n) return ZKP_OK;
    ZCHK(ctx().scalars.ensure(32 * n));
    const Fr* d_sc = reinterpret_cast<const Fr*>(ctx().scalars.p);
    const bool shared = bases->pre_c != 0;
    if (shared && n >= (1u << 19)) {  // pipeline the PCIe upload against the kernels
        if (!ctx().copy_stream) {
            HIPCHK(hipStreamCreateWithFlags(&ctx().copy_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx().copy_event, hipEventDisableTiming));
        }
        MsmFeed feed{scalars, ctx().copy_stream, ctx().copy_event, 0, 0};
        // two ranges: the upload of the second hides behind the first one's kernels; more ranges cost more in accumulate
        // efficiency (shorter runs per bucket, one bucket read-modify-write per range) than the shorter exposed first upload
        // saves -- 2^20: 1 range 3.49 ms, 2: 3.36-3.41, 4: 3.46-3.51, 8: 3.93 (gpurun_out/pcie_ranges.txt, round 2).
        // Round 4: the two ranges need not be equal.  The first one's upload is exposed and both ranges pay a pass over the buckets,
        // so the first is made just long enough for its kernels to cover the upload of the rest (profiles/r04_i).
        uint64_t parts = 2;
        // Round 5, with the uploader thread (profiles/r05_o): 25 % + 75 % up to 2^21 terms; from there a THIRD range pays for its pass over
        // the buckets -- 10 % + 30 % + 60 %: the first upload is short, and 40 % of the insertions are done by the time the last upload ends
        // (2^24: 36.0 -> 34.0 ms, 2^22: 9.8 -> 9.4-9.7; 2^20: 2.77 -> 2.80, not used there).
        unsigned first_pct = n >= (1u << 21) ? 10 : 25;
        if (const char* e = getenv("ZKP_MSM_FEED_RANGES")) {  // equal ranges, as rounds 2-3 (tuning aid)
            const int v = atoi(e);
            if (v >= 1 && v <= 64) { parts = (uint64_t)v; first_pct = 0; }
        }
        if (const char* e = getenv("ZKP_MSM_FEED_FIRST_PCT")) {  // tuning aid: share of the scalars in the first range (0 = equal ranges)
            const int v = atoi(e);
            if (v >= 0 && v <= 90) first_pct = (unsigned)v;
        }
        unsigned second_pct = n >= (1u << 21) ? 30 : 0;
        if (const char* e = getenv("ZKP_MSM_FEED_SECOND_PCT")) {  // tuning aid: a second short range before the rest
            const int v = atoi(e);
            if (v >= 0 && v <= 80) second_pct = (unsigned)v;
        }
        if (first_pct) {
            feed.first_len = std::max<uint64_t>(1024, ((uint64_t)n * first_pct / 100) & ~(uint64_t)1023);
            if (second_pct) feed.second_len = std::max<uint64_t>(1024, ((uint64_t)n * second_pct / 100) & ~(uint64_t)1023);
            parts = 1;  // the rest in one piece (or as many as the range limit asks for)
        }
        while ((parts << feed.range_log) < n) feed.range_log++;
        return msm_partial_batch(bases, &d_sc, 1, n, st, r, &feed);
    }
    HIPCHK(hipMemcpyAsync(ctx().scalars.p, scalars, 32 * n, hipMemcpyHostToDevice, st));
    return msm_partial(bases, d_sc, n, st, r);
}

// the same over a handle that may be sharded: every device takes the scalars of its chunk (uploaded by its own worker thread
// over its own PCIe link) and runs the whole Pippenger on it; the per-device partial sums (192 B each) come back to the host
// with each device's result anyway, so the exchange of SURVEY 8e is a host-side EC add of `devices` points
int msm_host_scalars_any(const zkp_bases* bases, const uint64_t* scalars, size_t n, HXyzz* r) {
    if (bases->shards.empty()) return msm_host_scalars(bases, scalars, n, r);
    if (n > bases->n) return fail(ZKP_E_SIZE, "more scalars than bases (kzg/src/scheme.rs:86)");
    const size_t k = bases->shards.size();
    std::vector<HXyzz> part(k, HXyzz::infinity());
    ZCHK(for_each_shard(bases, [&](size_t i) {
        const size_t lo = bases->shard_off[i];
        if (lo >= n) return (int)ZKP_OK;
        const size_t len = std::min(bases->shards[i]->n, n - lo);
        return msm_host_scalars(bases->shards[i], scalars + 4 * lo, len, &part[i]);
    }));
    HXyzz acc = HXyzz::infinity();
    for (size_t i = 0; i < k; i++) acc = acc.add(part[i]);
    *r = acc;
    return ZKP_OK;
}

}  // namespace

extern "C" {

int zkp_msm_g1_partial_dev(const zkp_bases* bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xyzz[24]) try {
    if (!bases || !out_xyzz || (n && !d_scalars)) return fail(ZKP_E_ARG, "null argument");
    if (!bases->shards.empty()) return fail(ZKP_E_ARG, kShardedDev);
    CTX_ENTER(bases->slot);
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    HXyzz r;
    ZCHK(msm_partial(bases, reinterpret_cast<const Fr*>(d_scalars), n, reinterpret_cast<hipStream_t>(stream), &r));
    r.store(out_xyzz);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_msm_g1_dev(const zkp_bases* bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xy[12],
                   uint8_t* out_is_inf) try {
    if (!bases || !out_xy || !out_is_inf || (n && !d_scalars)) return fail(ZKP_E_ARG, "null argument");
    if (!bases->shards.empty()) return fail(ZKP_E_ARG, kShardedDev);
    CTX_ENTER(bases->slot);
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    HXyzz r;
    ZCHK(msm_partial(bases, reinterpret_cast<const Fr*>(d_scalars), n, reinterpret_cast<hipStream_t>(stream), &r));
    r.to_affine(out_xy, out_is_inf);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_msm_g1(const zkp_bases* bases, const uint64_t* scalars, size_t n, uint64_t out_xy[12], uint8_t* out_is_inf) try {
    if (!bases || !out_xy || !out_is_inf || (n && !scalars)) return fail(ZKP_E_ARG, "null argument");
    HXyzz r;
    ZCHK(msm_host_scalars_any(bases, scalars, n, &r));
    r.to_affine(out_xy, out_is_inf);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_g1_bases_shard_count(const zkp_bases* b) { return !b ? 0 : b->shards.empty() ? 1 : (int)b->shards.size(); }

int zkp_g1_bases_shard(const zkp_bases* b, size_t i, int* slot, int* device, size_t* offset, size_t* len) try {
    if (!b || !slot || !device || !offset || !len) return fail(ZKP_E_ARG, "null argument");
    if (i >= (size_t)zkp_g1_bases_shard_count(b)) return fail(ZKP_E_ARG, "chunk index out of range");
    const zkp_bases* s = b->shards.empty() ? b : b->shards[i];
    *slot = s->slot;
    *device = s->device;
    *offset = b->shards.empty() ? 0 : b->shard_off[i];
    *len = s->n;
    return ZKP_OK;
} ZKP_CATCH_INT

// Ordering of a chunk's launch after the producer of its scalars: the slot's stream (or the legacy null stream of a single-slot
// runtime) does not wait for work on the caller's non-blocking streams by itself.  With an event the wait happens on the device
// (hipStreamWaitEvent, the host does not block); without one the entry waits for the whole device.
static int order_after_producer(hipStream_t st, void* ready_event) {
    if (ready_event) HIPCHK(hipStreamWaitEvent(st, reinterpret_cast<hipEvent_t>(ready_event), 0));
    else HIPCHK(hipDeviceSynchronize());
    return ZKP_OK;
}

int zkp_msm_g1_sharded_dev_after(const zkp_bases* bases, const void* const* d_scalars, void* const* ready_events, size_t n,
                                 uint64_t out_xy[12], uint8_t* out_is_inf) try {
    if (!bases || !out_xy || !out_is_inf || (n && !d_scalars)) return fail(ZKP_E_ARG, "null argument");
    if (n > bases->n) return fail(ZKP_E_SIZE, "more scalars than bases (kzg/src/scheme.rs:86)");
    HXyzz acc = HXyzz::infinity();
    if (bases->shards.empty()) {
        if (n) {
            if (!d_scalars[0]) return fail(ZKP_E_ARG, "null scalar pointer");
            CTX_ENTER(bases->slot);
            hipStream_t st = g_rt.multi ? ctx().stream : nullptr;
            // neither the slot's non-blocking stream nor the legacy null stream is ordered after a producer on a non-blocking
            // stream (torch's side streams are): wait for its event, or for the device (include/zkp_hip.h states this contract)
            ZCHK(order_after_producer(st, ready_events ? ready_events[0] : nullptr));
            WsOrder ord(st);
            ZCHK(msm_partial(bases, reinterpret_cast<const Fr*>(d_scalars[0]), n, st, &acc));
        }
    } else {
        const size_t k = bases->shards.size();
        for (size_t i = 0; i < k; i++)
            if (bases->shard_off[i] < n && bases->shards[i]->n && !d_scalars[i]) return fail(ZKP_E_ARG, "null scalar pointer for a chunk in use");
        std::vector<HXyzz> part(k, HXyzz::infinity());
        ZCHK(for_each_shard(bases, [&](size_t i) {
            const size_t lo = bases->shard_off[i];
            const zkp_bases* sh = bases->shards[i];
            if (lo >= n || !sh->n) return (int)ZKP_OK;
            const size_t len = std::min(sh->n, n - lo);
            CTX_ENTER(sh->slot);
            // d_scalars[i] may come from a copy or kernel still in flight on another stream of this device (a resident tensor made
            // by .to(device) a moment ago): the slot's stream is non-blocking and would not wait for it
            ZCHK(order_after_producer(ctx().stream, ready_events ? ready_events[i] : nullptr));
            WsOrder ord(ctx().stream);
            return msm_partial(sh, reinterpret_cast<const Fr*>(d_scalars[i]), len, ctx().stream, &part[i]);
        }));
        for (size_t i = 0; i < k; i++) acc = acc.add(part[i]);
    }
    acc.to_affine(out_xy, out_is_inf);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_msm_g1_sharded_dev(const zkp_bases* bases, const void* const* d_scalars, size_t n, uint64_t out_xy[12],
                           uint8_t* out_is_inf) {
    return zkp_msm_g1_sharded_dev_after(bases, d_scalars, nullptr, n, out_xy, out_is_inf);
}

int zkp_msm_g1_partial(const zkp_bases* bases, const uint64_t* scalars, size_t n, uint64_t out_xyzz[24]) try {
    if (!bases || !out_xyzz || (n && !scalars)) return fail(ZKP_E_ARG, "null argument");
    HXyzz r;
    ZCHK(msm_host_scalars_any(bases, scalars, n, &r));
    r.store(out_xyzz);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_msm_g1_batch_dev(const zkp_bases* bases, const void* const* d_scalars, size_t count, size_t n, void* stream,
                         uint64_t* out_xy, uint8_t* out_is_inf) try {
    if (!bases || (count && (!d_scalars || !out_xy || !out_is_inf))) return fail(ZKP_E_ARG, "null argument");
    if (!bases->shards.empty()) return fail(ZKP_E_ARG, kShardedDev);
    CTX_ENTER(bases->slot);
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    std::vector<HXyzz> r(count);
    ZCHK(msm_partial_batch(bases, reinterpret_cast<const Fr* const*>(d_scalars), count, n, reinterpret_cast<hipStream_t>(stream),
                           r.data()));
    // affine results with ONE field inversion for the whole batch (Montgomery's trick over the finite ZZZ): a Fermat inversion
    // is ~23 us of host time, and a PLONK proof makes nine commitments in four batches
    std::vector<HFq> prefix(count);
    HFq run = HFq::one();
    for (size_t m = 0; m < count; m++) {
        prefix[m] = run;
        if (!r[m].is_inf()) run = run * r[m].zzz;
    }
    HFq inv = run.inverse();
    for (size_t m = count; m-- > 0;) {
        if (r[m].is_inf()) {
            std::memset(out_xy + 12 * m, 0, 96);
            out_is_inf[m] = 1;
            continue;
        }
        const HFq zi3 = inv * prefix[m];  // 1 / ZZZ_m
        inv = inv * r[m].zzz;
        HFq zi2 = zi3 * r[m].zz;          // ZZ / ZZZ = 1 / Z, squared below = 1 / ZZ
        zi2 = zi2.sqr();
        (r[m].x * zi2).store(out_xy + 12 * m);
        (r[m].y * zi3).store(out_xy + 12 * m + 6);
        out_is_inf[m] = 0;
    }
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_g1_xyzz_sum(const uint64_t* partials, size_t count, uint64_t out_xy[12], uint8_t* out_is_inf) try {
    if ((count && !partials) || !out_xy || !out_is_inf) return fail(ZKP_E_ARG, "null argument");
    HXyzz acc = HXyzz::infinity();
    for (size_t i = 0; i < count; i++) acc = acc.add(HXyzz::load(partials + 24 * i));
    acc.to_affine(out_xy, out_is_inf);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_kzg_commit(const zkp_bases* srs, const uint64_t* coeffs, size_t len, uint64_t out_xy[12], uint8_t* out_is_inf) try {
    if (!srs || !out_xy || !out_is_inf || (len && !coeffs)) return fail(ZKP_E_ARG, "null argument");
    KzgScheme scheme(srs);
    KzgCommitment cm;
    int rc = scheme.commit(coeffs, len, &cm);
    if (rc == ZKP_E_SIZE && g_err.empty()) g_err = "SRS shorter than the polynomial (kzg/src/scheme.rs:86)";
    if (rc != ZKP_OK) return rc;
    std::memcpy(out_xy, cm.p.xy, 96);
    *out_is_inf = cm.p.infinity;
    return ZKP_OK;
} ZKP_CATCH_INT

int kzg_open_device(const zkp_bases* srs, const uint64_t* coeffs, size_t len, const uint64_t z[4], uint64_t out_xy[12],
                    uint8_t* out_is_inf, uint64_t out_eval[4]);  // plonk_host.inc

int zkp_kzg_open(const zkp_bases* srs, const uint64_t* coeffs, size_t len, const uint64_t z[4], uint64_t out_xy[12],
                 uint8_t* out_is_inf, uint64_t out_eval[4]) try {
    if (!srs || !out_xy || !out_is_inf || !out_eval || !z || (len && !coeffs)) return fail(ZKP_E_ARG, "null argument");
    if (len == 0) return fail(ZKP_E_ARG, "open of an empty polynomial (kzg/src/scheme.rs:112 expects at least 1)");
    // long polynomials: Horner evaluation and the division by (X - z) run on the GPU too (they are O(n) serial loops in
    // the reference, scheme.rs:110-118, and would dwarf the MSM on the host); short ones use the C++ mirror as is
    if (len >= 4096) return kzg_open_device(srs, coeffs, len, z, out_xy, out_is_inf, out_eval);
    KzgScheme scheme(srs);
    KzgOpening op;
    int rc = scheme.open(coeffs, len, z, &op);
    if (rc != ZKP_OK) return rc;
    std::memcpy(out_xy, op.p.xy, 96);
    *out_is_inf = op.p.infinity;
    op.eval.store(out_eval);
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_g1_mul(const uint64_t base_xy[12], uint8_t base_is_inf, const uint64_t scalar[4], uint64_t out_xy[12],
               uint8_t* out_is_inf) try {
    if (!base_xy || !scalar || !out_xy || !out_is_inf) return fail(ZKP_E_ARG, "null argument");
    HFr k = HFr::load(scalar).from_mont();
    HXyzz r = HXyzz::from_affine(base_xy, base_is_inf != 0).mul(k.l);
    r.to_affine(out_xy, out_is_inf);
    return ZKP_OK;
} ZKP_CATCH_INT

static int fixed_base_mul_locked(const void* d_scalars, size_t n, void* d_out_xy, uint8_t* d_out_is_inf, hipStream_t st) {
    ZCHK(ensure_fixed_base_table(st));
    hipLaunchKernelGGL(g1_fixed_base_kernel, dim3((unsigned)((n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                       st, reinterpret_cast<const Fr*>(d_scalars), (uint64_t)n,
                       reinterpret_cast<const uint4*>(ctx().fb_table.p), reinterpret_cast<uint4*>(d_out_xy), d_out_is_inf);
    HIPCHK(hipGetLastError());
    return ZKP_OK;
}

int zkp_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, void* d_out_xy, uint8_t* d_out_is_inf, void* stream) try {
    if (n && (!d_scalars || !d_out_xy)) return fail(ZKP_E_ARG, "null argument");
    CTX_ENTER(-1);
    if (!n) return ZKP_OK;
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    return fixed_base_mul_locked(d_scalars, n, d_out_xy, d_out_is_inf, reinterpret_cast<hipStream_t>(stream));
} ZKP_CATCH_INT

int zkp_selftest_fq_inverse_dev(const void* d_in, size_t n, int form, void* d_out, void* stream) try {
    if (n && (!d_in || !d_out)) return fail(ZKP_E_ARG, "null argument");
    if (form != 0 && form != 1) return fail(ZKP_E_ARG, "form must be 0 (12 x u32, radix 2^384) or 1 (14 x 28 bit + 2 pad words, radix 2^392)");
    CTX_ENTER(-1);
    if (!n) return ZKP_OK;
    hipLaunchKernelGGL(fq_inverse_selftest_kernel, dim3((unsigned)((n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint32_t*>(d_in), reinterpret_cast<uint32_t*>(d_out),
                       (uint64_t)n, form);
    HIPCHK(hipGetLastError());
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_srs_g1(const uint64_t secret[4], size_t n, uint64_t* out_xy) try {
    if (!secret || (n && !out_xy)) return fail(ZKP_E_ARG, "null argument");
    if (!n) return ZKP_OK;
    std::vector<uint64_t> pw(4 * n);
    HFr s = HFr::load(secret), cur = HFr::one();
    for (size_t i = 0; i < n; i++) {  // cur *= secret, kzg/src/srs.rs:58
        cur.store(&pw[4 * i]);
        cur = cur * s;
    }
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    ZCHK(ctx().tmp.ensure(32 * n + 96 * n));
    char* d = reinterpret_cast<char*>(ctx().tmp.p);
    HIPCHK(hipMemcpy(d, pw.data(), 32 * n, hipMemcpyHostToDevice));
    ZCHK(fixed_base_mul_locked(d, n, d + 32 * n, nullptr, nullptr));
    HIPCHK(hipMemcpy(out_xy, d + 32 * n, 96 * n, hipMemcpyDeviceToHost));
    return ZKP_OK;
} ZKP_CATCH_INT

// ---- NTT ---------------------------------------------------------------------------------------------
}  // extern "C"
bool ntt_should_shard(unsigned log_n);                                                         // ntt_sharded.inc
int ntt_fr_sharded_host(uint64_t* data, unsigned log_n, int inverse, const uint64_t* coset);  // ntt_sharded.inc
extern "C" {
int zkp_ntt_fr(uint64_t* data, unsigned log_n, int inverse, const uint64_t* coset) try {
    // several device slots and a transform worth spreading: the four-step transform over all of them (natural order in and out)
    if (ntt_should_shard(log_n)) return ntt_fr_sharded_host(data, log_n, inverse, coset);
    return ntt_host_entry<Fr>(data, log_n, inverse, coset);
} ZKP_CATCH_INT
int zkp_ntt_goldilocks(uint64_t* data, unsigned log_n, int inverse, const uint64_t* coset) try {
    return ntt_host_entry<Gl>(data, log_n, inverse, coset);
} ZKP_CATCH_INT
int zkp_ntt_fr_dev(void* d_data, unsigned log_n, size_t batch, int inverse, const uint64_t* coset, void* stream) try {
    if (!d_data) return fail(ZKP_E_ARG, "data is null");
    CTX_ENTER(-1);
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    return run_ntt<Fr>(reinterpret_cast<Fr*>(d_data), log_n, batch, inverse, coset, reinterpret_cast<hipStream_t>(stream));
} ZKP_CATCH_INT
int zkp_ntt_goldilocks_dev(void* d_data, unsigned log_n, size_t batch, int inverse, const uint64_t* coset, void* stream) try {
    if (!d_data) return fail(ZKP_E_ARG, "data is null");
    CTX_ENTER(-1);
    WsOrder ord(reinterpret_cast<hipStream_t>(stream));
    return run_ntt<Gl>(reinterpret_cast<Gl*>(d_data), log_n, batch, inverse, coset, reinterpret_cast<hipStream_t>(stream));
} ZKP_CATCH_INT

int zkp_ntt_fr_twiddle_dev(void* d_data, size_t rows, size_t cols, size_t row0, unsigned log_n, int inverse, void* stream) try {
    if (!d_data) return fail(ZKP_E_ARG, "data is null");
    if (log_n > 32 || log_n == 0) return fail(ZKP_E_ARG, "log_n out of range");
    if ((uint64_t)(row0 + rows - 1) * (cols - 1) >= (1ull << log_n) && rows && cols)
        return fail(ZKP_E_ARG, "twiddle exponent (row0 + rows - 1) * (cols - 1) must stay below n");
    CTX_ENTER(-1);
    if (!rows || !cols) return ZKP_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WsOrder ord(st);
    PowTab<Fr> tab;
    HFr w = fr_root_of_unity(log_n);  // get_coset_tables inverts the base itself when inverse != 0
    ZCHK(get_coset_tables<Fr>(log_n, inverse ? 1 : 0, w.l, HFr::one(), &tab, st));
    const uint64_t total = (uint64_t)rows * cols;
    hipLaunchKernelGGL(twiddle_rows_kernel<Fr>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<Fr*>(d_data), (uint64_t)rows, (uint64_t)cols, (uint64_t)row0, tab);
    HIPCHK(hipGetLastError());
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_ntt_fr_axis0_dev(const void* d_in, void* d_out, unsigned log_len, size_t cols, int inverse, unsigned tw_log_n,
                         size_t tw_col0, void* stream) try {
    if (!d_in || !d_out) return fail(ZKP_E_ARG, "data is null");
    CTX_ENTER(-1);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WsOrder ord(st);
    return run_ntt_axis0<Fr>(reinterpret_cast<const Fr*>(d_in), reinterpret_cast<Fr*>(d_out), log_len, cols, inverse, tw_log_n,
                             (uint64_t)tw_col0, st);
} ZKP_CATCH_INT

int zkp_ntt_fr_layout_dev(const void* d_in, void* d_out, unsigned log_n, size_t batch, int inverse, const zkp_ntt_layout* in_layout,
                          const zkp_ntt_layout* out_layout, unsigned tw_log_n, size_t tw_row0, void* stream) try {
    if (!d_in || !d_out) return fail(ZKP_E_ARG, "data is null");
    if (log_n > 32) return fail(ZKP_E_ARG, "log_n > 32");
    NttRemap rin, rout;
    const zkp_ntt_layout* ls[2] = {in_layout, out_layout};
    NttRemap* rs[2] = {&rin, &rout};
    for (int i = 0; i < 2; i++) {
        std::memset(rs[i], 0, sizeof(NttRemap));
        if (!ls[i]) continue;
        if (ls[i]->lo_bits + ls[i]->mid_bits > log_n || ls[i]->lo_bits < 2)
            return fail(ZKP_E_ARG, "layout: lo_bits must be >= 2 (16-byte runs of four elements) and lo_bits + mid_bits <= log_n");
        rs[i]->on = 1;
        rs[i]->lo_bits = ls[i]->lo_bits;
        rs[i]->mid_bits = ls[i]->mid_bits;
        rs[i]->mid_stride = ls[i]->mid_stride;
        rs[i]->hi_stride = ls[i]->hi_stride;
        rs[i]->batch_stride = ls[i]->batch_stride;
    }
    if (d_in == d_out && (in_layout || out_layout))
        return fail(ZKP_E_ARG, "a transform with a gathered or scattered layout cannot run in place");
    CTX_ENTER(-1);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WsOrder ord(st);
    NttIo io;
    io.in_remap = in_layout ? &rin : nullptr;
    io.out_remap = out_layout ? &rout : nullptr;
    io.tw_log_n = tw_log_n;
    io.tw_row0 = (uint64_t)tw_row0;
    return run_ntt<Fr>(reinterpret_cast<const Fr*>(d_in), reinterpret_cast<Fr*>(d_out), log_n, batch, inverse, nullptr, st, &io);
} ZKP_CATCH_INT

int zkp_fri_layer_eval(const uint64_t* coeffs, size_t d, uint64_t coset, unsigned log_D, uint64_t* out) try {
    if ((d && !coeffs) || !out) return fail(ZKP_E_ARG, "null argument");
    if (log_D > 32) return fail(ZKP_E_ARG, "log_D > 32");
    const size_t D = (size_t)1 << log_D;
    if (d > D) return fail(ZKP_E_ARG, "more coefficients than domain points");
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    ZCHK(ctx().tmp.ensure(8 * D));
    HIPCHK(hipMemsetAsync(ctx().tmp.p, 0, 8 * D, nullptr));
    if (d) HIPCHK(hipMemcpyAsync(ctx().tmp.p, coeffs, 8 * d, hipMemcpyHostToDevice, nullptr));
    ZCHK(run_ntt<Gl>(reinterpret_cast<Gl*>(ctx().tmp.p), log_D, 1, 0, &coset, nullptr));
    HIPCHK(hipMemcpyAsync(out, ctx().tmp.p, 8 * D, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return ZKP_OK;
} ZKP_CATCH_INT

}  // extern "C"

namespace {
// out[j] = c[2j] + r * c[2j+1]; r canonical, c Montgomery residues (plain product keeps the residue form)
__global__ void fri_fold_kernel(const uint64_t* c, uint64_t d, uint64_t r, uint64_t* out) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * j >= d) return;
    Gl v{c[2 * j]};
    if (2 * j + 1 < d) v = v + Gl{r} * Gl{c[2 * j + 1]};
    out[j] = v.v;
}
}  // namespace

extern "C" {

int zkp_fri_fold(const uint64_t* coeffs, size_t d, uint64_t r, uint64_t* out) try {
    if (d && (!coeffs || !out)) return fail(ZKP_E_ARG, "null argument");
    if (!d) return ZKP_OK;
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    const size_t m = (d + 1) / 2;
    ZCHK(ctx().tmp.ensure(8 * d + 8 * m));
    uint64_t* dc = reinterpret_cast<uint64_t*>(ctx().tmp.p);
    HIPCHK(hipMemcpyAsync(dc, coeffs, 8 * d, hipMemcpyHostToDevice, nullptr));
    HGl rr = HGl::load(&r).from_mont();
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nullptr, dc, (uint64_t)d, rr.l[0],
                       dc + d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dc + d, 8 * m, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return ZKP_OK;
} ZKP_CATCH_INT

int zkp_poly_mul_fr(const uint64_t* a, size_t la, const uint64_t* b, size_t lb, uint64_t* out) try {
    if (la == 0 || lb == 0) return ZKP_OK;  // zero operand => zero polynomial (no coefficients)
    if (!a || !b || !out) return fail(ZKP_E_ARG, "null argument");
    const size_t lo = la + lb - 1;
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < lo) log_n++;
    const size_t n = (size_t)1 << log_n;
    CTX_ENTER(-1);
    WsOrder ord(nullptr);
    ZCHK(ctx().tmp.ensure(2 * 32 * n));
    char* d = reinterpret_cast<char*>(ctx().tmp.p);
    HIPCHK(hipMemsetAsync(d, 0, 2 * 32 * n, nullptr));
    HIPCHK(hipMemcpyAsync(d, a, 32 * la, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(d + 32 * n, b, 32 * lb, hipMemcpyHostToDevice, nullptr));
    ZCHK(run_ntt<Fr>(reinterpret_cast<Fr*>(d), log_n, 2, 0, nullptr, nullptr));
    hipLaunchKernelGGL(pointwise_mul_kernel<Fr>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                       reinterpret_cast<const Fr*>(d), reinterpret_cast<const Fr*>(d + 32 * n), reinterpret_cast<Fr*>(d),
                       (uint64_t)n);
    HIPCHK(hipGetLastError());
    ZCHK(run_ntt<Fr>(reinterpret_cast<Fr*>(d), log_n, 1, 1, nullptr, nullptr));
    HIPCHK(hipMemcpyAsync(out, d, 32 * lo, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return ZKP_OK;
} ZKP_CATCH_INT

}  // extern "C"

#include "ntt_sharded.inc"
#include "plonk_host.inc"
#include "fri_host.inc"
#include "verify_host.inc"

// lrm_capi.cpp -- the extern "C" boundary of liblrm.so (include/lrm.h).
#include "../../include/lrm.h"
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>
#include "lrm_compile.h"
#include "lrm_launch.h"
#include "lrm_point.h"
#include "lrm_point_fast.h"
#include "lrm_point_tol.h"
#include "lrm_point_xtab.h"

namespace {

thread_local std::string g_err;
// Process-wide, as documented in lrm.h; atomic so that a helper thread (the host pipeline's) or a second caller thread reads a whole value.
std::atomic<int> g_mode{LRM_MODE_FAST}; // bit-identical to LRM_MODE_STRICT (tests/test_gpu_parity.py runs both)
bool tol_mode() { const int m = g_mode; return m == LRM_MODE_TOL || m == LRM_MODE_TOL_REL; }
uint32_t tol_flags() { return g_mode == LRM_MODE_TOL_REL ? LRM_TOLF_SHORT : 0u; }
const float kQuatTest[4] = {1.f, 0.f, 0.f, 0.f}; // settings.h:51

int fail(int code, const char* what) {
    g_err = what;
    return code;
}
int hip_fail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? LRM_ENOMEM : LRM_ENODEV;
}
#define HIP_TRY(expr, where)                              \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return hip_fail(e_, where); \
    } while (0)

const float* quat_or_default(const float* q) { return q ? q : kQuatTest; }

// ---- table-guided modes: plumbing ---------------------------------------------------------------
// One lock around the process-wide caches below (table cache with LRU eviction, queue workspaces, the statistic of the last
// call): host threads may call the entry points concurrently; a table is never freed under a launch that is being queued.
std::recursive_mutex g_cache_mu;
// The tolerance block of a (leg, quaternion) costs a few hundred microseconds of host geometry (arc
// intersections + their verification, lrm_compile_tol): a small cache keyed by the 18 input floats.
struct TolKey {
    float v[18];
    bool operator<(const TolKey& o) const { return std::memcmp(v, o.v, sizeof v) < 0; }
};
struct TolEntry {
    LrmTolLeg tl;
    LrmXtabLeg xl;                 // the strict value chain's constants (lrm_point_xtab.h)
    std::vector<uint8_t> tab;      // the plane table (lrm_build_tol_tab), built on first use
    int tab_state = 0;             // 0 not built yet, 1 built, -1 this leg has none (too many rows)
    float tab_build_ms = 0.f;      // what building it cost (host wall clock, or the device build's events)
    std::map<int, uint8_t*> tab_dev;
    uint64_t last_use = 0;
};
uint64_t g_tol_clock = 0;
std::map<TolKey, TolEntry> g_tol_cache;
float g_last_tab_build_ms = -1.f; // lrm_dbg_last_table_build_ms
TolEntry& tol_entry(const LrmLegDimensions& leg, const float* quat, const LrmCompiledLeg& L) {
    TolKey k;
    std::memcpy(k.v, &leg, 14 * sizeof(float));
    std::memcpy(k.v + 14, quat, 4 * sizeof(float));
    auto it = g_tol_cache.find(k);
    if (it != g_tol_cache.end()) {
        it->second.last_use = ++g_tol_clock;
        return it->second;
    }
    if (g_tol_cache.size() >= 64) { // a sweep over many orientations: the least recently used pair goes (hipFree waits for the device)
        auto old = g_tol_cache.begin();
        for (auto jt = g_tol_cache.begin(); jt != g_tol_cache.end(); ++jt)
            if (jt->second.last_use < old->second.last_use) old = jt;
        for (auto& d : old->second.tab_dev) (void)hipFree(d.second);
        g_tol_cache.erase(old);
    }
    TolEntry& e = g_tol_cache[k];
    e.last_use = ++g_tol_clock;
    lrm_compile_tol(L, &e.tl);
    lrm_make_xtab_leg(L, e.tl, &e.xl);
    return e;
}
// where the last tolerance-mode launch left its per-segment doubt counts, for lrm_dbg_tol_queue_counts
struct TolLast {
    int dev = -1;
    const uint32_t* counts = nullptr; // device: one count per queue segment of the main kernel
    size_t blocks = 0, n = 0;         // segments, points
    uint32_t cap = 0;                 // slots per segment
} g_tol_last;
// Device workspace of the doubt queues (rewritten by every call), one per (device, stream) in use, grown on demand.  Streams come
// and go (a caller rotating streams; lrm_reach_dist_multi's per-call streams drop theirs explicitly): the map keeps the
// kMaxWorkspaces most recently used entries, and an entry is looked up by the stream's address only while that stream lives.
struct TolWorkspace {
    uint32_t* p = nullptr;
    size_t words = 0;
    uint64_t last_use = 0;
};
constexpr size_t kMaxWorkspaces = 16;
std::map<std::pair<int, void*>, TolWorkspace> g_tol_ws;
void tol_workspace_free(TolWorkspace& w) {
    if (w.p && g_tol_last.counts == w.p) g_tol_last = TolLast{}; // the statistic of the last call goes with its workspace
    if (w.p) (void)hipFree(w.p); // synchronises with whatever still uses it
    w = TolWorkspace{};
}
// the workspace of a stream that is about to be destroyed (call on its device, after the stream has drained)
void tol_workspace_drop(int dev, void* stream) {
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    auto it = g_tol_ws.find(std::make_pair(dev, stream));
    if (it == g_tol_ws.end()) return;
    tol_workspace_free(it->second);
    g_tol_ws.erase(it);
}
int tol_workspace(size_t words, void* stream, uint32_t** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    const auto key = std::make_pair(dev, stream);
    if (g_tol_ws.find(key) == g_tol_ws.end() && g_tol_ws.size() >= kMaxWorkspaces) { // the least recently used entry goes
        auto old = g_tol_ws.begin();
        for (auto jt = g_tol_ws.begin(); jt != g_tol_ws.end(); ++jt)
            if (jt->second.last_use < old->second.last_use) old = jt;
        int cur = dev;
        (void)hipSetDevice(old->first.first);
        tol_workspace_free(old->second);
        (void)hipSetDevice(cur);
        g_tol_ws.erase(old);
    }
    TolWorkspace& w = g_tol_ws[key];
    w.last_use = ++g_tol_clock;
    if (words > w.words) {
        tol_workspace_free(w);
        w.last_use = g_tol_clock;
        void* p = nullptr;
        const size_t want = words + words / 4;
        HIP_TRY(hipMalloc(&p, want * sizeof(uint32_t)), "hipMalloc tolerance-mode queues");
        w.p = static_cast<uint32_t*>(p);
        w.words = want;
    }
    *out = w.p;
    return LRM_OK;
}
// The table kernels (dist_tab_kernel, dist_xtab_kernel) are the default of their modes from LRM_TOLTAB_MIN_POINTS points on
// (below, a call is launch-bound and the single launch of the kernels without a table wins); LRM_TOL_TABLE=0 in the
// environment keeps the kernels without a table (A/B runs).
#ifndef LRM_TOLTAB_MIN_POINTS
#define LRM_TOLTAB_MIN_POINTS 200000
#endif
// LRM_TOL_TABLE: "0" never, "2" for every size (tests), anything else / unset: from LRM_TOLTAB_MIN_POINTS points on
bool tol_tab_wanted(size_t n) {
    const char* e = std::getenv("LRM_TOL_TABLE");
    if (e && e[0] == '0') return false;
    if (e && e[0] == '2') return n > 0;
    return n >= (size_t)LRM_TOLTAB_MIN_POINTS;
}
// LRM_MODE_FAST takes the table-guided bit-exact kernel (dist_xtab_kernel) wherever the tolerance mode would take its table
// kernel; LRM_XTAB=0 keeps the filtered kernel of rounds 1-3 (A/B runs, and the reference the new kernel is tested against).
bool xtab_wanted(size_t n) {
    const char* e = std::getenv("LRM_XTAB");
    if (e && e[0] == '0') return false;
    return tol_tab_wanted(n);
}
// The table of E on the current device, or null (no table for this leg); rc != LRM_OK on a HIP error.  Built by the device
// builder (lrm_toltab_dev.hip, on `stream`: < 1 ms) unless LRM_TOLTAB_HOST=1 asks for the host builder or the device builder
// declines the leg (the host builder's ~30 ms + an upload); the two give the same bytes (tests/test_gpu_toltab.py).
bool toltab_host_wanted() {
    const char* e = std::getenv("LRM_TOLTAB_HOST");
    return e && e[0] == '1';
}
int tol_tab_device(TolEntry& E, void* stream, const uint8_t** out) {
    *out = nullptr;
    if (E.tab_state < 0) return LRM_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    uint8_t*& td = E.tab_dev[dev];
    if (!td) {
        bool host = toltab_host_wanted() || !E.tab.empty();
        if (!host) {
            uint8_t* t = nullptr;
            size_t bytes = 0;
            float ms = 0.f;
            const int rc = lrm_build_tol_tab_dev(E.tl, (hipStream_t)stream, &t, &bytes, &ms);
            if (rc < 0) return hip_fail((hipError_t)(-rc), "plane table (device build)");
            if (rc == 1) { E.tab_state = -1; return LRM_OK; }
            if (rc == 0) {
                td = t;
                E.tab_state = 1;
                E.tab_build_ms = ms;
                g_last_tab_build_ms = ms;
            } else host = true; // the device builder declines this leg
        }
        if (host) {
            if (E.tab.empty()) {
                const auto t0 = std::chrono::steady_clock::now();
                E.tab_state = lrm_build_tol_tab(E.tl, &E.tab) ? 1 : -1;
                E.tab_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                g_last_tab_build_ms = E.tab_build_ms;
                if (E.tab_state < 0) { E.tab.clear(); return LRM_OK; }
            }
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, E.tab.size()), "hipMalloc plane table");
            td = static_cast<uint8_t*>(p);
            HIP_TRY(hipMemcpy(td, E.tab.data(), E.tab.size(), hipMemcpyHostToDevice), "hipMemcpy plane table");
        }
    }
    *out = td;
    return LRM_OK;
}
// distance / fused launch in the current mode, SoA (xyz_aos null) or the float3 arrays of the apply_kernel boundary (xyz_aos set:
// x, y, z, bits, dy, dz unused, dx = the float3 output).  op: 1 distance, 2 reach + distance
int launch_dist_any(int op, const float* xyz_aos, const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions& leg,
                    const float* quat, const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                    float* dz, void* stream) {
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    const int mode = g_mode;
    hipStream_t st = (hipStream_t)stream;
    const bool tolm = mode == LRM_MODE_TOL || mode == LRM_MODE_TOL_REL;
    // 32-bit point indices in the table kernels (n + one grid stride < 2^32; LRM_MODE_TOL_REL keeps bit 31 of a queue record for a flag)
    const bool fits = n < (mode == LRM_MODE_TOL_REL ? 0x80000000ull : 0xc0000000ull);
    if ((tolm || (mode == LRM_MODE_FAST && xtab_wanted(n))) && L.fast_ok && fits) {
        TolEntry& E = tol_entry(leg, quat, L);
        const LrmTolLeg& TL = E.tl;
        if (TL.tol_ok) {
            const uint8_t* tab = nullptr;
            if (tol_tab_wanted(n)) {
                const int rc = tol_tab_device(E, stream, &tab);
                if (rc != LRM_OK) return rc;
            }
            if (tolm || tab) {
                uint32_t* w = nullptr;
                const int rc = tol_workspace(tab ? lrm_tol_tab_queue_words(n) : lrm_tol_queue_words(n), stream, &w);
                if (rc != LRM_OK) return rc;
                if (!tolm) {
                    if (xyz_aos) HIP_TRY(lrm_launch_dist_xtab_aos(op, xyz_aos, n, L, E.xl, tab, mask, dx, w, st), "table-guided bit-exact launch");
                    else HIP_TRY(lrm_launch_dist_xtab(op, x, y, z, n, L, E.xl, tab, mask, bits, dx, dy, dz, w, st), "table-guided bit-exact launch");
                } else if (tab) {
                    if (xyz_aos) HIP_TRY(lrm_launch_dist_tab_aos(op, xyz_aos, n, L, TL, E.xl, tab, mask, dx, w, tol_flags(), st), "tolerance-mode (table) launch");
                    else HIP_TRY(lrm_launch_dist_tab(op, x, y, z, n, L, TL, E.xl, tab, mask, bits, dx, dy, dz, w, tol_flags(), st), "tolerance-mode (table) launch");
                } else {
                    if (xyz_aos) HIP_TRY(lrm_launch_dist_tol_aos(op, xyz_aos, n, L, TL, mask, dx, w, tol_flags(), st), "tolerance-mode launch");
                    else HIP_TRY(lrm_launch_dist_tol(op, x, y, z, n, L, TL, mask, bits, dx, dy, dz, w, tol_flags(), st), "tolerance-mode launch");
                }
                int dev = 0;
                (void)hipGetDevice(&dev);
                // the workspace starts with one count per segment
                if (tab) g_tol_last = TolLast{dev, w, lrm_tol_tab_segments(n, mode == LRM_MODE_TOL_REL), n, (uint32_t)LRM_TOL_TAB_SEG_CAP};
                else g_tol_last = TolLast{dev, w, lrm_tol_queue_words(n) / (4 * LRM_TOL_SEG_CAP_WORDS + 4), n, (uint32_t)LRM_TOL_SEG_CAP_WORDS};
                return LRM_OK;
            }
        }
    }
    g_tol_last = TolLast{}; // this call ran the kernels without a table and without a queue: no queue statistic
    if (xyz_aos) HIP_TRY(lrm_launch_dist_aos(op, xyz_aos, n, L, mask, dx, mode != LRM_MODE_STRICT, st), "Kernel launch");
    else HIP_TRY(lrm_launch_dist_soa(op, x, y, z, n, L, mask, bits, dx, dy, dz, mode != LRM_MODE_STRICT, st), "distance launch");
    return LRM_OK;
}
int launch_dist_mode(int op, const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions& leg,
                     const float* quat, const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                     float* dz, void* stream) {
    return launch_dist_any(op, nullptr, x, y, z, n, leg, quat, L, mask, bits, dx, dy, dz, stream);
}
// distance / fused launch on the float3 arrays of the apply_kernel boundary in the current mode
int launch_dist_aos_mode(int op, const float* xyz, size_t n, const LrmLegDimensions& leg, const float* quat, const LrmCompiledLeg& L,
                         uint8_t* mask, float* dxyz, void* stream) {
    return launch_dist_any(op, xyz, nullptr, nullptr, nullptr, n, leg, quat, L, mask, nullptr, dxyz, nullptr, nullptr, stream);
}

// RAII device buffer for the host-buffer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// ---- pipelined host boundary (opt-in: LRM_HOST_PIPELINE=1) ---------------------------------------------------------
// apply_kernel (cross_compiled.cu:41-77) allocates, copies the whole input, runs the kernel, copies the whole output and
// frees, one after the other; for 1e7 fused points that is 16 ms of which 0.1-0.3 ms are kernels.  Where the time goes on
// this box (tools/pcie_probe.cpp, profiles/r03_pcie_probe.txt): a copy from / to memory the runtime has seen before runs
// at 56 GB/s, but the caller's output arrays are fresh (`new float3[n]`, bench.cpp:125-131): a copy INTO never-touched
// pages runs at 10 GB/s (one page fault per 4 KB, taken by the single thread that stages the copy) -- 13 of the 16 ms.
// With the pipeline the call keeps its device buffers, a few pinned staging slots and a handful of helper threads per
// device, and cuts the cloud into chunks (LRM_HOST_PIPELINE_CHUNK points, default 2^19: 6 MB in, 6.5 MB out; 2^20: 5.8 ms instead of 4.7):
//     kIn threads:   caller's input chunk -> pinned slot (memcpy) -> device (async DMA)
//     caller thread: the kernels of a chunk as soon as its input has landed
//     kOut threads:  device -> pinned slot (async DMA) -> caller's output arrays (memcpy: the page faults of the fresh
//                    arrays are taken by several threads at once, and overlap the DMA of the other chunks)
// Results are the same bytes as without the pipeline (same kernels on sub-ranges; the tolerance mode's fix-up is per
// launch anyway).  Returned ms = the sum of the chunks' kernel times (HIP events on the compute stream).
// Three streams in all (compute, H2D, D2H): ROCm maps streams onto a handful of hardware queues, and with one stream per
// helper thread a copy ended up queued behind another stream's event wait -- 6 ms stalls in one run out of two.  The
// helper threads of a direction share its stream and wait on their own events.
constexpr int kPipeIn = 4, kPipeOut = 4;
struct HostPipe {
    hipStream_t s_k = nullptr, s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_slot_in[kPipeIn] = {}, ev_slot_out[kPipeOut] = {};
    void *d_in = nullptr, *d_mask = nullptr, *d_out = nullptr;
    size_t cap = 0;        // points the device buffers hold
    size_t slot_chunk = 0; // points a pinned slot holds
    void *pin_in[kPipeIn] = {}, *pin_out[kPipeOut] = {}, *pin_mask[kPipeOut] = {};
    std::vector<hipEvent_t> ev_in, ev_k0, ev_k1;
};
std::map<int, HostPipe> g_host_pipe; // per device ordinal; lives until lrm_release_workspaces()
size_t host_pipeline_chunk() {
    const char* e = std::getenv("LRM_HOST_PIPELINE_CHUNK");
    const long v = e ? std::atol(e) : 0;
    return v >= 4096 ? ((size_t)v + 63) & ~(size_t)63 : (size_t)1 << 19;
}
bool host_pipeline_enabled() {
    const char* e = std::getenv("LRM_HOST_PIPELINE");
    return e && e[0] == '1';
}
void host_pipe_free(HostPipe& P) {
    for (void* p : {P.d_in, P.d_mask, P.d_out})
        if (p) (void)hipFree(p);
    for (int k = 0; k < kPipeIn; k++) {
        if (P.pin_in[k]) (void)hipHostFree(P.pin_in[k]);
        if (P.ev_slot_in[k]) (void)hipEventDestroy(P.ev_slot_in[k]);
    }
    for (int k = 0; k < kPipeOut; k++) {
        if (P.pin_out[k]) (void)hipHostFree(P.pin_out[k]);
        if (P.pin_mask[k]) (void)hipHostFree(P.pin_mask[k]);
        if (P.ev_slot_out[k]) (void)hipEventDestroy(P.ev_slot_out[k]);
    }
    for (hipStream_t st : {P.s_k, P.s_in, P.s_out})
        if (st) (void)hipStreamDestroy(st);
    for (auto* v : {&P.ev_in, &P.ev_k0, &P.ev_k1})
        for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    P = HostPipe{};
}
int host_apply_pipelined(int op, const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
                         uint8_t* mask_out, float* dxyz_out, float* ms, const LrmCompiledLeg& L) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    HostPipe& P = g_host_pipe[dev];
    const bool want_mask = (op != 1) || mask_out;
    const size_t chunk = host_pipeline_chunk();
    if (!P.s_k) {
        HIP_TRY(hipStreamCreateWithFlags(&P.s_k, hipStreamNonBlocking), "hipStreamCreate");
        HIP_TRY(hipStreamCreateWithFlags(&P.s_in, hipStreamNonBlocking), "hipStreamCreate");
        HIP_TRY(hipStreamCreateWithFlags(&P.s_out, hipStreamNonBlocking), "hipStreamCreate");
        for (int k = 0; k < kPipeIn; k++) HIP_TRY(hipEventCreateWithFlags(&P.ev_slot_in[k], hipEventDisableTiming), "hipEventCreate");
        for (int k = 0; k < kPipeOut; k++) HIP_TRY(hipEventCreateWithFlags(&P.ev_slot_out[k], hipEventDisableTiming), "hipEventCreate");
    }
    if (chunk > P.slot_chunk) {
        for (int k = 0; k < kPipeIn; k++) {
            if (P.pin_in[k]) (void)hipHostFree(P.pin_in[k]);
            P.pin_in[k] = nullptr;
        }
        for (int k = 0; k < kPipeOut; k++) {
            if (P.pin_out[k]) (void)hipHostFree(P.pin_out[k]);
            if (P.pin_mask[k]) (void)hipHostFree(P.pin_mask[k]);
            P.pin_out[k] = P.pin_mask[k] = nullptr;
        }
        P.slot_chunk = 0;
        for (int k = 0; k < kPipeIn; k++) HIP_TRY(hipHostMalloc(&P.pin_in[k], chunk * 3 * sizeof(float), hipHostMallocDefault), "hipHostMalloc staging");
        for (int k = 0; k < kPipeOut; k++) {
            HIP_TRY(hipHostMalloc(&P.pin_out[k], chunk * 3 * sizeof(float), hipHostMallocDefault), "hipHostMalloc staging");
            HIP_TRY(hipHostMalloc(&P.pin_mask[k], chunk, hipHostMallocDefault), "hipHostMalloc staging");
        }
        P.slot_chunk = chunk;
    }
    if (n > P.cap) {
        for (void** p : {&P.d_in, &P.d_mask, &P.d_out}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        P.cap = 0;
        HIP_TRY(hipMalloc(&P.d_in, n * 3 * sizeof(float)), "hipMalloc gpu_in.elements");
        HIP_TRY(hipMalloc(&P.d_mask, n), "hipMalloc gpu_out.elements");
        HIP_TRY(hipMalloc(&P.d_out, n * 3 * sizeof(float)), "hipMalloc gpu_out.elements");
        P.cap = n;
    }
    const size_t nchunks = (n + chunk - 1) / chunk;
    while (P.ev_in.size() < nchunks) {
        hipEvent_t a = nullptr, b = nullptr, c = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
        HIP_TRY(hipEventCreate(&b), "hipEventCreate");
        HIP_TRY(hipEventCreate(&c), "hipEventCreate");
        P.ev_in.push_back(a);
        P.ev_k0.push_back(b);
        P.ev_k1.push_back(c);
    }
    float* d_in = static_cast<float*>(P.d_in);
    uint8_t* d_mask = static_cast<uint8_t*>(P.d_mask);
    float* d_out = static_cast<float*>(P.d_out);
    // hand-over between the host threads: in_ready[c] = chunk c's "input landed" event is recorded; k_issued = chunks whose
    // kernels are queued (their "kernels done" event recorded)
    std::mutex mu, mu_in, mu_out;
    std::condition_variable cv;
    std::vector<char> in_ready(nchunks, 0);
    size_t k_issued = 0;
    hipError_t err_copy = hipSuccess;
    bool abort = false;
    auto fail_copy = [&](hipError_t e) {
        std::lock_guard<std::mutex> g(mu);
        if (err_copy == hipSuccess) err_copy = e;
        abort = true;
        cv.notify_all();
    };
    const bool dbg = std::getenv("LRM_HOST_PIPELINE_DEBUG") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<std::thread> helpers;
    for (int k = 0; k < kPipeIn; k++)
        helpers.emplace_back([&, k] {
            hipError_t e = hipSetDevice(dev);
            for (size_t c = (size_t)k; c < nchunks && e == hipSuccess; c += kPipeIn) {
                { std::lock_guard<std::mutex> g(mu); if (abort) return; }
                const size_t lo = c * chunk, cnt = std::min(chunk, n - lo);
                const double t0 = since();
                std::memcpy(P.pin_in[k], xyz + 3 * lo, cnt * 3 * sizeof(float));
                const double t1 = since();
                {   // one thread at a time queues on the shared stream: copy + its two events stay together
                    std::lock_guard<std::mutex> q(mu_in);
                    e = hipMemcpyAsync(d_in + 3 * lo, P.pin_in[k], cnt * 3 * sizeof(float), hipMemcpyHostToDevice, P.s_in);
                    if (e == hipSuccess) e = hipEventRecord(P.ev_in[c], P.s_in);
                    if (e == hipSuccess) e = hipEventRecord(P.ev_slot_in[k], P.s_in);
                }
                if (dbg) std::fprintf(stderr, "  in[%d] chunk %zu: memcpy %.2f-%.2f ms\n", k, c, t0, t1);
                if (e == hipSuccess) {
                    std::lock_guard<std::mutex> g(mu);
                    in_ready[c] = 1;
                    cv.notify_all();
                }
                if (e == hipSuccess) e = hipEventSynchronize(P.ev_slot_in[k]); // the slot is free again
            }
            if (e != hipSuccess) fail_copy(e);
        });
    for (int k = 0; k < kPipeOut; k++)
        helpers.emplace_back([&, k] {
            hipError_t e = hipSetDevice(dev);
            for (size_t c = (size_t)k; c < nchunks && e == hipSuccess; c += kPipeOut) {
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return k_issued > c || abort; });
                    if (abort) return;
                }
                const size_t lo = c * chunk, cnt = std::min(chunk, n - lo);
                const bool mask_here = want_mask && mask_out, field_here = op != 0;
                {
                    std::lock_guard<std::mutex> q(mu_out);
                    e = hipStreamWaitEvent(P.s_out, P.ev_k1[c], 0);
                    if (e == hipSuccess && mask_here) e = hipMemcpyAsync(P.pin_mask[k], d_mask + lo, cnt, hipMemcpyDeviceToHost, P.s_out);
                    if (e == hipSuccess && field_here) e = hipMemcpyAsync(P.pin_out[k], d_out + 3 * lo, cnt * 3 * sizeof(float), hipMemcpyDeviceToHost, P.s_out);
                    if (e == hipSuccess) e = hipEventRecord(P.ev_slot_out[k], P.s_out);
                }
                const double t0 = since();
                if (e == hipSuccess) e = hipEventSynchronize(P.ev_slot_out[k]);
                const double t1 = since();
                if (e == hipSuccess) {
                    if (mask_here) std::memcpy(mask_out + lo, P.pin_mask[k], cnt);
                    if (field_here) std::memcpy(dxyz_out + 3 * lo, P.pin_out[k], cnt * 3 * sizeof(float));
                }
                if (dbg) std::fprintf(stderr, "  out[%d] chunk %zu: D2H wait %.2f-%.2f, memcpy to the caller's arrays until %.2f ms\n", k, c, t0, t1, since());
            }
            if (e != hipSuccess) fail_copy(e);
        });
    int rc = LRM_OK;
    hipError_t err_k = hipSuccess;
    for (size_t c = 0; c < nchunks; c++) {
        {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return in_ready[c] || abort; });
            if (abort) break;
        }
        const size_t lo = c * chunk, cnt = std::min(chunk, n - lo);
        err_k = hipStreamWaitEvent(P.s_k, P.ev_in[c], 0);
        if (err_k == hipSuccess) err_k = hipEventRecord(P.ev_k0[c], P.s_k);
        if (err_k == hipSuccess) {
            if (op == 0) err_k = lrm_launch_reach_aos(d_in + 3 * lo, cnt, L, d_mask + lo, g_mode != LRM_MODE_STRICT, P.s_k);
            else rc = launch_dist_aos_mode(op, d_in + 3 * lo, cnt, *leg, quat_or_default(quat), L, want_mask ? d_mask + lo : nullptr, d_out + 3 * lo, P.s_k);
        }
        if (err_k == hipSuccess && rc == LRM_OK) err_k = hipEventRecord(P.ev_k1[c], P.s_k);
        std::lock_guard<std::mutex> g(mu);
        if (err_k != hipSuccess || rc != LRM_OK) abort = true;
        else k_issued = c + 1;
        cv.notify_all();
        if (abort) break;
    }
    const double t_launched = since();
    for (auto& t : helpers) t.join();
    if (dbg) std::fprintf(stderr, "host pipeline: %zu chunks of %zu points; kernels queued after %.2f ms, helpers done after %.2f ms\n", nchunks, chunk, t_launched, since());
    if (rc != LRM_OK) return rc;
    HIP_TRY(err_k, "Kernel launch");
    HIP_TRY(err_copy, "hipMemcpy (pipelined)");
    HIP_TRY(hipStreamSynchronize(P.s_k), "hipStreamSynchronize");
    float total = 0.f;
    for (size_t c = 0; c < nchunks; c++) {
        float e = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, P.ev_k0[c], P.ev_k1[c]), "hipEventElapsedTime");
        total += e;
    }
    if (ms) *ms = total;
    return LRM_OK;
}

// host-buffer skeleton of apply_kernel (cross_compiled.cu:33-79)
// op: 0 reach, 1 dist, 2 reach+dist
int host_apply(int op, const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
               uint8_t* mask_out, float* dxyz_out, float* ms) {
    if (!leg || (n && !xyz)) return fail(LRM_EINVAL, "null input");
    if ((op == 0 || op == 2) && n && !mask_out) return fail(LRM_EINVAL, "null mask output");
    if ((op == 1 || op == 2) && n && !dxyz_out) return fail(LRM_EINVAL, "null distance output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    if (host_pipeline_enabled() && n >= 2 * host_pipeline_chunk())
        return host_apply_pipelined(op, xyz, n, leg, quat, mask_out, dxyz_out, ms, L);
    DevBuf d_in, d_mask, d_out;
    HIP_TRY(d_in.alloc(n * 3 * sizeof(float)), "hipMalloc gpu_in.elements");
    const bool want_mask = (op != 1) || mask_out;
    if (want_mask) HIP_TRY(d_mask.alloc(n), "hipMalloc gpu_out.elements");
    if (op != 0) HIP_TRY(d_out.alloc(n * 3 * sizeof(float)), "hipMalloc gpu_out.elements");
    HIP_TRY(hipMemcpy(d_in.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy gpu_in.elements");
    HIP_TRY(lrm_launch_warmup(n, nullptr), "warm-up launch");
    Events ev;
    HIP_TRY(hipEventCreate(&ev.a), "hipEventCreate");
    HIP_TRY(hipEventCreate(&ev.b), "hipEventCreate");
    HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
    if (n) {
        if (op == 0) HIP_TRY(lrm_launch_reach_aos(d_in.as<float>(), n, L, d_mask.as<uint8_t>(), g_mode != LRM_MODE_STRICT, nullptr), "Kernel launch");
        else {
            const int rc = launch_dist_aos_mode(op, d_in.as<float>(), n, *leg, quat_or_default(quat), L, want_mask ? d_mask.as<uint8_t>() : nullptr,
                                                d_out.as<float>(), nullptr);
            if (rc != LRM_OK) return rc;
        }
    }
    HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
    HIP_TRY(hipEventSynchronize(ev.b), "Kernel launch");
    float elapsed = 0.f;
    HIP_TRY(hipEventElapsedTime(&elapsed, ev.a, ev.b), "hipEventElapsedTime");
    if (want_mask && mask_out) HIP_TRY(hipMemcpy(mask_out, d_mask.p, n, hipMemcpyDeviceToHost), "hipMemcpy gpu_out.elements");
    if (op != 0) HIP_TRY(hipMemcpy(dxyz_out, d_out.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy gpu_out.elements");
    HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (ms) *ms = elapsed;
    return LRM_OK;
}

} // namespace

namespace {
// Morton (Z-curve) order of an AoS cloud: indices sorted by the interleaved 10-bit cell
// coordinates of the bounding box.  Consecutive points are then spatially compact in ANY
// orientation, which is what the tile bounding-box skipping of the pair kernels needs (a raster
// row stops being a thin box as soon as the cloud is yawed).  "Any target" / per-body results do
// not depend on the order, so sorting changes no output.
std::vector<size_t> morton_order(const float* aos, const std::vector<size_t>& idx) {
    const size_t n = idx.size();
    std::vector<size_t> order(idx);
    if (n < 2) return order;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (size_t i : idx)
        for (int a = 0; a < 3; a++) {
            const float v = aos[3 * i + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    auto spread = [](uint32_t v) { // 10 bits -> every third bit
        v &= 0x3ffu;
        v = (v | (v << 16)) & 0x030000ffu;
        v = (v | (v << 8)) & 0x0300f00fu;
        v = (v | (v << 4)) & 0x030c30c3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    std::vector<uint32_t> key(n), key2(n);
    for (size_t k = 0; k < n; k++) {
        uint32_t kk = 0;
        for (int a = 0; a < 3; a++) {
            const float span = hi[a] - lo[a];
            float t = span > 0.f ? (aos[3 * idx[k] + a] - lo[a]) / span : 0.f;
            if (!(t >= 0.f)) t = 0.f; // nan / below
            if (t > 1.f) t = 1.f;
            kk |= spread((uint32_t)(t * 1023.f)) << a;
        }
        key[k] = kk;
    }
    // stable LSD radix sort on the 30-bit key, three 10-bit passes (equal keys keep the order of `idx`):
    // lrm_morton_order on 1e5 points: 2.3 ms against 5.5 ms with std::sort on (key, index) pairs
    std::vector<size_t> order2(n);
    for (int pass = 0; pass < 3; pass++) {
        size_t count[1025] = {0};
        const int shift = 10 * pass;
        for (size_t k = 0; k < n; k++) count[((key[k] >> shift) & 1023u) + 1]++;
        for (int b = 0; b < 1024; b++) count[b + 1] += count[b];
        for (size_t k = 0; k < n; k++) {
            const size_t dst = count[(key[k] >> shift) & 1023u]++;
            key2[dst] = key[k];
            order2[dst] = order[k];
        }
        key.swap(key2);
        order.swap(order2);
    }
    return order;
}
} // namespace

// Morton order for other translation units (lrm_octree.hip)
void lrm_host_morton_order(const float* xyz_aos, size_t n, std::vector<size_t>* order) {
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    *order = morton_order(xyz_aos, idx);
}

extern "C" {

const char* lrm_version(void) { return "lrm-mi355x 0.1 (gfx950)"; }
const char* lrm_last_error(void) { return g_err.c_str(); }

int lrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int lrm_set_device(int ordinal) {
    HIP_TRY(hipSetDevice(ordinal), "hipSetDevice");
    return LRM_OK;
}
int lrm_set_mode(int mode) {
    if (mode != LRM_MODE_STRICT && mode != LRM_MODE_FAST && mode != LRM_MODE_TOL && mode != LRM_MODE_TOL_REL) return fail(LRM_EINVAL, "unknown mode");
    g_mode = mode;
    return LRM_OK;
}
int lrm_get_mode(void) { return g_mode; }

void lrm_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia, float tibia2femur,
                     float femur2tip, float coxa_angle_deg, float femur_angle_deg, float tibia_angle_deg,
                     float tib_abs_pos, float tib_abs_neg, LrmLegDimensions* out) {
    lrm_host_leg_factory(azimut, body2coxa, coxa_pitch_deg, coxa2tibia, tibia2femur, femur2tip, coxa_angle_deg,
                         femur_angle_deg, tibia_angle_deg, tib_abs_pos, tib_abs_neg, out);
}
void lrm_get_M2_leg(float azimut, LrmLegDimensions* out) {
    lrm_host_leg_factory(azimut, 181, -45, 65.5f, 129, 135, 60.0f, 90.0f, 120.0f, -5, -5, out);
}
void lrm_get_moonbot_leg(float azimut, LrmLegDimensions* out) {
    lrm_host_leg_factory(azimut, 181, 0, 65.5f, 129, 160, 60.0f, 90.0f, 120.0f, -5, -5, out);
}
void lrm_rotate_leg_data(const float quat[4], const LrmLegDimensions* leg, LrmLegDimensions* out) {
    lrm_host_rotate_leg_data(quat, *leg, out);
}

int lrm_reach(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
              float* ms) {
    return host_apply(0, xyz, n, leg, quat, mask_out, nullptr, ms);
}
int lrm_dist(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, float* dxyz_out,
             uint8_t* valid_out, float* ms) {
    return host_apply(1, xyz, n, leg, quat, valid_out, dxyz_out, ms);
}
int lrm_reach_dist(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                   float* dxyz_out, float* ms) {
    return host_apply(2, xyz, n, leg, quat, mask_out, dxyz_out, ms);
}

// Host buffers in the reference's ON-DISK layout (one f32 array per component,
// several_leg.cpp:126-131, :201-219): straight to the SoA kernels, no AoS detour
// (threeArrays2float3Arr, math_util.cpp:92, disappears).  op: 0 reach, 1 dist, 2 both.
namespace {
int host_apply_soa(int op, const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                   const float* quat, uint8_t* mask_out, float* dx, float* dy, float* dz, float* ms) {
    if (!leg || (n && (!x || !y || !z))) return fail(LRM_EINVAL, "null input");
    if ((op == 0 || op == 2) && n && !mask_out) return fail(LRM_EINVAL, "null mask output");
    if ((op == 1 || op == 2) && n && (!dx || !dy || !dz)) return fail(LRM_EINVAL, "null distance output");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
    if (n == 0) { if (ms) *ms = 0.f; return LRM_OK; }
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    const size_t pad = (n + 3) & ~(size_t)3; // keeps the three component arrays 16-byte aligned
    DevBuf d_in, d_mask, d_out;
    HIP_TRY(d_in.alloc(3 * pad * sizeof(float)), "hipMalloc gpu_in.elements");
    const bool want_mask = (op != 1) || mask_out;
    if (want_mask) HIP_TRY(d_mask.alloc(n), "hipMalloc gpu_out.elements");
    if (op != 0) HIP_TRY(d_out.alloc(3 * pad * sizeof(float)), "hipMalloc gpu_out.elements");
    float* I = d_in.as<float>();
    HIP_TRY(hipMemcpy(I, x, n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy x");
    HIP_TRY(hipMemcpy(I + pad, y, n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy y");
    HIP_TRY(hipMemcpy(I + 2 * pad, z, n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy z");
    HIP_TRY(lrm_launch_warmup(n, nullptr), "warm-up launch");
    Events ev;
    HIP_TRY(hipEventCreate(&ev.a), "hipEventCreate");
    HIP_TRY(hipEventCreate(&ev.b), "hipEventCreate");
    HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
    const bool fast = g_mode != LRM_MODE_STRICT;
    float* O = d_out.as<float>();
    if (op == 0) HIP_TRY(lrm_launch_reach_soa(I, I + pad, I + 2 * pad, n, L, d_mask.as<uint8_t>(), nullptr, fast, nullptr), "Kernel launch");
    else {
        const int rc = launch_dist_mode(op, I, I + pad, I + 2 * pad, n, *leg, quat_or_default(quat), L,
                                        want_mask ? d_mask.as<uint8_t>() : nullptr, nullptr, O, O + pad, O + 2 * pad, nullptr);
        if (rc != LRM_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
    HIP_TRY(hipEventSynchronize(ev.b), "Kernel launch");
    float elapsed = 0.f;
    HIP_TRY(hipEventElapsedTime(&elapsed, ev.a, ev.b), "hipEventElapsedTime");
    if (want_mask && mask_out) HIP_TRY(hipMemcpy(mask_out, d_mask.p, n, hipMemcpyDeviceToHost), "hipMemcpy mask");
    if (op != 0) {
        HIP_TRY(hipMemcpy(dx, O, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy dx");
        HIP_TRY(hipMemcpy(dy, O + pad, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy dy");
        HIP_TRY(hipMemcpy(dz, O + 2 * pad, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy dz");
    }
    if (ms) *ms = elapsed;
    return LRM_OK;
}
} // namespace

int lrm_reach_soa(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                  const float* quat, uint8_t* mask_out, float* ms) {
    return host_apply_soa(0, x, y, z, n, leg, quat, mask_out, nullptr, nullptr, nullptr, ms);
}
int lrm_dist_soa(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                 const float* quat, float* dx, float* dy, float* dz, uint8_t* valid_out, float* ms) {
    return host_apply_soa(1, x, y, z, n, leg, quat, valid_out, dx, dy, dz, ms);
}

// apply_reach_cpu / apply_dist_cpu (cross_compiled.cu:163-181): chrono-timed serial loops
int lrm_reach_cpu(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                  double* ms) {
    if (!leg || (n && (!xyz || !mask_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (size_t i = 0; i < n; i++)
        mask_out[i] = lrm_reach_global(L, &L.lists[0][0], LrmVec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (ms) *ms = std::chrono::duration<double>(t1 - t0).count() * 1000.0;
    return LRM_OK;
}
int lrm_dist_cpu(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, float* dxyz_out,
                 uint8_t* valid_out, double* ms) {
    if (!leg || (n && (!xyz || !dxyz_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        const bool v = lrm_dist_global(L, &L.lists[0][0], p);
        dxyz_out[3 * i] = p.x;
        dxyz_out[3 * i + 1] = p.y;
        dxyz_out[3 * i + 2] = p.z;
        if (valid_out) valid_out[i] = v;
    }
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (ms) *ms = std::chrono::duration<double>(t1 - t0).count() * 1000.0;
    return LRM_OK;
}

// ---- device-resident entry points ------------------------------------------------------
int lrm_reach_bits_dev(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                       const float* quat, uint8_t* mask, uint64_t* bits, void* stream) {
    if (!leg || (n && (!x || !y || !z || (!mask && !bits)))) return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    HIP_TRY(lrm_launch_reach_soa(x, y, z, n, L, mask, bits, g_mode != LRM_MODE_STRICT, (hipStream_t)stream), "reach launch");
    return LRM_OK;
}
int lrm_reach_dev(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                  const float* quat, uint8_t* mask, void* stream) {
    return lrm_reach_bits_dev(x, y, z, n, leg, quat, mask, nullptr, stream);
}
int lrm_dist_dev(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                 const float* quat, float* dx, float* dy, float* dz, uint8_t* valid, void* stream) {
    if (!leg || (n && (!x || !y || !z || !dx || !dy || !dz))) return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    return launch_dist_mode(1, x, y, z, n, *leg, quat_or_default(quat), L, valid, nullptr, dx, dy, dz, stream);
}
int lrm_reach_dist_bits_dev(const float* x, const float* y, const float* z, size_t n,
                            const LrmLegDimensions* leg, const float* quat, uint8_t* mask, uint64_t* bits,
                            float* dx, float* dy, float* dz, void* stream) {
    if (!leg || (n && (!x || !y || !z || (!mask && !bits) || !dx || !dy || !dz)))
        return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    return launch_dist_mode(2, x, y, z, n, *leg, quat_or_default(quat), L, mask, bits, dx, dy, dz, stream);
}
int lrm_reach_dist_dev(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                       const float* quat, uint8_t* mask, float* dx, float* dy, float* dz, void* stream) {
    if (n && !mask) return fail(LRM_EINVAL, "null argument");
    return lrm_reach_dist_bits_dev(x, y, z, n, leg, quat, mask, nullptr, dx, dy, dz, stream);
}
int lrm_reach_aos_dev(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask,
                      void* stream) {
    if (!leg || (n && (!xyz || !mask))) return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    HIP_TRY(lrm_launch_reach_aos(xyz, n, L, mask, g_mode != LRM_MODE_STRICT, (hipStream_t)stream), "reach launch");
    return LRM_OK;
}
int lrm_dist_aos_dev(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, float* dxyz,
                     uint8_t* valid, void* stream) {
    if (!leg || (n && (!xyz || !dxyz))) return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    return launch_dist_aos_mode(1, xyz, n, *leg, quat_or_default(quat), L, valid, dxyz, stream);
}

// ---- body x target aggregation ---------------------------------------------------------
namespace {
// Library-owned device workspaces of the pair kernels, ONE SET PER DEVICE (a process that walks over several GPUs
// keeps them all; nothing is leaked or overwritten on a device switch):
//   boxes  the bounding boxes of the target cloud (6 floats per 1024-target tile and per 64-target chunk), grown on
//          demand: the first call for a larger cloud allocates (not capturable in a graph), later calls only launch;
//   legs   kLegSlots slots of LRM_MAX_LEGS compiled legs for in-flight launches, handed out round-robin; a slot is
//          rewritten only after the event recorded behind the launch that last read it has completed, so launches
//          queued on several streams never see each other's legs.
constexpr int kLegSlots = 16;
struct DevicePool {
    float* boxes = nullptr;
    size_t boxes_cap = 0; // tiles
    LrmCompiledLeg* legs = nullptr;
    hipEvent_t slot_done[kLegSlots] = {};
    unsigned next_slot = 0;
};
std::map<int, DevicePool> g_pools;
int device_pool(DevicePool** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    *out = &g_pools[dev];
    return LRM_OK;
}
int tile_boxes(size_t nt, float** out) {
    const size_t ntiles = (nt + 1023) / 1024 * 17; // one box per tile + 16 chunk boxes per tile
    DevicePool* P = nullptr;
    int rc = device_pool(&P);
    if (rc != LRM_OK) return rc;
    if (ntiles > P->boxes_cap) {
        if (P->boxes) (void)hipFree(P->boxes); // synchronises with launches that still read it
        P->boxes = nullptr;
        P->boxes_cap = 0;
        void* p = nullptr;
        const size_t cap = ntiles + ntiles / 2 + 16;
        HIP_TRY(hipMalloc(&p, cap * 6 * sizeof(float)), "hipMalloc tile boxes");
        P->boxes = static_cast<float*>(p);
        P->boxes_cap = cap;
    }
    *out = P->boxes;
    return LRM_OK;
}
// a free slot for the legs of one launch; leg_slot_used() records the event behind that launch
int leg_slot(LrmCompiledLeg** out, hipEvent_t** done) {
    DevicePool* P = nullptr;
    int rc = device_pool(&P);
    if (rc != LRM_OK) return rc;
    if (!P->legs) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(LrmCompiledLeg) * LRM_MAX_LEGS * kLegSlots), "hipMalloc leg pool");
        P->legs = static_cast<LrmCompiledLeg*>(p);
    }
    const unsigned slot = P->next_slot++ % kLegSlots;
    if (P->slot_done[slot]) HIP_TRY(hipEventSynchronize(P->slot_done[slot]), "hipEventSynchronize leg slot");
    else HIP_TRY(hipEventCreateWithFlags(&P->slot_done[slot], hipEventDisableTiming), "hipEventCreate leg slot");
    *out = P->legs + (size_t)slot * LRM_MAX_LEGS;
    *done = &P->slot_done[slot];
    return LRM_OK;
}
} // namespace

namespace {
int reach_any_impl(const float* bx, const float* by, const float* bz, size_t nb, const float* tx, const float* ty,
                   const float* tz, size_t nt, const LrmLegDimensions* legs, size_t nlegs, const float* quat,
                   const uint8_t* body_active, bool boxes_ready, uint8_t* out_leg_body, uint8_t* all_legs_out, void* stream);
int any_in_shape_impl(int shape, const float* cx, const float* cy, const float* cz, size_t nc, const float* tx,
                      const float* ty, const float* tz, size_t nt, float radius, float plus_z, float minus_z,
                      bool boxes_ready, uint8_t* out, void* stream);
}

int lrm_reach_any_dev(const float* bx, const float* by, const float* bz, size_t nb, const float* tx,
                      const float* ty, const float* tz, size_t nt, const LrmLegDimensions* legs, size_t nlegs,
                      const float* quat, uint8_t* out_leg_body, uint8_t* all_legs_out, void* stream) {
    return reach_any_impl(bx, by, bz, nb, tx, ty, tz, nt, legs, nlegs, quat, nullptr, false, out_leg_body, all_legs_out, stream);
}

namespace {
int reach_any_impl(const float* bx, const float* by, const float* bz, size_t nb, const float* tx, const float* ty,
                   const float* tz, size_t nt, const LrmLegDimensions* legs, size_t nlegs, const float* quat,
                   const uint8_t* body_active, bool boxes_ready, uint8_t* out_leg_body, uint8_t* all_legs_out, void* stream) {
    if (!legs || nlegs == 0 || nlegs > LRM_MAX_LEGS) return fail(LRM_EINVAL, "nlegs must be 1..LRM_MAX_LEGS");
    if (!out_leg_body || (nb && (!bx || !by || !bz)) || (nt && (!tx || !ty || !tz)))
        return fail(LRM_EINVAL, "null argument");
    if (nb == 0) return LRM_OK;
    LrmCompiledLeg host_legs[LRM_MAX_LEGS];
    for (size_t l = 0; l < nlegs; l++) lrm_compile_leg(legs[l], quat_or_default(quat), 0, &host_legs[l]);
    LrmCompiledLeg* dev_legs = nullptr;
    hipEvent_t* slot_done = nullptr;
    int rc = leg_slot(&dev_legs, &slot_done);
    if (rc != LRM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(dev_legs, host_legs, sizeof(LrmCompiledLeg) * nlegs, hipMemcpyHostToDevice,
                           (hipStream_t)stream), "hipMemcpyAsync legs");
    // pageable-source async copies are staged by the runtime before returning, so host_legs may die
    bool fast = g_mode != LRM_MODE_STRICT;
    for (size_t l = 0; l < nlegs; l++) fast = fast && host_legs[l].fast_ok;
    float* boxes = nullptr;
    if (nt >= 4096) { // below that the whole cloud is a handful of tiles: nothing to skip
        rc = tile_boxes(nt, &boxes);
        if (rc != LRM_OK) return rc;
    }
    HIP_TRY(lrm_launch_reach_any(bx, by, bz, nb, tx, ty, tz, nt, dev_legs, (int)nlegs, boxes, boxes_ready && boxes, body_active,
                                 out_leg_body, all_legs_out, fast, (hipStream_t)stream), "reach_any launch");
    HIP_TRY(hipEventRecord(*slot_done, (hipStream_t)stream), "hipEventRecord leg slot"); // the slot is free again after this launch
    return LRM_OK;
}
} // namespace

namespace {
int any_in_shape_impl(int shape, const float* cx, const float* cy, const float* cz, size_t nc, const float* tx,
                      const float* ty, const float* tz, size_t nt, float radius, float plus_z, float minus_z,
                      bool boxes_ready, uint8_t* out, void* stream) {
    if (!out || (nc && (!cx || !cy || !cz)) || (nt && (!tx || !ty || !tz))) return fail(LRM_EINVAL, "null argument");
    if (nc == 0) return LRM_OK;
    float* boxes = nullptr;
    if (nt >= 4096) {
        const int rc = tile_boxes(nt, &boxes);
        if (rc != LRM_OK) return rc;
    }
    HIP_TRY(lrm_launch_any_in_shape(shape, cx, cy, cz, nc, tx, ty, tz, nt, radius, plus_z, minus_z, boxes, boxes_ready && boxes, out,
                                    (hipStream_t)stream), shape ? "in_cylinder launch" : "in_sphere launch");
    return LRM_OK;
}
} // namespace
int lrm_any_in_sphere_dev(const float* cx, const float* cy, const float* cz, size_t nc, const float* tx,
                          const float* ty, const float* tz, size_t nt, float radius, uint8_t* out, void* stream) {
    return any_in_shape_impl(0, cx, cy, cz, nc, tx, ty, tz, nt, radius, 0.f, 0.f, false, out, stream);
}
int lrm_any_in_cylinder_dev(const float* cx, const float* cy, const float* cz, size_t nc, const float* tx,
                            const float* ty, const float* tz, size_t nt, float radius, float plus_z, float minus_z,
                            uint8_t* out, void* stream) {
    return any_in_shape_impl(1, cx, cy, cz, nc, tx, ty, tz, nt, radius, plus_z, minus_z, false, out, stream);
}

// Morton order of a host cloud (see morton_order above): order_out[k] = index of the k-th point.
int lrm_morton_order(const float* xyz_aos, size_t n, uint64_t* order_out) {
    if (n && (!xyz_aos || !order_out)) return fail(LRM_EINVAL, "null argument");
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    const std::vector<size_t> o = morton_order(xyz_aos, idx);
    for (size_t i = 0; i < n; i++) order_out[i] = (uint64_t)o[i];
    return LRM_OK;
}

// ---- diagnostics: lrm_exact_math.h on arrays (host build / device build) -------------------
int lrm_dbg_exact_math_host(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs) {
    if (n && (!a || !b || !at2 || !sn || !cs)) return fail(LRM_EINVAL, "null argument");
    for (size_t i = 0; i < n; i++) {
        at2[i] = lrm_atan2f(a[i], b[i]);
        lrm_sincosf(a[i], &sn[i], &cs[i]);
    }
    return LRM_OK;
}
int lrm_dbg_exact_math_dev(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs,
                           void* stream) {
    if (n && (!a || !b || !at2 || !sn || !cs)) return fail(LRM_EINVAL, "null argument");
    if (n == 0) return LRM_OK;
    HIP_TRY(lrm_launch_exact_math(a, b, n, at2, sn, cs, (hipStream_t)stream), "exact_math launch");
    return LRM_OK;
}

int lrm_dbg_sqrt_check_dev(uint64_t* mismatches_out, uint32_t* first_bad_out) {
    if (!mismatches_out || !first_bad_out) return fail(LRM_EINVAL, "null argument");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, 2 * sizeof(unsigned long long)), "hipMalloc");
    const unsigned long long init[2] = {0ull, ~0ull};
    hipError_t e = hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = lrm_launch_sqrt_check(d, nullptr);
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(e, "sqrt check");
    *mismatches_out = out[0];
    *first_bad_out = out[0] ? (uint32_t)(out[1] - 1) : 0u;
    return LRM_OK;
}

int lrm_dbg_pair_sphere(const LrmLegDimensions* leg, const float* quat, float* out4) {
    if (!leg || !out4) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 0, &L);
    for (int k = 0; k < 3; k++) out4[k] = L.pair_center[k];
    out4[3] = L.pair_r2;
    return LRM_OK;
}

int lrm_dbg_fused_reach_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
                             uint8_t* mask_out, uint8_t* doubt_out) {
    if (!leg || (n && (!xyz || !mask_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    if (!L.fast_ok) return fail(LRM_EINVAL, "leg not eligible for the filtered evaluation");
    const LrmDistTables T{&L.lists[0][0], &L.dist_tab[0][0], &L.corner_tab[0]};
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        uint32_t unc = 0;
        LrmDistByproduct by;
        lrm_dist_global_fast(L, T, p, unc, &by);
        bool doubt;
        mask_out[i] = lrm_reach_from_dist(L, by, doubt);
        doubt_out[i] = doubt;
    }
    return LRM_OK;
}

// The filtered evaluation (lrm_point_fast.h) on the host, WITHOUT the strict fallback, with its
// `uncertain` flags: tests check that every point not flagged equals the strict result and
// count how many are flagged.  Outputs may be NULL.
int lrm_dbg_fast_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
                      uint8_t* mask_out, uint8_t* mask_unc_out, float* dxyz_out, uint8_t* valid_out,
                      uint8_t* dist_unc_out) {
    if (!leg || (n && !xyz)) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    if (!L.fast_ok) return fail(LRM_EINVAL, "leg not eligible for the filtered evaluation");
    for (size_t i = 0; i < n; i++) {
        const LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        if (mask_out) {
            uint32_t unc = 0;
            mask_out[i] = lrm_reach_global_fast(L, &L.lean[0][0], p, unc);
            if (mask_unc_out) mask_unc_out[i] = (uint8_t)unc;
        }
        if (dxyz_out) {
            uint32_t unc = 0;
            LrmVec3 d = p;
            const bool v = lrm_dist_global_fast(L, LrmDistTables{&L.lists[0][0], &L.dist_tab[0][0], &L.corner_tab[0]}, d, unc);
            dxyz_out[3 * i] = d.x;
            dxyz_out[3 * i + 1] = d.y;
            dxyz_out[3 * i + 2] = d.z;
            if (valid_out) valid_out[i] = v;
            if (dist_unc_out) dist_unc_out[i] = (uint8_t)unc;
        }
    }
    return LRM_OK;
}

// The contract-tolerance evaluation on the host, WITHOUT the re-evaluation of its doubtful points.
int lrm_dbg_tol_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                     float* dxyz_out, uint32_t* doubt_out) {
    if (!leg || (n && (!xyz || !mask_out || !dxyz_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the tolerance mode");
    const LrmTolTables T{&TL.circ[0][0], &TL.feat[0]};
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        uint32_t doubt = 0;
        mask_out[i] = lrm_dist_tol(TL, T, p, doubt);
        dxyz_out[3 * i] = p.x;
        dxyz_out[3 * i + 1] = p.y;
        dxyz_out[3 * i + 2] = p.z;
        doubt_out[i] = doubt;
    }
    return LRM_OK;
}
// as lrm_dbg_tol_host with the plane table with deferred decisions (lrm_toltab.cpp) in place of the full plane evaluation;
// doubt bit 0x100 = a cell without an answer.  stats_out[4] (optional): rows, validity rows, refined cells, table bytes
int lrm_dbg_toltab_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                        float* dxyz_out, uint32_t* doubt_out, uint32_t* stats_out) {
    if (!leg || (n && (!xyz || !mask_out || !dxyz_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the tolerance mode");
    std::vector<uint8_t> tab;
    if (!lrm_build_tol_tab(TL, &tab)) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab.data());
    const LrmTolTabView G = lrm_toltab_view(tab.data(), hd->rows, hd->vrows, lrm_toltab_bound_inner(tab.data()), TL.r_outer);
    if (stats_out) {
        stats_out[0] = hd->n_rows;
        stats_out[1] = hd->n_vrows;
        stats_out[2] = hd->n_fine[0] + hd->n_fine[1];
        stats_out[3] = (uint32_t)tab.size();
    }
    lrm_tab_host_seconds = 0;
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        uint32_t doubt = 0;
        mask_out[i] = lrm_tab_point(TL, G, p, doubt);
        dxyz_out[3 * i] = p.x;
        dxyz_out[3 * i + 1] = p.y;
        dxyz_out[3 * i + 2] = p.z;
        doubt_out[i] = doubt;
    }
    if (stats_out) stats_out[4] = (uint32_t)std::min<unsigned long long>(lrm_tab_host_seconds, 0xffffffffull);
    return LRM_OK;
}
// The bit-exact table-guided evaluation (lrm_point_xtab.h) on the host, WITHOUT the re-evaluation of its doubtful points:
// every point with doubt 0 must equal lrm_dist_cpu / lrm_reach_cpu bit for bit.  stats_out[2] (optional): points whose second
// chain (twin or second candidate) ran, table bytes
int lrm_dbg_xtab_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                      float* dxyz_out, uint32_t* doubt_out, uint32_t* stats_out) {
    if (!leg || (n && (!xyz || !mask_out || !dxyz_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the table-guided modes");
    std::vector<uint8_t> tab;
    if (!lrm_build_tol_tab(TL, &tab)) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab.data());
    const LrmTolTabView G = lrm_toltab_view(tab.data(), hd->rows, hd->vrows, lrm_toltab_bound_inner(tab.data()), TL.r_outer);
    LrmXtabLeg X;
    lrm_make_xtab_leg(L, TL, &X);
    lrm_xtab_host_seconds = 0;
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        uint32_t doubt = 0;
        mask_out[i] = lrm_xtab_point(X, G, p, doubt);
        dxyz_out[3 * i] = p.x;
        dxyz_out[3 * i + 1] = p.y;
        dxyz_out[3 * i + 2] = p.z;
        doubt_out[i] = doubt;
    }
    if (stats_out) {
        stats_out[0] = (uint32_t)std::min<unsigned long long>(lrm_xtab_host_seconds, 0xffffffffull);
        stats_out[1] = (uint32_t)tab.size();
    }
    return LRM_OK;
}
// LRM_MODE_TOL_REL's two steps on the host: the tolerance evaluation with the plane table (lrm_tab_point<true>: mask, doubt bits and
// the DECISIONS it took), then the strict replay of the winner's value chain from those decisions (lrm_xtab_replay) for every point
// without doubt: dxyz_out must then equal lrm_dist_cpu bit for bit (tests/test_xtab_cpu.py).
int lrm_dbg_replay_host(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, uint8_t* mask_out,
                        float* dxyz_out, uint32_t* doubt_out) {
    if (!leg || (n && (!xyz || !mask_out || !dxyz_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the table-guided modes");
    std::vector<uint8_t> tab;
    if (!lrm_build_tol_tab(TL, &tab)) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab.data());
    const LrmTolTabView G = lrm_toltab_view(tab.data(), hd->rows, hd->vrows, lrm_toltab_bound_inner(tab.data()), TL.r_outer);
    LrmXtabLeg X;
    lrm_make_xtab_leg(L, TL, &X);
    for (size_t i = 0; i < n; i++) {
        const LrmVec3 p_in{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        LrmVec3 p = p_in;
        uint32_t doubt = 0, info = 0;
        mask_out[i] = lrm_tab_point<true>(TL, G, p, doubt, &info);
        if ((doubt & 0xffffu) == 0u) {
            p = p_in;
            lrm_xtab_replay(X, hd->rows, p, info);
        }
        dxyz_out[3 * i] = p.x;
        dxyz_out[3 * i + 1] = p.y;
        dxyz_out[3 * i + 2] = p.z;
        doubt_out[i] = doubt;
    }
    return LRM_OK;
}
// The plane table of (leg, quat) as the host builder (device = 0) or the device builder (device = 1, on the current device) makes
// it: the table's bytes into out[cap] (when it fits), its size, the build's milliseconds.  LRM_EINVAL when the leg has no table.
int lrm_dbg_toltab_build(const LrmLegDimensions* leg, const float* quat, int device, uint8_t* out, size_t cap, size_t* size_out, float* ms_out) {
    if (!leg || !size_out) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the table-guided modes");
    if (!device) {
        std::vector<uint8_t> tab;
        const auto t0 = std::chrono::steady_clock::now();
        if (!lrm_build_tol_tab(TL, &tab)) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
        if (ms_out) *ms_out = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *size_out = tab.size();
        if (out && cap >= tab.size()) std::memcpy(out, tab.data(), tab.size());
        return LRM_OK;
    }
    uint8_t* t = nullptr;
    size_t bytes = 0;
    float ms = 0.f;
    const int rc = lrm_build_tol_tab_dev(TL, nullptr, &t, &bytes, &ms);
    if (rc < 0) return hip_fail((hipError_t)(-rc), "plane table (device build)");
    if (rc == 1) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
    if (rc == 2) return fail(LRM_EINVAL, "the device builder declines this leg (too many unanswered cells)");
    *size_out = bytes;
    if (ms_out) *ms_out = ms;
    hipError_t e = hipSuccess;
    if (out && cap >= bytes) e = hipMemcpy(out, t, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(t);
    HIP_TRY(e, "hipMemcpy plane table");
    return LRM_OK;
}
// The table's lower bound of the in-plane distance at plane points (x = abscissa - coxa_length, z) of the INNER grid, next
// to what the full plane evaluation (lrm_tol_plane) finds there: tests/test_tol_cpu.py checks bound <= distance.
int lrm_dbg_toltab_bounds(const float* xz, size_t n, const LrmLegDimensions* leg, const float* quat, float* lb_out,
                          float* dist_out, uint8_t* valid_out, uint32_t* doubt_out) {
    if (!leg || (n && (!xz || !lb_out || !dist_out || !valid_out || !doubt_out))) return fail(LRM_EINVAL, "null argument");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    if (!TL.tol_ok) return fail(LRM_EINVAL, "leg not eligible for the tolerance mode");
    std::vector<uint8_t> tab;
    if (!lrm_build_tol_tab(TL, &tab)) return fail(LRM_EINVAL, "leg needs more table rows than a cell code can name");
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab.data());
    const LrmTolTabView G = lrm_toltab_view(tab.data(), hd->rows, hd->vrows, lrm_toltab_bound_inner(tab.data()), TL.r_outer);
    const LrmTolTables T{&TL.circ[0][0], &TL.feat[0]};
    for (size_t i = 0; i < n; i++) {
        const float x = xz[2 * i], z = xz[2 * i + 1];
        const int g = (std::fabs(x) < G.far_limit && std::fabs(z) < G.far_limit) ? 0 : 1;
        uint32_t c0, c1;
        float l0, l1;
        { bool bd; uint32_t s0, s1, fb; if (g) lrm_toltab_lookup2<true>(G, true, 0.f, x, x, z, c0, c1, s0, s1, fb, l0, l1, bd); else lrm_toltab_lookup2<false>(G, false, 0.f, x, x, z, c0, c1, s0, s1, fb, l0, l1, bd); }
        lb_out[i] = l0;
        const float band = TL.band_base + TL.band_slope * (std::fabs(x) + std::fabs(z)), tau = band * LRM_TOL_TIE;
        float du, dz;
        bool valid;
        uint32_t lu = 0;
        lrm_tol_plane(TL, T, x + TL.coxa_length, z, band, tau, du, dz, valid, lu);
        dist_out[i] = std::sqrt(du * du + dz * dz);
        valid_out[i] = valid;
        doubt_out[i] = lu;
    }
    return LRM_OK;
}
// Counting build only (-DLRM_PAIR_COUNT, tools/c3_evidence.py): what reach_any_wave_kernel evaluated since the last call:
// out[0] full (leg, target) evaluations, [1] leg bounding-sphere tests, [2] footholds inside a body's reach sphere, [3] footholds loaded
int lrm_dbg_pair_counts(uint64_t out[4]) {
    if (!out) return fail(LRM_EINVAL, "null argument");
    unsigned long long c[4] = {0, 0, 0, 0};
    const hipError_t e = lrm_pair_counts(c);
    if (e == hipErrorNotSupported) return fail(LRM_EINVAL, "this library was built without -DLRM_PAIR_COUNT");
    HIP_TRY(e, "pair counters");
    for (int i = 0; i < 4; i++) out[i] = c[i];
    return LRM_OK;
}
// After a distance / fused call on device buffers in LRM_MODE_TOL (synchronises that device): how many of its points the
// main kernel queued for the bit-exact fix-up, and how many workgroups overflowed their queue segment (their points are
// all re-evaluated).  Valid until the next tolerance-mode call on a larger cloud regrows the workspace.
int lrm_dbg_tol_queue_counts(uint64_t* n_points, uint64_t* n_queued, uint64_t* n_overflowed) {
    if (!n_points || !n_queued || !n_overflowed) return fail(LRM_EINVAL, "null argument");
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    if (!g_tol_last.counts) return fail(LRM_EINVAL, "no tolerance-mode call on device buffers yet");
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur), "hipGetDevice");
    HIP_TRY(hipSetDevice(g_tol_last.dev), "hipSetDevice");
    std::vector<uint32_t> c(g_tol_last.blocks);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(c.data(), g_tol_last.counts, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipSetDevice(cur);
    HIP_TRY(e, "hipMemcpy counts");
    uint64_t q = 0, o = 0;
    for (uint32_t v : c) {
        q += v;
        o += v > g_tol_last.cap;
    }
    *n_points = g_tol_last.n;
    *n_queued = q;
    *n_overflowed = o;
    return LRM_OK;
}
int lrm_dbg_tol_ok(const LrmLegDimensions* leg, const float* quat) {
    if (!leg) return 0;
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    LrmTolLeg TL;
    lrm_compile_tol(L, &TL);
    return TL.tol_ok;
}

namespace {
// The orientation loop of robot_full_struct (several_leg.cu:831-857 -> runPipeline :762-787) on device-resident clouds:
// per orientation rotate both clouds, (optionally) the cylinder culls of eliminateFarAndColliding (:504-525), every
// leg's any-target reachability, and the monotone acceptance (an `active` byte per body).  Everything on the null
// stream; only launches (the per-orientation leg data is a few hundred bytes of kernel arguments).
// b*/t*: the unrotated clouds (SoA), rb*/rt*: scratch for the rotated ones, rot_dev: nquat LrmCompiledLeg slots,
// m1/m2: max(nb, nt) bytes, leg_body: nlegs * nb, all: nb, active (in/out) and accepted (in/out): nb bytes.
int sweep_orientations_dev(const float* bx, const float* by, const float* bz, size_t nb, const float* tx, const float* ty, const float* tz, size_t mt,
                           const LrmLegDimensions* legs, size_t nlegs, const float* quats, size_t nquat, int culls,
                           float* rbx, float* rby, float* rbz, float* rtx, float* rty, float* rtz, LrmCompiledLeg* rot_dev,
                           uint8_t* m1, uint8_t* m2, uint8_t* leg_body, uint8_t* all, uint8_t* active, uint8_t* accepted) {
    {   // every orientation's qtRotate coefficients in one upload: the loop below only launches
        std::vector<LrmCompiledLeg> rots(nquat);
        LrmLegDimensions dummy{};
        for (size_t qi = 0; qi < nquat; qi++) lrm_compile_leg(dummy, quats + 4 * qi, 0, &rots[qi]); // only fwd_rot is used
        if (nquat) HIP_TRY(hipMemcpy(rot_dev, rots.data(), sizeof(LrmCompiledLeg) * nquat, hipMemcpyHostToDevice), "hipMemcpy rotations");
    }
    for (size_t qi = 0; qi < nquat; qi++) {
        const float* q = quats + 4 * qi;
        LrmLegDimensions rl[LRM_MAX_LEGS];
        for (size_t l = 0; l < nlegs; l++) lrm_host_rotate_leg_data(q, legs[l], &rl[l]);
        HIP_TRY(lrm_launch_rotate_soa(bx, by, bz, nb, rot_dev + qi, rbx, rby, rbz, nullptr), "rotate bodies");
        if (mt) HIP_TRY(lrm_launch_rotate_soa(tx, ty, tz, mt, rot_dev + qi, rtx, rty, rtz, nullptr), "rotate targets");
        bool boxes_ready = false; // the kernels of one orientation share the rotated cloud's boxes: only the first builds them
        int rc = LRM_OK;
        if (culls) {
            // eliminateFarAndColliding, several_leg.cu:504-525, with the rotated leg 0
            const LrmLegDimensions& d = rl[0];
            const float s_pitch = sinf(d.coxa_pitch), c_pitch = cosf(d.coxa_pitch);
            const float radius_in = d.body + c_pitch * d.coxa_length + d.femur_length + d.tibia_length;
            const float half_pi = 3.14159265358979323846264338327950288419716939937510582097f / 2;
            const float plus_abs = d.tibia_length * sinf(d.tibia_absolute_pos) +
                                   d.femur_length * sinf(half_pi < d.max_angle_femur ? half_pi : d.max_angle_femur);
            const float plus_z_in = s_pitch * d.coxa_length + plus_abs;
            const float minus_z_in = s_pitch * d.coxa_length - d.femur_length - d.tibia_length;
            rc = any_in_shape_impl(1, rbx, rby, rbz, nb, rtx, rty, rtz, mt, radius_in, plus_z_in, minus_z_in, boxes_ready, m1, nullptr);
            boxes_ready = mt >= 4096;
            if (rc == LRM_OK) rc = any_in_shape_impl(1, rbx, rby, rbz, nb, rtx, rty, rtz, mt, d.body, 250.f, -110.f, boxes_ready, m2, nullptr);
            if (rc != LRM_OK) return rc;
        }
        rc = reach_any_impl(rbx, rby, rbz, nb, rtx, rty, rtz, mt, rl, nlegs, q, active, boxes_ready, leg_body, all, nullptr);
        if (rc != LRM_OK) return rc;
        HIP_TRY(lrm_launch_sweep_update(all, m1, m2, culls ? 1 : 0, nb, active, accepted, nullptr), "sweep update");
    }
    return LRM_OK;
}
} // namespace

// robot_full_struct's pipeline (several_leg.cu:326-877) with masks instead of thrust stream
// compaction, resident on the device from the first upload to the final mask.
// reference_culls != 0 adds the estimator's culls:
//   once      eliminateAlwaysColliding (sphere r = 60, :413-440), eliminateFarBody (r = 400, :442-474),
//             eliminateFarTarget (r = 400 around the surviving bodies, :476-502)
//   per quat  eliminateFarAndColliding (:504-559): keep a body iff some target lies in the big
//             cylinder of (rotated) leg 0 and none in the body cylinder (r = body, z in (-110, 250))
// For each orientation bodies and targets are rotated by the quaternion (rotateData :401-411: a
// kernel with the strict qtRotate arithmetic), the legs' limits are rotated (rotateLegsLimits
// :743-760, host), every leg must find a reachable target (eliminateUnreachable :707-741,
// generalised from 4 to nlegs legs), and a body accepted by one orientation is not tested again
// (flipWorkingSide :396-399: an `active` byte per body instead of partitioning).
int lrm_positionability(const float* bodies, size_t nb, const float* targets, size_t nt,
                        const LrmLegDimensions* legs, size_t nlegs, const float* quats, size_t nquat,
                        int reference_culls, uint8_t* body_mask_out, float* ms) {
    if (!legs || nlegs == 0 || nlegs > LRM_MAX_LEGS) return fail(LRM_EINVAL, "nlegs must be 1..LRM_MAX_LEGS");
    if ((nb && (!bodies || !body_mask_out)) || (nt && !targets) || (nquat && !quats))
        return fail(LRM_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
    if (nb == 0) return LRM_OK;
    Events ev;
    HIP_TRY(hipEventCreate(&ev.a), "hipEventCreate");
    HIP_TRY(hipEventCreate(&ev.b), "hipEventCreate");
    const size_t pb = (nb + 3) & ~(size_t)3, pt = ((nt ? nt : 1) + 3) & ~(size_t)3; // keep component arrays aligned
    DevBuf d_b0, d_t0, d_b, d_t, d_m1, d_m2, d_leg_body, d_all, d_active, d_accepted, d_rot;
    HIP_TRY(d_b0.alloc(3 * pb * sizeof(float)), "hipMalloc bodies");
    HIP_TRY(d_b.alloc(3 * pb * sizeof(float)), "hipMalloc bodies");
    HIP_TRY(d_t0.alloc(3 * pt * sizeof(float)), "hipMalloc targets");
    HIP_TRY(d_t.alloc(3 * pt * sizeof(float)), "hipMalloc targets");
    HIP_TRY(d_m1.alloc(nb > nt ? nb : nt), "hipMalloc mask");
    HIP_TRY(d_m2.alloc(nb > nt ? nb : nt), "hipMalloc mask");
    HIP_TRY(d_leg_body.alloc(nlegs * nb), "hipMalloc leg results");
    HIP_TRY(d_all.alloc(nb), "hipMalloc body results");
    HIP_TRY(d_active.alloc(nb), "hipMalloc active");
    HIP_TRY(d_accepted.alloc(nb), "hipMalloc accepted");
    HIP_TRY(d_rot.alloc(sizeof(LrmCompiledLeg) * (nquat ? nquat : 1)), "hipMalloc rotations");

    // AoS host -> SoA device (component stride `pad`)
    std::vector<float> stage;
    auto upload = [&](const float* aos, const size_t* idx, size_t m, size_t pad, void* dst) -> hipError_t {
        stage.assign(3 * pad, 0.f);
        for (size_t k = 0; k < m; k++) {
            const size_t i = idx ? idx[k] : k;
            stage[k] = aos[3 * i];
            stage[pad + k] = aos[3 * i + 1];
            stage[2 * pad + k] = aos[3 * i + 2];
        }
        return hipMemcpy(dst, stage.data(), 3 * pad * sizeof(float), hipMemcpyHostToDevice);
    };
    // both clouds go to the device in Morton order (body_order maps device slot -> caller index)
    std::vector<size_t> ident_b(nb), ident_t(nt);
    for (size_t i = 0; i < nb; i++) ident_b[i] = i;
    for (size_t i = 0; i < nt; i++) ident_t[i] = i;
    const std::vector<size_t> body_order = morton_order(bodies, ident_b);
    const std::vector<size_t> target_order = morton_order(targets, ident_t);
    HIP_TRY(upload(bodies, body_order.data(), nb, pb, d_b0.p), "hipMemcpy bodies");
    size_t mt = nt; // targets kept after the one-time cull
    if (nt) HIP_TRY(upload(targets, target_order.data(), nt, pt, d_t0.p), "hipMemcpy targets");
    float* B0 = d_b0.as<float>();
    float* T0 = d_t0.as<float>();
    float* B = d_b.as<float>();
    float* T = d_t.as<float>();
    std::vector<uint8_t> active(nb, 1), h1(nb > nt ? nb : nt), h2(nb);
    float total_ms = 0.f;
    HIP_TRY(hipMemset(d_accepted.p, 0, nb), "hipMemset accepted");

    if (reference_culls == 1 && nt) { // 2: the caller has applied the one-time culls (sharded drivers)
        HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
        int rc = lrm_any_in_sphere_dev(B0, B0 + pb, B0 + 2 * pb, nb, T0, T0 + pt, T0 + 2 * pt, nt, 60.f, d_m1.as<uint8_t>(), nullptr);
        if (rc == LRM_OK) rc = lrm_any_in_sphere_dev(B0, B0 + pb, B0 + 2 * pb, nb, T0, T0 + pt, T0 + 2 * pt, nt, 400.f, d_m2.as<uint8_t>(), nullptr);
        if (rc != LRM_OK) return rc;
        HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
        HIP_TRY(hipMemcpy(h1.data(), d_m1.p, nb, hipMemcpyDeviceToHost), "hipMemcpy mask");
        HIP_TRY(hipMemcpy(h2.data(), d_m2.p, nb, hipMemcpyDeviceToHost), "hipMemcpy mask");
        float e = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, ev.a, ev.b), "hipEventElapsedTime");
        total_ms += e;
        std::vector<size_t> alive; // caller indices, in device (Morton) order
        for (size_t i = 0; i < nb; i++) {
            active[i] = (h1[i] == 0 && h2[i] != 0) ? 1 : 0;
            if (active[i]) alive.push_back(body_order[i]);
        }
        // eliminateFarTarget: keep the targets with a surviving body within 400 (compacted once, on the host)
        mt = 0;
        if (!alive.empty()) {
            const size_t na = alive.size(), pa = (na + 3) & ~(size_t)3;
            HIP_TRY(upload(bodies, alive.data(), na, pa, d_b.p), "hipMemcpy bodies");
            HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
            rc = lrm_any_in_sphere_dev(T0, T0 + pt, T0 + 2 * pt, nt, B, B + pa, B + 2 * pa, na, 400.f, d_m1.as<uint8_t>(), nullptr);
            if (rc != LRM_OK) return rc;
            HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
            HIP_TRY(hipMemcpy(h1.data(), d_m1.p, nt, hipMemcpyDeviceToHost), "hipMemcpy mask");
            HIP_TRY(hipEventElapsedTime(&e, ev.a, ev.b), "hipEventElapsedTime");
            total_ms += e;
            std::vector<size_t> kept; // caller indices, still in Morton order
            for (size_t i = 0; i < nt; i++)
                if (h1[i]) kept.push_back(target_order[i]);
            mt = kept.size();
            if (mt) HIP_TRY(upload(targets, kept.data(), mt, pt, d_t0.p), "hipMemcpy targets");
        }
    }
    HIP_TRY(hipMemcpy(d_active.p, active.data(), nb, hipMemcpyHostToDevice), "hipMemcpy active");

    HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
    {
        const int rc = sweep_orientations_dev(B0, B0 + pb, B0 + 2 * pb, nb, T0, T0 + pt, T0 + 2 * pt, mt, legs, nlegs, quats, nquat, reference_culls ? 1 : 0,
                                              B, B + pb, B + 2 * pb, T, T + pt, T + 2 * pt, d_rot.as<LrmCompiledLeg>(), d_m1.as<uint8_t>(), d_m2.as<uint8_t>(),
                                              d_leg_body.as<uint8_t>(), d_all.as<uint8_t>(), d_active.as<uint8_t>(), d_accepted.as<uint8_t>());
        if (rc != LRM_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
    HIP_TRY(hipMemcpy(h2.data(), d_accepted.p, nb, hipMemcpyDeviceToHost), "hipMemcpy result");
    for (size_t i = 0; i < nb; i++) body_mask_out[body_order[i]] = h2[i];
    float e = 0.f;
    HIP_TRY(hipEventElapsedTime(&e, ev.a, ev.b), "hipEventElapsedTime");
    total_ms += e;
    if (ms) *ms = total_ms;
    return LRM_OK;
}

// lrm_positionability on clouds that already live on the device (SoA float32) with device-resident masks in and out: the
// orientation sweep alone (reference_culls 0: none; 2: the per-orientation cylinder culls -- the caller has applied the
// one-time culls of multi_rot_estimator, several_leg.cu:413-502, as the sharded drivers do).  active_in: NULL = every
// body.  No copies of the clouds, no reordering (feed Morton-ordered clouds: lrm_morton_order); returns after the device
// has finished (null stream).
int lrm_positionability_dev(const float* bx, const float* by, const float* bz, size_t nb, const float* tx, const float* ty,
                            const float* tz, size_t nt, const LrmLegDimensions* legs, size_t nlegs, const float* quats,
                            size_t nquat, int reference_culls, const uint8_t* active_in, uint8_t* accepted_out, float* ms) {
    if (!legs || nlegs == 0 || nlegs > LRM_MAX_LEGS) return fail(LRM_EINVAL, "nlegs must be 1..LRM_MAX_LEGS");
    if ((nb && (!bx || !by || !bz || !accepted_out)) || (nt && (!tx || !ty || !tz)) || (nquat && !quats)) return fail(LRM_EINVAL, "null argument");
    if (reference_culls != 0 && reference_culls != 2) return fail(LRM_EINVAL, "reference_culls must be 0 or 2 here (the one-time culls are the caller's)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
    if (ms) *ms = 0.f;
    if (nb == 0) return LRM_OK;
    Events ev;
    HIP_TRY(hipEventCreate(&ev.a), "hipEventCreate");
    HIP_TRY(hipEventCreate(&ev.b), "hipEventCreate");
    const size_t pb = (nb + 3) & ~(size_t)3, pt = ((nt ? nt : 1) + 3) & ~(size_t)3;
    DevBuf d_b, d_t, d_m1, d_m2, d_leg_body, d_all, d_active, d_rot;
    HIP_TRY(d_b.alloc(3 * pb * sizeof(float)), "hipMalloc bodies");
    HIP_TRY(d_t.alloc(3 * pt * sizeof(float)), "hipMalloc targets");
    HIP_TRY(d_m1.alloc(nb > nt ? nb : nt), "hipMalloc mask");
    HIP_TRY(d_m2.alloc(nb > nt ? nb : nt), "hipMalloc mask");
    HIP_TRY(d_leg_body.alloc(nlegs * nb), "hipMalloc leg results");
    HIP_TRY(d_all.alloc(nb), "hipMalloc body results");
    HIP_TRY(d_active.alloc(nb), "hipMalloc active");
    HIP_TRY(d_rot.alloc(sizeof(LrmCompiledLeg) * (nquat ? nquat : 1)), "hipMalloc rotations");
    if (active_in) HIP_TRY(hipMemcpy(d_active.p, active_in, nb, hipMemcpyDeviceToDevice), "hipMemcpy active");
    else HIP_TRY(hipMemset(d_active.p, 1, nb), "hipMemset active");
    HIP_TRY(hipMemset(accepted_out, 0, nb), "hipMemset accepted");
    float* B = d_b.as<float>();
    float* T = d_t.as<float>();
    HIP_TRY(hipEventRecord(ev.a, nullptr), "hipEventRecord");
    const int rc = sweep_orientations_dev(bx, by, bz, nb, tx, ty, tz, nt, legs, nlegs, quats, nquat, reference_culls ? 1 : 0, B, B + pb, B + 2 * pb,
                                          T, T + pt, T + 2 * pt, d_rot.as<LrmCompiledLeg>(), d_m1.as<uint8_t>(), d_m2.as<uint8_t>(),
                                          d_leg_body.as<uint8_t>(), d_all.as<uint8_t>(), d_active.as<uint8_t>(), accepted_out);
    if (rc != LRM_OK) return rc;
    HIP_TRY(hipEventRecord(ev.b, nullptr), "hipEventRecord");
    HIP_TRY(hipEventSynchronize(ev.b), "hipEventSynchronize");
    float e = 0.f;
    HIP_TRY(hipEventElapsedTime(&e, ev.a, ev.b), "hipEventElapsedTime");
    if (ms) *ms = e;
    return LRM_OK;
}

} // extern "C"

// =====================================================================================================================
// Several GPUs behind the apply_kernel boundary (SURVEY.md section 8(b) "lrm_*_multi", 8(e)): ONE process, ONE host
// thread, ndev devices.  The reference has no such entry (several_leg.cu:800 uses device 0).  The cloud is cut into the
// 64-point-aligned contiguous shards of lrm_shard_bounds -- the same arithmetic as lrm_amd/shard.py, so that a
// torch.distributed run and this call own identical slices -- each device gets one stream, its slice of the input,
// the fused kernels of the current mode, and packs its reach bytes into ballot words; the words are then all-gathered
// over xGMI with RCCL's C API (ncclCommInitAll once per device set, one grouped ncclAllGather per call), so that every
// device ends up with the bit-packed reach mask of the WHOLE cloud (the exchange step of BASELINE config 4), and the
// per-shard bytes / vectors go back to the caller's arrays.  librccl.so is opened on first use with more than one
// device (or when LRM_MULTI_FORCE_RCCL=1 asks for it with one): liblrm.so itself keeps depending on HIP alone.
// =====================================================================================================================
#include <dlfcn.h>

namespace {

__global__ void pack_mask_bits_kernel(const uint8_t* __restrict__ mask, size_t n, uint64_t* __restrict__ words, size_t nwords) {
    // one wave per word: lane i reads byte 64 w + i; words beyond ceil(n / 64) (the shard's padding) become 0
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= nwords) return;
    const size_t i = w * 64 + (threadIdx.x & 63);
    const uint64_t b = __ballot(i < n && mask[i] != 0);
    if ((threadIdx.x & 63) == 0) words[w] = b;
}

struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string* why) {
        if (handle) return true;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names)
            if ((handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!handle) { *why = std::string("dlopen librccl.so: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
            *why = "librccl.so lacks an expected symbol";
            dlclose(handle);
            handle = nullptr;
            return false;
        }
        return true;
    }
} g_rccl;
constexpr int kNcclUint64 = 5; // ncclUint64, rccl.h

// communicators of the last device set (ncclCommInitAll costs ~0.1-1 s: kept across calls)
struct MultiComms {
    std::vector<int> devs;
    std::vector<void*> comms;
} g_multi;
void multi_release() {
    if (g_rccl.handle)
        for (void* c : g_multi.comms)
            if (c) (void)g_rccl.CommDestroy(c);
    g_multi.comms.clear();
    g_multi.devs.clear();
}

struct MultiDev {
    int dev = 0;
    size_t lo = 0, hi = 0;
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    void *in = nullptr, *mask = nullptr, *out = nullptr, *words = nullptr, *gathered = nullptr;
};
struct MultiState { // releases everything on every exit path
    std::vector<MultiDev> d;
    int home = 0;
    ~MultiState() {
        for (auto& m : d) {
            if (hipSetDevice(m.dev) != hipSuccess) continue;
            if (m.st) (void)hipStreamSynchronize(m.st);
            if (m.st) tol_workspace_drop(m.dev, (void*)m.st); // the queue workspace keyed by this call's stream goes with the stream
            for (void* p : {m.in, m.mask, m.out, m.words, m.gathered})
                if (p) (void)hipFree(p);
            if (m.a) (void)hipEventDestroy(m.a);
            if (m.b) (void)hipEventDestroy(m.b);
            if (m.st) (void)hipStreamDestroy(m.st);
        }
        (void)hipSetDevice(home);
    }
};

} // namespace

extern "C" {

// LRM_MODE_TOL ahead of time: tables compiled and uploaded, queues of (current device, stream) sized for n_max points
int lrm_tol_prepare(const LrmLegDimensions* leg, const float* quat, size_t n_max, void* stream) {
    if (!leg) return fail(LRM_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    if (!L.fast_ok) return LRM_OK; // this leg runs the strict kernels in every mode: nothing to prepare
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    TolEntry& E = tol_entry(*leg, quat_or_default(quat), L);
    if (!E.tl.tol_ok) return LRM_OK; // falls back to LRM_MODE_FAST
    const uint8_t* tab = nullptr;
    if (tol_tab_wanted(n_max)) {
        const int rc = tol_tab_device(E, stream, &tab);
        if (rc != LRM_OK) return rc;
    }
    uint32_t* w = nullptr;
    return tol_workspace(std::max(lrm_tol_tab_queue_words(n_max), lrm_tol_queue_words(n_max)), stream, &w);
}

// milliseconds the most recent plane-table build of this process took (host wall clock around the builder; -1: none yet)
int lrm_tol_table_build_ms(float* ms_out) {
    if (!ms_out) return fail(LRM_EINVAL, "null argument");
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    *ms_out = g_last_tab_build_ms;
    return LRM_OK;
}

void lrm_release_workspaces(void) {
    std::lock_guard<std::recursive_mutex> g(g_cache_mu);
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto& w : g_tol_ws) {
        if (!w.second.p) continue;
        (void)hipSetDevice(w.first.first);
        (void)hipFree(w.second.p);
    }
    g_tol_ws.clear();
    g_tol_last = TolLast{};
    for (auto& e : g_tol_cache)
        for (auto& d : e.second.tab_dev) {
            (void)hipSetDevice(d.first);
            (void)hipFree(d.second);
        }
    g_tol_cache.clear();
    for (auto& hp : g_host_pipe) {
        (void)hipSetDevice(hp.first);
        host_pipe_free(hp.second);
    }
    g_host_pipe.clear();
    lrm_toltab_dev_release();
    multi_release();
    if (have) (void)hipSetDevice(cur);
}

// shard r of n items over `world` owners: [lo, hi), boundaries multiples of `align` (64: no ballot word straddles two
// owners); ceil(n / world) rounded up to `align` per owner, the last ones may be short or empty.  = shard.shard_bounds
int lrm_shard_bounds(size_t n, int world, int rank, size_t align, size_t* lo_out, size_t* hi_out) {
    if (world < 1 || rank < 0 || rank >= world || align == 0 || !lo_out || !hi_out) return fail(LRM_EINVAL, "bad shard arguments");
    size_t per = (n + (size_t)world - 1) / (size_t)world;
    per = (per + align - 1) / align * align;
    const size_t lo = std::min(n, (size_t)rank * per);
    *lo_out = lo;
    *hi_out = std::min(n, lo + per);
    return LRM_OK;
}

void lrm_multi_release(void) { multi_release(); }

// apply_kernel (cross_compiled.cu:33-79) for the fused kernel over `ndev` devices.  devices: ordinals, or NULL = 0 .. ndev-1.
// mask_out[n], dxyz_out[3n]: as lrm_reach_dist.  bits_out: NULL, or ceil(n / 64) words = the gathered bit-packed mask as
// device devices[0] holds it after the exchange.  ms_per_dev: NULL or [ndev], each device's kernel milliseconds (HIP events
// on its stream around its launches; 0 for an empty shard).
int lrm_reach_dist_multi(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat, int ndev, const int* devices,
                         uint8_t* mask_out, float* dxyz_out, uint64_t* bits_out, float* ms_per_dev) {
    if (!leg || (n && (!xyz || !mask_out || !dxyz_out))) return fail(LRM_EINVAL, "null argument");
    if (ndev < 1 || ndev > 64) return fail(LRM_EINVAL, "ndev must be 1 .. 64");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have == 0) return fail(LRM_ENODEV, "no HIP device");
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; d++) {
        devs[d] = devices ? devices[d] : d;
        if (devs[d] < 0 || devs[d] >= have) return fail(LRM_EINVAL, "device ordinal out of range");
        for (int e = 0; e < d; e++)
            if (devs[e] == devs[d]) return fail(LRM_EINVAL, "a device is listed twice");
    }
    const char* force = std::getenv("LRM_MULTI_FORCE_RCCL");
    const bool use_rccl = ndev > 1 || (force && force[0] == '1');
    if (use_rccl) {
        std::string why;
        if (!g_rccl.load(&why)) return fail(LRM_ENODEV, why.c_str());
        if (g_multi.devs != devs) {
            multi_release();
            g_multi.comms.assign(ndev, nullptr);
            const int rc = g_rccl.CommInitAll(g_multi.comms.data(), ndev, devs.data());
            if (rc != 0) {
                g_multi.comms.clear();
                return fail(LRM_ENODEV, (std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc)).c_str());
            }
            g_multi.devs = devs;
        }
    }
    LrmCompiledLeg L;
    lrm_compile_leg(*leg, quat_or_default(quat), 1, &L);
    MultiState S;
    (void)hipGetDevice(&S.home);
    S.d.resize(ndev);
    size_t lo0 = 0, per = 0;
    lrm_shard_bounds(n, ndev, 0, 64, &lo0, &per); // shard 0 has the common size (or n, when n is smaller)
    size_t common = (n + (size_t)ndev - 1) / (size_t)ndev;
    common = (common + 63) / 64 * 64;
    const size_t per_words = std::max<size_t>(common / 64, 1);
    // 1. per device: stream, buffers, input slice on its way, kernels, bit words
    for (int d = 0; d < ndev; d++) {
        MultiDev& m = S.d[d];
        m.dev = devs[d];
        lrm_shard_bounds(n, ndev, d, 64, &m.lo, &m.hi);
        const size_t cnt = m.hi - m.lo;
        HIP_TRY(hipSetDevice(m.dev), "hipSetDevice");
        HIP_TRY(hipStreamCreateWithFlags(&m.st, hipStreamNonBlocking), "hipStreamCreate");
        HIP_TRY(hipEventCreate(&m.a), "hipEventCreate");
        HIP_TRY(hipEventCreate(&m.b), "hipEventCreate");
        HIP_TRY(hipMalloc(&m.in, std::max<size_t>(cnt, 1) * 3 * sizeof(float)), "hipMalloc shard input");
        HIP_TRY(hipMalloc(&m.mask, std::max<size_t>(cnt, 1)), "hipMalloc shard mask");
        HIP_TRY(hipMalloc(&m.out, std::max<size_t>(cnt, 1) * 3 * sizeof(float)), "hipMalloc shard output");
        HIP_TRY(hipMalloc(&m.words, per_words * sizeof(uint64_t)), "hipMalloc shard words");
        HIP_TRY(hipMalloc(&m.gathered, per_words * (size_t)ndev * sizeof(uint64_t)), "hipMalloc gathered words");
        if (cnt) HIP_TRY(hipMemcpyAsync(m.in, xyz + 3 * m.lo, cnt * 3 * sizeof(float), hipMemcpyHostToDevice, m.st), "hipMemcpyAsync shard input");
        HIP_TRY(hipEventRecord(m.a, m.st), "hipEventRecord");
        if (cnt) {
            const int rc = launch_dist_aos_mode(2, static_cast<const float*>(m.in), cnt, *leg, quat_or_default(quat), L,
                                                static_cast<uint8_t*>(m.mask), static_cast<float*>(m.out), m.st);
            if (rc != LRM_OK) return rc;
        }
        HIP_TRY(hipEventRecord(m.b, m.st), "hipEventRecord");
        const unsigned pblocks = (unsigned)((per_words * 64 + 255) / 256);
        hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(pblocks), dim3(256), 0, m.st, static_cast<const uint8_t*>(m.mask), cnt,
                           static_cast<uint64_t*>(m.words), per_words);
        HIP_TRY(hipGetLastError(), "pack_mask_bits_kernel");
    }
    // 2. the exchange: every device receives every shard's words (grouped: one collective over all communicators)
    if (use_rccl) {
        int rc = g_rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (int d = 0; d < ndev && rc == 0 && he == hipSuccess; d++) {
            MultiDev& m = S.d[d];
            he = hipSetDevice(m.dev);
            if (he == hipSuccess) rc = g_rccl.AllGather(m.words, m.gathered, per_words, kNcclUint64, g_multi.comms[d], m.st);
        }
        const int rc2 = g_rccl.GroupEnd(); // on every path: a group left open would hang the process's next RCCL call
        HIP_TRY(he, "hipSetDevice");
        if (rc != 0 || rc2 != 0) return fail(LRM_ENODEV, (std::string("ncclAllGather: ") + g_rccl.GetErrorString(rc ? rc : rc2)).c_str());
    } else {
        MultiDev& m = S.d[0];
        HIP_TRY(hipSetDevice(m.dev), "hipSetDevice");
        HIP_TRY(hipMemcpyAsync(m.gathered, m.words, per_words * sizeof(uint64_t), hipMemcpyDeviceToDevice, m.st), "hipMemcpyAsync words");
    }
    // 3. results back: each device's slice of the bytes and vectors, the gathered words from the first device
    for (int d = 0; d < ndev; d++) {
        MultiDev& m = S.d[d];
        const size_t cnt = m.hi - m.lo;
        HIP_TRY(hipSetDevice(m.dev), "hipSetDevice");
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(mask_out + m.lo, m.mask, cnt, hipMemcpyDeviceToHost, m.st), "hipMemcpyAsync shard mask");
            HIP_TRY(hipMemcpyAsync(dxyz_out + 3 * m.lo, m.out, cnt * 3 * sizeof(float), hipMemcpyDeviceToHost, m.st), "hipMemcpyAsync shard output");
        }
        if (d == 0 && bits_out && n)
            HIP_TRY(hipMemcpyAsync(bits_out, m.gathered, (n + 63) / 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, m.st), "hipMemcpyAsync gathered words");
    }
    for (int d = 0; d < ndev; d++) {
        MultiDev& m = S.d[d];
        HIP_TRY(hipSetDevice(m.dev), "hipSetDevice");
        HIP_TRY(hipStreamSynchronize(m.st), "hipStreamSynchronize");
        float e = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, m.a, m.b), "hipEventElapsedTime");
        if (ms_per_dev) ms_per_dev[d] = (m.hi > m.lo) ? e : 0.f;
    }
    (void)per;
    (void)lo0;
    return LRM_OK;
}

} // extern "C"

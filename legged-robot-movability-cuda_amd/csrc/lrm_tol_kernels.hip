// lrm_tol_kernels.hip -- gfx950 kernels of the contract-tolerance mode (LRM_MODE_TOL, lrm_point_tol.h).
//
// Two launches per call, no host synchronisation between them and NO global atomics (a first version appended
// the doubtful points to one queue with one atomicAdd per wave: 42 000 same-address atomics per 1e7 points cost
// 0.42 ms, five times the arithmetic):
//   dist_tab_kernel   (clouds of >= 2e5 points) every point: FP32-FMA / v_rsq_f32 evaluation (no trigonometry, no IEEE
//                     sqrt / div) in which a yaw candidate's plane evaluation is one look-up in the plane table with
//                     deferred decisions (lrm_toltab.cpp) plus the reduced evaluation of what the cell names; both
//                     candidates in the owning lane, no barrier in the loop.  Reach mask + ballot bit words + distance
//                     vector.  A point with any decision inside its error band, or with an unanswered cell, is appended
//                     as a 16-byte record {index, x, y, z} to the SEGMENT of its workgroup (kSegCap slots, slot numbers
//                     from an LDS counter); the workgroup stores its count at the end.
//   dist_tol_staged_kernel  (smaller clouds, legs without a table) the same with the full plane evaluation: the more
//                     promising yaw candidate in the owning lane, the second one only where a lower bound cannot
//                     exclude it, compacted over the workgroup through LDS (two barriers per round).
//   tol_fixup_kernel  one workgroup per kSegPerWave segments: prefix sum of their counts, then the queued points
//                     (0.5 % of the cloud), two lanes per point, one yaw candidate each, through the bit-exact filtered
//                     code of LRM_MODE_FAST, overwriting their outputs.  A segment that overflowed (a cloud hugging a
//                     decision boundary) has ALL the points of its workgroup re-evaluated: slow, never wrong.  Counts are
//                     rewritten by every call: nothing to reset.
// Layout as lrm_kernels.hip: SoA coordinates, byte mask, ballot words, SoA distance field.  The per-leg block
// (LrmTolLeg, 1.5 KB) travels by value in the kernarg segment; its per-lane tables are staged in LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "lrm_launch.h"
#include "lrm_types.h"
#define LRM_FRESH(L) lrm_fresh(L)
#include "lrm_point.h"
#include "lrm_point_fast.h"
#include "lrm_point_tol.h"
#include "lrm_point_xtab.h"

#ifndef LRM_TOL_MIN_WAVES
#define LRM_TOL_MIN_WAVES 8 // the staged kernel needs 59 VGPRs; its barriers like occupancy: 6 / 7 / 8 waves -> 123.0 / 122.3 / 119.9 us per step
#endif
#ifndef LRM_TOL_LEG_IN_LDS
#define LRM_TOL_LEG_IN_LDS 0 // 1: the scalars too go through LDS (full-rate VGPR-only operands by the issue-class table, but measured 148 us against 114: the extra lgkmcnt waits cost more than the half-rate operands)
#endif
#ifndef LRM_TOL_BLOCK
#define LRM_TOL_BLOCK 256
#endif
#ifndef LRM_TOL_STAGED
#define LRM_TOL_STAGED 1
#endif
#ifndef LRM_TOL_SEG_PER_WAVE
#define LRM_TOL_SEG_PER_WAVE 8 // segments one fix-up workgroup compacts: ~20 queued points at the usual 0.4 % of doubt, one pass of its first wave
#endif
#ifndef LRM_TOL_PREFETCH
#define LRM_TOL_PREFETCH 1
#endif
#ifndef LRM_TOL_GRID_MULT
#define LRM_TOL_GRID_MULT 8
#endif
#ifndef LRM_TAB_FIX_SEGS
#define LRM_TAB_FIX_SEGS 8 // segments per fix-up workgroup: ~60 queued points at the usual 0.5 % of doubt, 1000 workgroups (4 segments, 2000 workgroups: 3 us slower)
#endif

#if defined(LRM_FIX_TRACE)
// timing experiment (tools/fix_trace.py): s_memrealtime stamps (100 MHz) of the phases of the fix-up waves and the
// end of every workgroup of the main kernel
__device__ uint64_t g_fix_trace[8 * 4096];
__device__ uint64_t g_main_trace[2 * 32768];
extern "C" int lrm_dbg_fix_trace(uint64_t* fix_out, uint64_t* main_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(fix_out, HIP_SYMBOL(g_fix_trace), sizeof(g_fix_trace)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(main_out, HIP_SYMBOL(g_main_trace), sizeof(g_main_trace)) != hipSuccess) return -1;
    return 0;
}
#define LRM_TRACE_FIX(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_fix_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LRM_TRACE_FIX(k) do {} while (0)
#endif

namespace {

constexpr int kBlock = LRM_TOL_BLOCK;
#ifndef LRM_TOL_FIX_BLOCK
#define LRM_TOL_FIX_BLOCK 128 // 64 / 128 / 256 threads with 8 segments: config-2 cube 0.1158 (4 segments) / 0.1153 / 0.1174 ms per step, planar bench grid 0.3635 / 0.1292 / 0.1279
#endif
constexpr int kFixBlock = LRM_TOL_FIX_BLOCK; // threads of a fix-up workgroup: with the usual handful of queued points only its first wave works,
                                             // a cloud that hugs decision boundaries (the planar bench grid: 7 % in doubt) keeps all of them busy
constexpr int kSegCap = LRM_TOL_SEG_CAP;         // doubt slots per workgroup of dist_tol_kernel
constexpr int kSegPerWave = LRM_TOL_SEG_PER_WAVE;                 // segments one fix-up wave compacts

struct KernargTol { // leading parameters of dist_tol_kernel, in order
    const float *x, *y, *z;
    size_t n;
    LrmTolLeg L;
};
struct KernargFix {
    const float *x, *y, *z;
    size_t n;
    LrmCompiledLeg L;
};
constexpr unsigned kTolLegArg = (unsigned)offsetof(KernargTol, L), kFixLegArg = (unsigned)offsetof(KernargFix, L);
static_assert(kTolLegArg == 32 && kFixLegArg == 32, "kernarg layout");

// element at BYTE offset `off` (32 bits, zero-extended) of the global array p.  With a wave-uniform p and an offset the
// optimiser cannot see through (lrm_opaque below) the access is "scalar base + 32-bit vector offset"; otherwise the
// loop-invariant base + thread part is hoisted into a 64-bit VGPR pair per array and the round offset is added with one
// v_lshl_add_u64 per access.
#define LRM_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ LRM_GLOBAL T& lrm_at(T* p, uint32_t off) {
    using C = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    return *(LRM_GLOBAL T*)((LRM_GLOBAL C*)p + off);
}
__device__ __forceinline__ uint32_t lrm_opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

// LRM_MODE_TOL_REL: vectors shorter than this (mm) are computed by the bit-exact code.  The tolerance arithmetic's absolute error
// grows with the coordinates (~10 ulp of |p|_1 + body: measured <= 0.011 of the decision band on the config-2 cloud and in the
// campaigns of tools/stress_tol.py --rel); 1e-5 of LRM_TOL_REL_BANDS bands is 0.02 of a band.
__device__ __forceinline__ float lrm_tol_rel_threshold(const LrmTolLeg& L, const LrmVec3& p) {
    const float band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), L.band_slope, L.band_base);
    return fmaxf(LRM_TOL_REL_MM, LRM_TOL_REL_BANDS * band);
}

struct TolLds {
    LrmTolLeg::Circle circ[16];
    LrmCircle feat[LRM_TOL_FEATS];
};

// kOp 1: distance + optional validity byte; kOp 2: reach mask (+ bit words) + distance.  In this mode the two
// flags are the same function of the point wherever no decision is in doubt.

// A queued point travels as a 16-byte record {index, x, y, z}: the fix-up then reads its coordinates with the queue
// (one or two lines per segment) instead of three scattered 4-byte loads per point (384 B of HBM traffic each), and has
// one dependent memory round trip less.
struct QueueRec {
    uint32_t i;
    float x, y, z;
};
static_assert(sizeof(QueueRec) == 16, "one 16-byte store / load per queued point");

// ------------------------------------------------------------------------------------------------------------
// Staged variant (LRM_TOL_STAGED): every lane evaluates the more promising yaw candidate of its point; the lanes
// whose second candidate can still win (lower bound on its norm, lrm_tol_need_second: ~25 % of a random cloud)
// hand its PLANE evaluation -- four floats -- to the workgroup through LDS, where it runs compacted (one wave's
// worth per 256 points instead of all four waves), and take (du, dz, valid, doubt) back.  Everything else, and every
// global access, stays with the owning lane: coalesced as before.  Two barriers per iteration.
// ------------------------------------------------------------------------------------------------------------
// kAoS: x is the float3 array of the apply_kernel boundary (y, z unused), dx the float3 output (dy, dz unused)
template <int kOp, bool kAoS = false>
__global__ __launch_bounds__(kBlock, LRM_TOL_MIN_WAVES) void dist_tol_staged_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmTolLeg L_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, QueueRec* __restrict__ queue, uint32_t* __restrict__ counts, uint32_t selftest) {
    __shared__ TolLds s_tab;
    __shared__ uint32_t s_qn;
    __shared__ uint32_t s_cnt[2];
    __shared__ float4 s_task[kBlock]; // {u, z, band, tau} of a pending second plane evaluation
    __shared__ float4 s_res[kBlock];  // {du, dz, valid, doubt bits}
    const LrmTolLeg& L = lrm_kernarg<LrmTolLeg>(kTolLegArg);
#if LRM_TOL_PREFETCH
    // The point of the NEXT round is loaded while this round's second-candidate stage runs, the first one in front of the
    // table staging: the load latency leaves the critical path of the workgroup (three more VGPRs: 64, the limit of 8 waves).
    LrmVec3 p_next{0.f, 0.f, 0.f};
    {
        const uint32_t i0 = blockIdx.x * kBlock + threadIdx.x;
        const size_t rb0 = (size_t)blockIdx.x * kBlock;
        const uint32_t to = lrm_opaque(threadIdx.x * 4u);
        if (i0 < n) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb0, 3u * to), lrm_at(x + 3 * rb0, 3u * to + 4u), lrm_at(x + 3 * rb0, 3u * to + 8u)}
                                  : LrmVec3{lrm_at(x + rb0, to), lrm_at(y + rb0, to), lrm_at(z + rb0, to)};
    }
#endif
    {
        const float* csrc = reinterpret_cast<const float*>(&L.circ[0][0]);
        const float* fsrc = reinterpret_cast<const float*>(&L.feat[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.circ) / 4); i += kBlock) reinterpret_cast<float*>(s_tab.circ)[i] = csrc[i];
        if (threadIdx.x < (int)(sizeof(s_tab.feat) / 4)) reinterpret_cast<float*>(s_tab.feat)[threadIdx.x] = fsrc[threadIdx.x];
        if (threadIdx.x == 0) { s_qn = 0; s_cnt[0] = 0; s_cnt[1] = 0; }
        __syncthreads();
    }
    const LrmTolTables T{s_tab.circ, s_tab.feat};
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t n_pad = (uint32_t)((n + kBlock - 1) / kBlock) * kBlock; // whole workgroups iterate together (barriers below)
    QueueRec* seg = queue + (size_t)blockIdx.x * kSegCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t toff0 = threadIdx.x * 4u;
#if defined(LRM_FIX_TRACE)
    if (threadIdx.x == 0 && blockIdx.x < 32768) g_main_trace[blockIdx.x * 2] = wall_clock64();
#endif
    uint32_t round = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride, round++) {
        const bool live = i < n;
        // Addresses as (wave-uniform base of this round) + threadIdx.x: the base lives in SGPRs and advances on the
        // scalar unit, the vector offset is a loop invariant -- no 64-bit address arithmetic on the vector unit
        const size_t rbase = (size_t)blockIdx.x * kBlock + (size_t)round * stride;
        const uint32_t toff = lrm_opaque(toff0), tid_o = lrm_opaque(threadIdx.x);
#if LRM_TOL_PREFETCH
        LrmVec3 p = p_next;
#else
        LrmVec3 p{0.f, 0.f, 0.f};
        if (live) p = kAoS ? LrmVec3{lrm_at(x + 3 * rbase, 3u * toff), lrm_at(x + 3 * rbase, 3u * toff + 4u), lrm_at(x + 3 * rbase, 3u * toff + 8u)}
                           : LrmVec3{lrm_at(x + rbase, toff), lrm_at(y + rbase, toff), lrm_at(z + rbase, toff)};
#endif
        // ---- A: first candidate ----
        const LrmTolPoint S = lrm_tol_prologue(L, p);
        const float short_mm = fmaxf(LRM_TOL_REL_MM, LRM_TOL_REL_BANDS * S.band); // LRM_MODE_TOL_REL's threshold (lrm_tol_rel_threshold)
        uint32_t lu = S.lu;
        float du, dzz;
        bool valid;
        lrm_tol_plane(L, T, S.u0, S.z, S.band, S.tau, du, dzz, valid, lu);
        const LrmTolCand A = lrm_tol_candidate(S, false, du, dzz, valid, lu);
        const bool need = live && lrm_tol_need_second(L, S, A);
        // ---- hand the pending second evaluations to the workgroup: slots from an LDS counter (one atomic per wave;
        // two counters alternate so that resetting one never races with the round that uses the other) ----
        const uint64_t nm = __ballot(need);
        uint32_t base = 0;
        if (lane == 0 && nm) base = atomicAdd(&s_cnt[round & 1u], (uint32_t)__popcll(nm));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t slot = base + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        if (need) s_task[slot] = make_float4(S.u1, S.z, S.band, S.tau);
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0; // nobody touches the other counter during this round
#if !defined(LRM_TOL_EXP_NOBARRIER) // timing experiments only (wrong results)
        __syncthreads();
#endif
#if LRM_TOL_PREFETCH
        {
            const uint32_t i_next = i + stride;
            const size_t rb_next = rbase + stride;
            p_next = LrmVec3{0.f, 0.f, 0.f};
            if (i_next < n) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb_next, 3u * toff), lrm_at(x + 3 * rb_next, 3u * toff + 4u), lrm_at(x + 3 * rb_next, 3u * toff + 8u)}
                                          : LrmVec3{lrm_at(x + rb_next, toff), lrm_at(y + rb_next, toff), lrm_at(z + rb_next, toff)};
        }
#endif
#if defined(LRM_TOL_EXP_NOB)
        const uint32_t total = 0;
#else
        const uint32_t total = s_cnt[round & 1u];
#endif
        // ---- B: the compacted second plane evaluations; the lanes that take them rotate from round to round ----
        {
            uint32_t t = threadIdx.x + (round % (uint32_t)(kBlock / 64)) * 64u; // (threadIdx.x + round * 64) mod kBlock
            if (t >= (uint32_t)kBlock) t -= (uint32_t)kBlock;
            if (t < total) {
                const float4 task = s_task[t];
                float bu, bz;
                bool bvalid;
                uint32_t bd = 0;
                lrm_tol_plane(L, T, task.x, task.y, task.z, task.w, bu, bz, bvalid, bd);
                s_res[t] = make_float4(bu, bz, bvalid ? 1.f : 0.f, lrm_u2f(bd));
            }
        }
#if !defined(LRM_TOL_EXP_NOBARRIER)
        __syncthreads();
#endif
        // ---- C: back with the owner ----
        LrmTolCand B = A;
        if (need) {
            const float4 r = s_res[slot];
            uint32_t bd = lrm_f2u(r.w);
            B = lrm_tol_candidate(S, true, r.x, r.y, r.z != 0.f, bd);
            lu |= bd;
        }
        uint32_t doubt = lu;
        const bool m = lrm_tol_finish(L, S, A, need, B, p, doubt) && live;
        doubt = live ? ((doubt & 0xffffu) | (selftest & 1u)) : 0u; // selftest: every point goes to the fix-up
        if (selftest & LRM_TOLF_SHORT) { // LRM_MODE_TOL_REL (wave-uniform): a vector shorter than the threshold comes from the bit-exact code
            const float nn = __builtin_fmaf(p.x, p.x, __builtin_fmaf(p.y, p.y, p.z * p.z));
            doubt |= (live && !(nn >= short_mm * short_mm)) ? 1u : 0u;
        }
        if (live) {
            if (kAoS) {
                lrm_at(dx + 3 * rbase, 3u * toff) = p.x;
                lrm_at(dx + 3 * rbase, 3u * toff + 4u) = p.y;
                lrm_at(dx + 3 * rbase, 3u * toff + 8u) = p.z;
            } else {
                lrm_at(dx + rbase, toff) = p.x;
                lrm_at(dy + rbase, toff) = p.y;
                lrm_at(dz + rbase, toff) = p.z;
            }
            if (mask) lrm_at(mask + rbase, tid_o) = m;
        }
        if (bits && i < (uint32_t)((n + 63) & ~(size_t)63)) { // wave-uniform
            const uint64_t w = __ballot(m);
            if (lane == 0) lrm_at(bits + (rbase >> 6), lrm_opaque((uint32_t)wave * 8u)) = w;
        }
        const uint64_t dm = __ballot(doubt != 0);
        if (dm) {
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&s_qn, (uint32_t)__popcll(dm));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            if (doubt) {
                const uint32_t qs = qb + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (qs < (uint32_t)kSegCap) { // beyond: the count tells the fix-up to redo the workgroup.  The point comes back from the L2 (loaded a round ago)
                    const LrmVec3 q = kAoS ? LrmVec3{lrm_at(x + 3 * rbase, 3u * toff), lrm_at(x + 3 * rbase, 3u * toff + 4u), lrm_at(x + 3 * rbase, 3u * toff + 8u)}
                                           : LrmVec3{lrm_at(x + rbase, toff), lrm_at(y + rbase, toff), lrm_at(z + rbase, toff)};
                    seg[qs] = QueueRec{i, q.x, q.y, q.z};
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_qn;
#if defined(LRM_FIX_TRACE)
    if (threadIdx.x == 0 && blockIdx.x < 32768) g_main_trace[blockIdx.x * 2 + 1] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Table variant (LrmTolTabHeader, lrm_toltab.cpp): the plane evaluation of a yaw candidate is ONE look-up in the
// plane table with deferred decisions plus the reduced evaluation of what the cell names (lrm_tol_plane_tab: two
// clamp targets, one circle's validity), and the table's lower bounds decide which candidate is evaluated -- the
// second one in 0.3 % of the points (17 % of the waves of a random cloud).  No workgroup stage, no barriers in the
// loop.  A point whose cell carries no answer is queued with the doubtful ones for the bit-exact fix-up.
//   Memory: the cell codes (32 KB coarse per grid + 512 B per refined cell) stay in global memory -- read by every
// workgroup of every launch, they live in the L1 / L2; the rows (1.5 KB) and the inner grid's bounds (16 KB) are staged
// in LDS by every workgroup: a wave's scattered look-up costs 8 cycles of its CU there, 42 in a table the L1 holds,
// 110-140 in one of 128-512 KB (tools/gather_rates.hip) -- with 16 mm bound cells in global memory (256 KB with the
// codes) the kernel was bound by exactly that, at 80 us.
//   Tried (profiles/r03_ab_tab_persistent.txt): bounds per coarse cell (64 KB of LDS) in PERSISTENT workgroups of 768
// threads, two per CU, work items taken from an LDS counter: 83-87 us.  The hardware issues the oldest wave first, so
// the first workgroup of a CU runs ahead and leaves the second one the CU at half occupancy for the last 20 us (wave
// priorities that fall with a workgroup's progress: -4 %); with small workgroups the dispatcher back-fills instead.
// ------------------------------------------------------------------------------------------------------------
#ifndef LRM_TAB_MIN_WAVES
#define LRM_TAB_MIN_WAVES 7 // at 8 waves (64 VGPRs) the compiler spills
#endif
#ifndef LRM_TAB_GRID_MULT
#define LRM_TAB_GRID_MULT 8
#endif
#ifndef LRM_SHORT_MIN_WAVES
#define LRM_SHORT_MIN_WAVES 6 // LRM_MODE_TOL_REL's variant carries the info word, the in-loop flush and the replay tail: 80 VGPRs + 12 bytes of scratch per lane (the flush) at 6 waves: 92.2 us; 96 VGPRs at 5 waves: 96.2
#endif
#ifndef LRM_TAB_PREFETCH2
#define LRM_TAB_PREFETCH2 0 // two rounds of loads in flight per wave (6 waves/SIMD): 80 / 123 / 933 us at 1e7 / 1.25e7 / 1e8 points against 75 / 119 / 891 (profiles/r04_ab_prefetch_nt.txt)
#endif
#ifndef LRM_TAB_NT_STORE
#define LRM_TAB_NT_STORE 1 // the distance field is written once and never read by the kernels: non-temporal stores.  Nothing at 1e7 points (71.7 -> 70.8 us:
                           // inputs and outputs fit the 256 MiB Infinity Cache together) or 1e8 (822 -> 827), but the 1.25e7-point share of the 1e8 cloud --
                           // 312 MB cycling through a 256 MiB cache -- 117.7 -> 85.9 us: ordinary stores evict the inputs the next step re-reads
#endif
#ifndef LRM_TAB_NT_LOAD
#define LRM_TAB_NT_LOAD 0 // non-temporal loads of the points as well: 72 -> 89 / 91 -> 109 / 800 -> 809 us at 1e7 / 1.25e7 / 1e8 points: no
#endif
struct TabLds {
    LrmTabRow rows[32];
    LrmTabVRow vrows[32];
};
constexpr int kTabSegCap = LRM_TOL_TAB_SEG_CAP; // doubt slots per workgroup of dist_tab_kernel
#ifndef LRM_SHORT_CAP
#define LRM_SHORT_CAP 64 // LDS slots per wave for the short vectors of LRM_MODE_TOL_REL (~15 expected over a wave's five rounds; a segment without room for a round's records is replayed on the spot)
#endif
// The replayed vectors leave with non-temporal stores as well: ordinary ones make the L2 fetch the rest of each line (the tolerance
// vectors of the neighbours, streamed out moments before) from memory: 92.97 -> 88.34 us per 1e7 points, 128.2 -> 108.1 us per 1.25e7
// (profiles/r04_ab_rel_fixup.txt; the fix-up's few stores gain nothing from it: 14.2 -> 14.3 us, 14.6 -> 16.5 at 1.25e7)
#ifndef LRM_REPLAY_NT_STORE
#define LRM_REPLAY_NT_STORE 1
#endif
#ifndef LRM_SHORT_TAIL_WAIT
#define LRM_SHORT_TAIL_WAIT 1
#endif
constexpr int kShortCap = LRM_SHORT_CAP;
static_assert(kShortCap >= 64, "an emptied segment holds one round of a wave");
__device__ __forceinline__ void wave_lds_fence_tol() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
constexpr int kTabBoundVecs = LRM_TT_NB * LRM_TT_NB * 4 / 16; // 16-byte pieces of the inner grid's bounds
static_assert(kTabBoundVecs % kBlock == 0, "every thread stages the same number of pieces");
// kShort: LRM_MODE_TOL_REL (a template argument: as a run-time flag the compiler computes the norm for every point of every mode)
template <int kOp, bool kAoS = false, bool kShort = false>
__global__ __launch_bounds__(kBlock, kShort ? LRM_SHORT_MIN_WAVES : LRM_TAB_MIN_WAVES) void dist_tab_kernel( // (kShort carries the info word and the replay tail: spills at 7 waves)
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmTolLeg L_kernarg, const LrmXtabLeg X_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const uint8_t* __restrict__ tab, QueueRec* __restrict__ queue,
    uint32_t* __restrict__ counts, uint32_t selftest) {
    __shared__ TabLds s_tab;
    __shared__ uint32_t s_bound[LRM_TT_NB * LRM_TT_NB];
    __shared__ uint32_t s_qn;
    // LRM_MODE_TOL_REL: the short vectors of this workgroup (4-5 % of a random cloud: ~32 over its life) wait in LDS, a segment per
    // wave (the count in a scalar register: no atomic), five words per record; the workgroup's first wave replays them at the end
    __shared__ uint32_t s_short[kShort ? 5 * (kBlock / 64) * kShortCap : 1];
    __shared__ uint32_t s_wcnt[kBlock / 64];
    const LrmTolLeg& L = lrm_kernarg<LrmTolLeg>(kTolLegArg);
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab);
    LrmVec3 p_next{0.f, 0.f, 0.f}; // the first point in front of the table staging
#if LRM_TAB_PREFETCH2
    LrmVec3 p_next2{0.f, 0.f, 0.f}; // and the second round's: two rounds of loads in flight per wave (clouds beyond the Infinity Cache wait for HBM)
#endif
    {
        const uint32_t i0 = blockIdx.x * kBlock + threadIdx.x;
        const size_t rb0 = (size_t)blockIdx.x * kBlock;
        const uint32_t to = lrm_opaque(threadIdx.x * 4u);
        if (i0 < n) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb0, 3u * to), lrm_at(x + 3 * rb0, 3u * to + 4u), lrm_at(x + 3 * rb0, 3u * to + 8u)}
                                  : LrmVec3{lrm_at(x + rb0, to), lrm_at(y + rb0, to), lrm_at(z + rb0, to)};
#if LRM_TAB_PREFETCH2
        const size_t rb1 = rb0 + (size_t)gridDim.x * kBlock;
        if (i0 + gridDim.x * kBlock < n) p_next2 = kAoS ? LrmVec3{lrm_at(x + 3 * rb1, 3u * to), lrm_at(x + 3 * rb1, 3u * to + 4u), lrm_at(x + 3 * rb1, 3u * to + 8u)}
                                                       : LrmVec3{lrm_at(x + rb1, to), lrm_at(y + rb1, to), lrm_at(z + rb1, to)};
#endif
    }
    {
        static_assert(sizeof(TabLds) == sizeof(hd->rows) + sizeof(hd->vrows) && sizeof(TabLds) % 16 == 0, "rows | vrows");
        const uint4* src = reinterpret_cast<const uint4*>(&hd->rows[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(TabLds) / 16); i += kBlock) reinterpret_cast<uint4*>(&s_tab)[i] = src[i];
        // the bounds: a thread's loads in flight together (a loop of load - wait - store costs one L2 round trip per pass)
        const uint4* bsrc = reinterpret_cast<const uint4*>(tab + sizeof(LrmTolTabHeader) + 2 * (size_t)hd->bound_off[0]);
        uint4 v[kTabBoundVecs / kBlock];
#pragma unroll
        for (int k = 0; k < kTabBoundVecs / kBlock; k++) v[k] = bsrc[k * kBlock + (int)threadIdx.x];
#pragma unroll
        for (int k = 0; k < kTabBoundVecs / kBlock; k++) reinterpret_cast<uint4*>(s_bound)[k * kBlock + (int)threadIdx.x] = v[k];
        if (threadIdx.x == 0) s_qn = 0;
        __syncthreads();
    }
    const LrmTolTabView G = lrm_toltab_view(tab, s_tab.rows, s_tab.vrows, s_bound, L.r_outer);
#if defined(LRM_FIX_TRACE)
    if (threadIdx.x == 0 && blockIdx.x < 32768) g_main_trace[blockIdx.x * 2] = wall_clock64();
#endif
    // LRM_MODE_TOL_REL: records [k, count) of a segment, one lane each: the tolerance evaluation took every DECISION for them (none
    // in doubt); lrm_xtab_replay recomputes the winner's VALUE chain with the reference's own operations (no table look-ups, no
    // bands): these vectors are bit-identical to the reference's.
    const LrmXtabLeg& X = lrm_kernarg<LrmXtabLeg>(kTolLegArg + (unsigned)sizeof(LrmTolLeg));
    auto replay_records = [&](const uint32_t* segment, uint32_t count, uint32_t k) {
        if (k < count) {
            const uint32_t* r = segment + 5u * k;
            const size_t i = r[0];
            LrmVec3 q{lrm_u2f(r[1]), lrm_u2f(r[2]), lrm_u2f(r[3])};
            lrm_xtab_replay(lrm_fresh(X), s_tab.rows, q, r[4]);
#if defined(LRM_EXP_NOSCATTER) // timing experiment (wrong results): the replayed vectors are computed but (practically) never stored
            if (!(q.x != q.x)) return;
#endif
            if (kAoS) {
                dx[3 * i] = q.x;
                dx[3 * i + 1] = q.y;
                dx[3 * i + 2] = q.z;
            } else {
#if LRM_REPLAY_NT_STORE
                __builtin_nontemporal_store(q.x, dx + i);
                __builtin_nontemporal_store(q.y, dy + i);
                __builtin_nontemporal_store(q.z, dz + i);
#else
                dx[i] = q.x;
                dy[i] = q.y;
                dz[i] = q.z;
#endif
            }
        }
    };
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t n_pad = (uint32_t)((n + 63) & ~(size_t)63); // whole waves iterate together (ballots below)
    const uint32_t n32 = (uint32_t)n; // the C ABI sends clouds of 0xc0000000 points and more to the bit-exact kernels: indices fit 32 bits
    QueueRec* seg = queue + (size_t)blockIdx.x * kTabSegCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane(wave); // (the compiler cannot know that threadIdx.x >> 6 is uniform)
    uint32_t wq = 0;
    const uint32_t toff0 = threadIdx.x * 4u;
    uint32_t round = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride, round++) {
        const bool live = i < n32;
        const size_t rbase = (size_t)blockIdx.x * kBlock + (size_t)round * stride;
        const uint32_t toff = lrm_opaque(toff0), tid_o = lrm_opaque(threadIdx.x);
        LrmVec3 p = p_next;
#if LRM_TAB_PREFETCH2
        {   // the points of the next two rounds are in flight while this one is evaluated
            const uint32_t i_next = i + 2u * stride;
            const size_t rb_next = rbase + 2 * (size_t)stride;
            p_next = p_next2;
            p_next2 = LrmVec3{0.f, 0.f, 0.f};
            if (i_next < n32 && i_next > i) p_next2 = kAoS ? LrmVec3{lrm_at(x + 3 * rb_next, 3u * toff), lrm_at(x + 3 * rb_next, 3u * toff + 4u), lrm_at(x + 3 * rb_next, 3u * toff + 8u)}
                                                          : LrmVec3{lrm_at(x + rb_next, toff), lrm_at(y + rb_next, toff), lrm_at(z + rb_next, toff)};
        }
#else
        {   // the next round's point is in flight while this one is evaluated
            const uint32_t i_next = i + stride;
            const size_t rb_next = rbase + stride;
            p_next = LrmVec3{0.f, 0.f, 0.f};
#if LRM_TAB_NT_LOAD
            if (i_next < n32) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb_next, 3u * toff), lrm_at(x + 3 * rb_next, 3u * toff + 4u), lrm_at(x + 3 * rb_next, 3u * toff + 8u)}
                                          : LrmVec3{__builtin_nontemporal_load(&lrm_at(x + rb_next, toff)), __builtin_nontemporal_load(&lrm_at(y + rb_next, toff)), __builtin_nontemporal_load(&lrm_at(z + rb_next, toff))};
#else
            if (i_next < n32) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb_next, 3u * toff), lrm_at(x + 3 * rb_next, 3u * toff + 4u), lrm_at(x + 3 * rb_next, 3u * toff + 8u)}
                                          : LrmVec3{lrm_at(x + rb_next, toff), lrm_at(y + rb_next, toff), lrm_at(z + rb_next, toff)};
#endif
        }
#endif
        uint32_t doubt = 0;
        const LrmVec3 p_in = p; // kept for the queue record (three registers; re-loading it cost a pushing wave an L2 round trip)
        uint32_t info = 0; // kShort: what the evaluation decided (the tail replays the winner's value chain from it)
#if defined(LRM_EXP_NOINFO) // timing experiment (wrong results): the decisions are not packed
        const bool m = lrm_tab_point<false>(L, G, p, doubt, &info) && live;
#else
        const bool m = lrm_tab_point<kShort>(L, G, p, doubt, &info) && live;
#endif
        doubt = live ? ((doubt & 0xffffu) | (selftest & 1u)) : 0u; // selftest: every point goes to the fix-up
        bool is_short = false;
        if (kShort) { // LRM_MODE_TOL_REL: a vector shorter than the threshold (and not in doubt) gets its value chain replayed strictly
            const float nn = __builtin_fmaf(p.x, p.x, __builtin_fmaf(p.y, p.y, p.z * p.z));
            const float short_mm = lrm_tol_rel_threshold(L, p_in); // (from the point kept for the queue record: nothing more stays live across the evaluation)
            is_short = live && doubt == 0u && !(nn >= short_mm * short_mm);
        }
        if (live) {
            if (kAoS) {
                lrm_at(dx + 3 * rbase, 3u * toff) = p.x;
                lrm_at(dx + 3 * rbase, 3u * toff + 4u) = p.y;
                lrm_at(dx + 3 * rbase, 3u * toff + 8u) = p.z;
            } else {
#if LRM_TAB_NT_STORE // streaming stores of the field: see LRM_TAB_NT_STORE above
                __builtin_nontemporal_store(p.x, &lrm_at(dx + rbase, toff));
                __builtin_nontemporal_store(p.y, &lrm_at(dy + rbase, toff));
                __builtin_nontemporal_store(p.z, &lrm_at(dz + rbase, toff));
#else
                lrm_at(dx + rbase, toff) = p.x;
                lrm_at(dy + rbase, toff) = p.y;
                lrm_at(dz + rbase, toff) = p.z;
#endif
            }
            if (mask) lrm_at(mask + rbase, tid_o) = m;
        }
        if (bits) {
            const uint64_t w = __ballot(m);
            if (lane == 0) lrm_at(bits + (rbase >> 6), lrm_opaque((uint32_t)wave * 8u)) = w;
        }
        if (kShort) { // the wave's LDS segment; a record that finds no room goes to the doubt queue instead (the filtered code takes it)
            const uint64_t sm = __ballot(is_short);
#if defined(LRM_EXP_NOPUSH) // timing experiment (wrong results): the short vectors are never recorded
            if (false) {
#else
            if (sm) {
#endif
                // A segment without room for this round's records is replayed on the spot by its own wave (a cloud of reachable
                // points only has 35 % short vectors: sending what does not fit to the doubt queue made its workgroups overflow and
                // the filtered code redo them whole, 0.71 ms per 1e7 points; profiles/r04_bench_a.json).  The earlier stores of the
                // same points came from this wave: they are complete before the replayed ones go out.
                if (wq + (uint32_t)__popcll(sm) > (uint32_t)kShortCap) { // wave-uniform
                    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores have reached the L2
                    replay_records(s_short + 5u * wave_s * (uint32_t)kShortCap, wq, (uint32_t)lane);
                    wave_lds_fence_tol();
                    wq = 0;
                }
                const uint32_t qs = wq + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
                if (is_short) { // (at most 64 records per round, kShortCap >= 64: they fit)
                    uint32_t* r = s_short + 5u * (wave_s * (uint32_t)kShortCap + qs);
                    r[0] = i;
                    r[1] = lrm_f2u(p_in.x);
                    r[2] = lrm_f2u(p_in.y);
                    r[3] = lrm_f2u(p_in.z);
                    r[4] = info;
                }
                wq += (uint32_t)__popcll(sm);
            }
        }
        const uint64_t dm = __ballot(doubt != 0); // queue A: per-workgroup segment, slots from an LDS counter
        if (dm) {
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&s_qn, (uint32_t)__popcll(dm));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            if (doubt) {
                const uint32_t qs = qb + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (qs < (uint32_t)kTabSegCap) seg[qs] = QueueRec{i, p_in.x, p_in.y, p_in.z}; // beyond: the count tells the fix-up to redo the workgroup
            }
        }
    }
    if (kShort && lane == 0) s_wcnt[wave_s] = wq;
#if LRM_SHORT_TAIL_WAIT
    // The replayed vectors below overwrite tolerance vectors stored by OTHER waves of this workgroup: every wave first waits until
    // its own stores have been acknowledged by the L2 (the workgroup-scope fence of __syncthreads does not: one CU, one L1 -- it
    // relies on the CU's stores to one address reaching the L2 in issue order).
    if (kShort) __builtin_amdgcn_s_waitcnt(0);
#endif
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_qn;
#if defined(LRM_EXP_NOTAIL) // timing experiment (wrong results): the short vectors are recorded but never replayed
    if (false) {
#else
    if (kShort) {
#endif
        // ---- the tail: what is left in the four segments, compacted over the workgroup's waves (usually its first wave alone) ----
        static_assert(kBlock / 64 == 4, "four wave segments");
        const uint32_t c0 = s_wcnt[0], c1 = c0 + s_wcnt[1], c2 = c1 + s_wcnt[2], total = c2 + s_wcnt[3];
        for (uint32_t k0 = wave_s * 64u; k0 < total; k0 += (uint32_t)kBlock) { // wave-uniform
            const uint32_t k = k0 + (uint32_t)lane;
            const uint32_t w = (k >= c0 ? 1u : 0u) + (k >= c1 ? 1u : 0u) + (k >= c2 ? 1u : 0u);
            const uint32_t first = w == 0u ? 0u : (w == 1u ? c0 : (w == 2u ? c1 : c2));
            replay_records(s_short + 5u * w * (uint32_t)kShortCap, k < total ? (k - first) + 1u : 0u, k - first);
        }
    }
#if defined(LRM_FIX_TRACE)
    if (threadIdx.x == 0 && blockIdx.x < 32768) g_main_trace[blockIdx.x * 2 + 1] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------------------
// LRM_MODE_FAST through the plane table (round 4; lrm_point_xtab.h): the DECISIONS of distance_global come from the table
// exactly as in dist_tab_kernel, the VALUES from the reference's own operations in the reference's own order -- the outputs
// are bit-identical to LRM_MODE_STRICT.  A point with a decision inside its band (0.3 % of a random cloud) is queued for
// tol_fixup_kernel, i.e. the filtered code this kernel replaces for everybody else.  Same launch shape, staging and queue
// as dist_tab_kernel.
// ------------------------------------------------------------------------------------------------------------
#ifndef LRM_XTAB_MIN_WAVES
#define LRM_XTAB_MIN_WAVES 5
#endif
struct KernargXtab {
    const float *x, *y, *z;
    size_t n;
    LrmXtabLeg X;
};
constexpr unsigned kXtabLegArg = (unsigned)offsetof(KernargXtab, X);
static_assert(kXtabLegArg == 32, "kernarg layout");
template <int kOp, bool kAoS = false>
__global__ __launch_bounds__(kBlock, LRM_XTAB_MIN_WAVES) void dist_xtab_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmXtabLeg X_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const uint8_t* __restrict__ tab, QueueRec* __restrict__ queue,
    uint32_t* __restrict__ counts, uint32_t selftest) {
    __shared__ TabLds s_tab;
    __shared__ uint32_t s_bound[LRM_TT_NB * LRM_TT_NB];
    __shared__ uint32_t s_qn;
    const LrmXtabLeg& X = lrm_kernarg<LrmXtabLeg>(kXtabLegArg);
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab);
    LrmVec3 p_next{0.f, 0.f, 0.f};
    {
        const uint32_t i0 = blockIdx.x * kBlock + threadIdx.x;
        const size_t rb0 = (size_t)blockIdx.x * kBlock;
        const uint32_t to = lrm_opaque(threadIdx.x * 4u);
        if (i0 < n) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb0, 3u * to), lrm_at(x + 3 * rb0, 3u * to + 4u), lrm_at(x + 3 * rb0, 3u * to + 8u)}
                                  : LrmVec3{lrm_at(x + rb0, to), lrm_at(y + rb0, to), lrm_at(z + rb0, to)};
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(&hd->rows[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(TabLds) / 16); i += kBlock) reinterpret_cast<uint4*>(&s_tab)[i] = src[i];
        const uint4* bsrc = reinterpret_cast<const uint4*>(tab + sizeof(LrmTolTabHeader) + 2 * (size_t)hd->bound_off[0]);
        uint4 v[kTabBoundVecs / kBlock];
#pragma unroll
        for (int k = 0; k < kTabBoundVecs / kBlock; k++) v[k] = bsrc[k * kBlock + (int)threadIdx.x];
#pragma unroll
        for (int k = 0; k < kTabBoundVecs / kBlock; k++) reinterpret_cast<uint4*>(s_bound)[k * kBlock + (int)threadIdx.x] = v[k];
        if (threadIdx.x == 0) s_qn = 0;
        __syncthreads();
    }
    const LrmTolTabView G = lrm_toltab_view(tab, s_tab.rows, s_tab.vrows, s_bound, X.r_outer);
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t n_pad = (uint32_t)((n + 63) & ~(size_t)63);
    const uint32_t n32 = (uint32_t)n;
    QueueRec* seg = queue + (size_t)blockIdx.x * kTabSegCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t toff0 = threadIdx.x * 4u;
    uint32_t round = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride, round++) {
        const bool live = i < n32;
        const size_t rbase = (size_t)blockIdx.x * kBlock + (size_t)round * stride;
        const uint32_t toff = lrm_opaque(toff0), tid_o = lrm_opaque(threadIdx.x);
        LrmVec3 p = p_next;
        {
            const uint32_t i_next = i + stride;
            const size_t rb_next = rbase + stride;
            p_next = LrmVec3{0.f, 0.f, 0.f};
            if (i_next < n32) p_next = kAoS ? LrmVec3{lrm_at(x + 3 * rb_next, 3u * toff), lrm_at(x + 3 * rb_next, 3u * toff + 4u), lrm_at(x + 3 * rb_next, 3u * toff + 8u)}
                                          : LrmVec3{lrm_at(x + rb_next, toff), lrm_at(y + rb_next, toff), lrm_at(z + rb_next, toff)};
        }
        uint32_t doubt = 0;
        const bool m = lrm_xtab_point(lrm_fresh(X), G, p, doubt) && live;
        doubt = live ? ((doubt & 0xffffu) | (selftest & 1u)) : 0u;
        if (live) {
            if (kAoS) {
                lrm_at(dx + 3 * rbase, 3u * toff) = p.x;
                lrm_at(dx + 3 * rbase, 3u * toff + 4u) = p.y;
                lrm_at(dx + 3 * rbase, 3u * toff + 8u) = p.z;
            } else {
#if LRM_TAB_NT_STORE
                __builtin_nontemporal_store(p.x, &lrm_at(dx + rbase, toff));
                __builtin_nontemporal_store(p.y, &lrm_at(dy + rbase, toff));
                __builtin_nontemporal_store(p.z, &lrm_at(dz + rbase, toff));
#else
                lrm_at(dx + rbase, toff) = p.x;
                lrm_at(dy + rbase, toff) = p.y;
                lrm_at(dz + rbase, toff) = p.z;
#endif
            }
            if (mask) lrm_at(mask + rbase, tid_o) = m;
        }
        if (bits) {
            const uint64_t w = __ballot(m);
            if (lane == 0) lrm_at(bits + (rbase >> 6), lrm_opaque((uint32_t)wave * 8u)) = w;
        }
        const uint64_t dm = __ballot(doubt != 0);
        if (dm) { // the point is re-loaded by the pushing lanes (it comes back from the L2): three registers fewer in the chain above
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&s_qn, (uint32_t)__popcll(dm));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            if (doubt) {
                const uint32_t qs = qb + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (qs < (uint32_t)kTabSegCap) {
                    const LrmVec3 q = kAoS ? LrmVec3{lrm_at(x + 3 * rbase, 3u * toff), lrm_at(x + 3 * rbase, 3u * toff + 4u), lrm_at(x + 3 * rbase, 3u * toff + 8u)}
                                           : LrmVec3{lrm_at(x + rbase, toff), lrm_at(y + rbase, toff), lrm_at(z + rbase, toff)};
                    seg[qs] = QueueRec{i, q.x, q.y, q.z};
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_qn;
}

// Bit i of the ballot words an earlier launch wrote becomes `m`.  Only this lane ever changes that bit, so a plain
// read decides whether anything has to change; the (rare) change is an atomic on the word, which other lanes patch
// other bits of.  (Unconditional atomics cost the middle kernel 40 us per 7e5 points.)
__device__ __forceinline__ void patch_bit(uint64_t* bits, size_t i, bool m) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(bits) + (i >> 6);
    const unsigned long long bit = 1ull << (i & 63);
    const bool cur = (__builtin_nontemporal_load(w) & bit) != 0ull;
    if (cur == m) return;
    if (m) atomicOr(w, bit);
    else atomicAnd(w, ~bit);
}

// The fix-up launch is one latency chain per wave (tools/fix_trace.py: 2.4 us launch gap, 2.3 us tables / counts /
// prefix, then ~1000 dependent instructions of the filtered code at one wave per SIMD, 6.6 us -- 14.7 us where a lane
// needs the strict plane evaluation), and its waves are mostly empty (13 queued points on average).  So a point takes TWO
// lanes: lane ^ 1 holds the same point and each lane evaluates ONE of the two yaw candidates of distance_circles
// (one_leg.cu:321-341), half the chain; the results change lanes with one DPP move each.  Same operations in the same
// order as lrm_reach_dist_global_filtered / lrm_dist_global_filtered: bit-identical results.
__device__ __forceinline__ uint32_t lrm_pair_swap(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
}
__device__ __forceinline__ float lrm_pair_swap(float v) { return lrm_u2f(lrm_pair_swap(lrm_f2u(v))); }

template <int kOp>
__device__ __forceinline__ bool lrm_redo_pair(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& p, int cand) {
    const LrmVec3 p_in = p;
    // lrm_dist_global_fast
    LrmVec3 u = lrm_qrot(L.inv_rot, p);
    float buffer = u.x * L.sin_body;
    u.x = u.x * L.cos_body - u.y * L.sin_body;
    u.y = buffer + u.y * L.cos_body;
    // lrm_dist_circles_fast, one candidate per lane
    LrmVec3 mine = u;
    mine.x -= L.body;
    buffer = mine.x * L.sin_pitch;
    mine.x = mine.x * L.cos_pitch - mine.z * L.sin_pitch;
    mine.z = buffer + mine.z * L.cos_pitch;
    const float ax = mine.x;
    const float ang = lrm_atan2f(mine.y, mine.x);
    const float ang_flip = (ang > 0) ? ang - LRM_PI_F : ang + LRM_PI_F;
    uint32_t u_mine = 0;
    const bool r_mine = lrm_finish_closest_fast<true>(LRM_FRESH(L), T, mine, cand ? ang_flip : ang, u_mine); // whole wave: see lrm_plane_dist_coop
    const LrmVec3 other{lrm_pair_swap(mine.x), lrm_pair_swap(mine.y), lrm_pair_swap(mine.z)};
    const bool r_other = lrm_pair_swap(r_mine ? 1u : 0u) != 0u;
    const uint32_t u_other = lrm_pair_swap(u_mine);
    const LrmVec3 a = cand ? other : mine, b = cand ? mine : other;
    const bool res = cand ? r_other : r_mine, resflip = cand ? r_mine : r_other;
    const bool use_direct = (res == resflip) ? (lrm_norm3(a) < lrm_norm3(b)) : res;
    LrmVec3 r = use_direct ? a : b;
    const LrmCompiledLeg& Le = LRM_FRESH(L);
    buffer = r.x * Le.sin_pitch_rev;
    r.x = r.x * Le.cos_pitch_rev - r.z * Le.sin_pitch_rev;
    r.z = buffer + r.z * Le.cos_pitch_rev;
    buffer = r.x * -Le.sin_body;
    r.x = r.x * Le.cos_body - r.y * -Le.sin_body;
    r.y = buffer + r.y * Le.cos_body;
    p = lrm_qrot(Le.fwd_rot, r);
    if (kOp != 2) return res || resflip;
    // lrm_reach_dist_global_filtered: the mask out of the distance's by-products
    const LrmDistByproduct by{res, resflip, cand ? u_mine : u_other, ax, ang, ang_flip};
    bool doubt;
    bool reach = lrm_reach_from_dist(L, by, doubt);
    if (doubt) reach = lrm_reach_global(L, T.lists, p_in);
    return reach;
}

struct FixLds {
    LrmCircle lists[16];
    LrmCompiledLeg::DistCircle dist[16];
    LrmCircle corners[LRM_N_CORNERS];
};

// One workgroup: the segments [blockIdx.x * kSegPerWave, +kSegPerWave) of the `nseg` segments (= workgroups) of a main-kernel
// launch with grid stride `main_stride` points; seg_cap slots per segment.
// (a device function: tol_fixup_kernel is it, and the fix-up launch of LRM_MODE_TOL_REL runs it in its second kind of workgroups;
// bid = the workgroup's number among those that run it, L = the leg in the caller's kernarg segment)
template <int kOp, bool kAoS, int kSegPerWave, int kThreads>
__device__ __forceinline__ void tol_fixup_body(
    const uint32_t bid, const LrmCompiledLeg& L, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const QueueRec* __restrict__ queue,
    const uint32_t* __restrict__ counts, uint32_t nseg, uint32_t seg_cap, size_t main_stride, uint32_t selftest) {
    __shared__ FixLds s_tab;
    __shared__ uint32_t s_pre[kSegPerWave + 1], s_cnt[kSegPerWave];
    constexpr uint32_t kQueueAhead = kThreads / kSegPerWave;
    static_assert(kSegPerWave * kQueueAhead <= kThreads, "one slot per thread");
    __shared__ QueueRec s_q[kSegPerWave][kQueueAhead];
    const uint32_t seg0 = bid * kSegPerWave;
    const int lane = threadIdx.x;
    LRM_TRACE_FIX(0);
    // This kernel is a chain of latencies (launch, counts, tables, gather, ~2000 dependent instructions of the exact
    // code at one wave per SIMD): the table loads (16 bytes per lane and pass) go out first, the counts behind them.
    static_assert(sizeof(s_tab.lists) % 16 == 0 && sizeof(s_tab.dist) % 16 == 0 && sizeof(s_tab.corners) % 16 == 0, "16-byte staging");
    {
        const uint4* src = reinterpret_cast<const uint4*>(&L.lists[0][0]);
        const uint4* dsrc = reinterpret_cast<const uint4*>(&L.dist_tab[0][0]);
        const uint4* csrc = reinterpret_cast<const uint4*>(&L.corner_tab[0]);
        for (int i = lane; i < (int)(sizeof(s_tab.lists) / 16); i += kThreads) reinterpret_cast<uint4*>(s_tab.lists)[i] = src[i];
        for (int i = lane; i < (int)(sizeof(s_tab.dist) / 16); i += kThreads) reinterpret_cast<uint4*>(s_tab.dist)[i] = dsrc[i];
        for (int i = lane; i < (int)(sizeof(s_tab.corners) / 16); i += kThreads) reinterpret_cast<uint4*>(s_tab.corners)[i] = csrc[i];
    }
    if (lane < kSegPerWave) s_cnt[lane] = (seg0 + lane < nseg) ? counts[seg0 + lane] : 0u;
    // the first kQueueAhead slots of every segment (usually all that is queued: 2-3 points per segment) come along with the
    // counts instead of one round trip later; slots beyond a segment's count hold stale indices that are never used
    if (lane < kSegPerWave * kQueueAhead) {
        const uint32_t sj = (uint32_t)lane / kQueueAhead, so = (uint32_t)lane % kQueueAhead;
        s_q[sj][so] = (seg0 + sj < nseg && so < seg_cap) ? queue[(size_t)(seg0 + sj) * seg_cap + so] : QueueRec{0u, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    LRM_TRACE_FIX(1);
    if (lane == 0) {
        uint32_t acc = 0;
        for (int j = 0; j < kSegPerWave; j++) {
            s_pre[j] = acc;
            acc += s_cnt[j] <= seg_cap ? s_cnt[j] : 0u; // an overflowed segment is redone as a whole below
        }
        s_pre[kSegPerWave] = acc;
    }
    uint32_t any = 0;
    for (int j = 0; j < kSegPerWave; j++) any |= s_cnt[j];
    if (any == 0) return; // nothing in doubt in these workgroups (wave-uniform)
    __syncthreads();
    const LrmDistTables T{s_tab.lists, s_tab.dist, s_tab.corners, (selftest & 2u) ? 1u : 0u};
    constexpr int kPerPass = kThreads / 2; // points per pass of the wave
    const int slot = lane >> 1, cand = lane & 1;
    // Every lane of the wave comes here in every pass (`live`: this pair has a point): the strict plane evaluations inside
    // are run by the whole wave (lrm_plane_dist_coop).  Both lanes of a pair hold the same point.
    auto redo = [&](size_t i, LrmVec3 p, bool live) {
        const bool m = lrm_redo_pair<kOp>(L, T, p, cand);
        if (live && cand == 0) {
            if (kAoS) {
                dx[3 * i] = p.x;
                dx[3 * i + 1] = p.y;
                dx[3 * i + 2] = p.z;
            } else {
                dx[i] = p.x;
                dy[i] = p.y;
                dz[i] = p.z;
            }
            if (mask) mask[i] = m;
            if (bits) patch_bit(bits, i, m);
        }
    };
    const uint32_t total = s_pre[kSegPerWave];
    LRM_TRACE_FIX(2);
#if defined(LRM_FIX_TRACE)
    int pass = 0;
#endif
    for (uint32_t k0 = 0; k0 < total; k0 += kPerPass) { // workgroup-uniform trip count
        const uint32_t k = k0 + (uint32_t)slot;
        const bool live = k < total;
        QueueRec rec{0u, 300.f, 0.f, -100.f}; // a lane without a point evaluates a harmless one
        if (live) {
            int j = 0;
#pragma unroll
            for (int t = 1; t < kSegPerWave; t++) j += (s_pre[t] <= k) ? 1 : 0; // segments with nothing queued share a prefix
            const uint32_t off = k - s_pre[j];
            rec = off < kQueueAhead ? s_q[j][off] : queue[(size_t)(seg0 + j) * seg_cap + off];
        }
        redo((size_t)rec.i, LrmVec3{rec.x, rec.y, rec.z}, live);
#if defined(LRM_FIX_TRACE)
        if (lane == 0 && pass < 4 && blockIdx.x < 4096) g_fix_trace[blockIdx.x * 8 + 3 + pass] = wall_clock64();
        pass++;
#endif
    }
#if defined(LRM_FIX_TRACE)
    if (lane == 0 && blockIdx.x < 4096) g_fix_trace[blockIdx.x * 8 + 7] = total;
#endif
    for (int j = 0; j < kSegPerWave; j++) {
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[j]) <= seg_cap) continue; // wave-uniform, and known to be
        // every point of workgroup seg0 + j: i = (seg0 + j) * kBlock + t + round * main_stride
        if (main_stride == 0) continue;
        for (size_t base = (size_t)(seg0 + j) * kBlock; base < n; base += main_stride)
            for (int t0 = 0; t0 < kBlock; t0 += kPerPass) { // workgroup-uniform trip count: kBlock / kPerPass passes for every lane
                const size_t i = base + t0 + slot;
                const bool live = i < n;
                LrmVec3 p{300.f, 0.f, -100.f};
                if (live) p = kAoS ? LrmVec3{x[3 * i], x[3 * i + 1], x[3 * i + 2]} : LrmVec3{x[i], y[i], z[i]};
                redo(i, p, live);
            }
    }
}

template <int kOp, bool kAoS = false, int kSegPerWave = LRM_TOL_SEG_PER_WAVE, int kThreads = kFixBlock>
__global__ __launch_bounds__(kThreads) void tol_fixup_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmCompiledLeg L_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const QueueRec* __restrict__ queue,
    const uint32_t* __restrict__ counts, uint32_t nseg, uint32_t seg_cap, size_t main_stride, uint32_t selftest) {
    tol_fixup_body<kOp, kAoS, kSegPerWave, kThreads>(blockIdx.x, lrm_kernarg<LrmCompiledLeg>(kFixLegArg), x, y, z, n, mask, bits, dx, dy, dz, queue, counts,
                                                     nseg, seg_cap, main_stride, selftest);
}

} // namespace

// LRM_TOL_SELFTEST (tests): bit 0 -- every point is queued (every workgroup's segment overflows: the fix-up re-evaluates the
// whole cloud, so the outputs must be bit-identical to LRM_MODE_FAST); bit 1 -- every plane evaluation of the fix-up takes
// the strict path, i.e. the wave-cooperative lrm_plane_dist_coop (still bit-identical).
static uint32_t tol_selftest_env() {
    const char* e = getenv("LRM_TOL_SELFTEST");
    return e ? (uint32_t)atoi(e) & 3u : 0u;
}

// Workgroups of the main kernel for n points: every resident slot LRM_TOL_GRID_MULT times over, and for larger clouds as many
// as keep a workgroup at about three rounds -- its doubt segment (kSegCap slots) is sized for that.  (With a fixed grid a
// 1e8-point cloud gave each workgroup 24 rounds, most segments overflowed, and the fix-up redid whole workgroups with
// the bit-exact code: 2.34 ms, slower than LRM_MODE_FAST.)
static size_t tol_main_blocks(size_t n) {
    const size_t base = (size_t)256 * LRM_TOL_MIN_WAVES * LRM_TOL_GRID_MULT;
    const size_t need = (n + kBlock - 1) / kBlock;
    size_t blocks = std::max(base, (need + 2) / 3);
    if (blocks > need) blocks = need;
    if (blocks == 0) blocks = 1;
    return blocks;
}
size_t lrm_tol_queue_words(size_t n) { return tol_main_blocks(n) * (4 * kSegCap + 4); } // counts (padded to 16 bytes per workgroup) | 16-byte records

hipError_t lrm_launch_dist_tol(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz,
                               uint32_t* workspace /* lrm_tol_queue_words(n) uint32 */, uint32_t flags, hipStream_t st) {
    const size_t blocks = tol_main_blocks(n);
    const size_t cap = blocks;
    uint32_t* counts = workspace;
    QueueRec* queue = reinterpret_cast<QueueRec*>(workspace + 4 * cap); // 16-byte aligned behind the counts
    if (op == 2) hipLaunchKernelGGL(dist_tol_staged_kernel<2>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counts, flags | tol_selftest_env());
    else hipLaunchKernelGGL(dist_tol_staged_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counts, flags | tol_selftest_env());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned fblocks = (unsigned)((blocks + kSegPerWave - 1) / kSegPerWave);
    const size_t stride = blocks * kBlock;
    if (op == 2) hipLaunchKernelGGL(tol_fixup_kernel<2>, dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts, (uint32_t)blocks, (uint32_t)kSegCap, stride, flags | tol_selftest_env());
    else hipLaunchKernelGGL(tol_fixup_kernel<1>, dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts, (uint32_t)blocks, (uint32_t)kSegCap, stride, flags | tol_selftest_env());
    return hipGetLastError();
}

// Table variant: tab_dev = the device copy of lrm_build_tol_tab's table for TL.  Same workspace layout and fix-up as above, with
// kTabSegCap slots per workgroup.
#ifndef LRM_TAB_ROUNDS
#define LRM_TAB_ROUNDS 6 // rounds of a workgroup on large clouds (it stages 17.5 KB of tables once): its doubt segment (kTabSegCap slots) then holds 17 % of its points
#endif
#ifndef LRM_TAB_FIX_SEGS_REL
#define LRM_TAB_FIX_SEGS_REL 8 // LRM_MODE_TOL_REL queues ~3.6 % of a random cloud: ~200 points per fix-up workgroup,
#endif
#ifndef LRM_TAB_FIX_BLOCK_REL
#define LRM_TAB_FIX_BLOCK_REL 256 // of 256 threads: two passes of 128 points
#endif
// Workgroups of the table kernels for n points.  Every workgroup runs the SAME number of rounds (a grid that is not a divisor of
// the cloud leaves a ragged last round: the 1.25e7-point share of the 1e8 cloud ran 3.4 rounds on 14 336 workgroups and took 11.4 ps
// per point against 9.9 for the whole 1e8, profiles/r03_tol_clouds.txt): rounds = what the base grid (every resident slot
// LRM_TAB_GRID_MULT times over) needs, at most LRM_TAB_ROUNDS (a workgroup's doubt segment holds 17 % of LRM_TAB_ROUNDS rounds),
// at least `min_rounds`; blocks = ceil(need / rounds).
static size_t tab_blocks(size_t n, size_t min_rounds) {
    const size_t base = (size_t)256 * LRM_TAB_MIN_WAVES * LRM_TAB_GRID_MULT; // workgroups of four waves
    const size_t need = (n + kBlock - 1) / kBlock;
    if (need <= 1) return 1;
    size_t rounds = (need + base - 1) / base;
    if (rounds > LRM_TAB_ROUNDS) rounds = LRM_TAB_ROUNDS;
    if (rounds < min_rounds) rounds = min_rounds;
    size_t blocks = (need + rounds - 1) / rounds;
    const size_t floor_blocks = std::min(need, (size_t)2048); // small clouds: never fewer workgroups than fill the chip
    if (blocks < floor_blocks) blocks = floor_blocks;
    return blocks;
}
static size_t tab_main_blocks(size_t n) { return tab_blocks(n, 1); }
// LRM_MODE_TOL_REL: LRM_SHORT_ROUNDS rounds per workgroup, so that its short vectors (4-5 % of a random cloud) about fill the one
// wave that replays them at the end (with three rounds that wave ran half empty: 46 of the step's 407 VALU instructions per point)
#ifndef LRM_SHORT_ROUNDS
#define LRM_SHORT_ROUNDS 5
#endif
static size_t tab_short_blocks(size_t n) { return tab_blocks(n, LRM_SHORT_ROUNDS); }
size_t lrm_tol_tab_queue_words(size_t n) { return tab_main_blocks(n) * (4 * (size_t)kTabSegCap + 4); } // counts (padded to 16 bytes per workgroup) | 16-byte records; (tab_main_blocks >= tab_short_blocks)
size_t lrm_tol_tab_segments(size_t n, bool rel) { return rel ? tab_short_blocks(n) : tab_main_blocks(n); }
template <int kOp, bool kAoS>
static hipError_t launch_tab(const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L, const LrmTolLeg& TL, const LrmXtabLeg& X,
                             const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz, uint32_t* workspace,
                             uint32_t flags, hipStream_t st) {
    const size_t blocks = (flags & LRM_TOLF_SHORT) ? tab_short_blocks(n) : tab_main_blocks(n);
    uint32_t* counts = workspace;
    QueueRec* queue = reinterpret_cast<QueueRec*>(workspace + 4 * blocks); // 16-byte aligned behind the counts
    if (flags & LRM_TOLF_SHORT)
        hipLaunchKernelGGL((dist_tab_kernel<kOp, kAoS, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, X, mask, bits, dx, dy, dz, tab_dev, queue, counts,
                           flags | tol_selftest_env());
    else
        hipLaunchKernelGGL((dist_tab_kernel<kOp, kAoS, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, X, mask, bits, dx, dy, dz, tab_dev, queue, counts,
                           flags | tol_selftest_env());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t stride = blocks * kBlock;
    const unsigned fblocks = (unsigned)((blocks + LRM_TAB_FIX_SEGS - 1) / LRM_TAB_FIX_SEGS);
    hipLaunchKernelGGL((tol_fixup_kernel<kOp, kAoS, LRM_TAB_FIX_SEGS>), dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts,
                       (uint32_t)blocks, (uint32_t)kTabSegCap, stride, flags | tol_selftest_env());
    return hipGetLastError();
}
hipError_t lrm_launch_dist_tab(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, const LrmXtabLeg& X, const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                               float* dz, uint32_t* workspace /* lrm_tol_tab_queue_words(n) uint32 */, uint32_t flags, hipStream_t st) {
    return op == 2 ? launch_tab<2, false>(x, y, z, n, L, TL, X, tab_dev, mask, bits, dx, dy, dz, workspace, flags, st)
                   : launch_tab<1, false>(x, y, z, n, L, TL, X, tab_dev, mask, bits, dx, dy, dz, workspace, flags, st);
}
hipError_t lrm_launch_dist_tab_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmTolLeg& TL, const LrmXtabLeg& X, const uint8_t* tab_dev,
                                   uint8_t* mask, float* dxyz, uint32_t* workspace, uint32_t flags, hipStream_t st) {
    return op == 2 ? launch_tab<2, true>(xyz, nullptr, nullptr, n, L, TL, X, tab_dev, mask, nullptr, dxyz, nullptr, nullptr, workspace, flags, st)
                   : launch_tab<1, true>(xyz, nullptr, nullptr, n, L, TL, X, tab_dev, mask, nullptr, dxyz, nullptr, nullptr, workspace, flags, st);
}

// LRM_MODE_FAST through the table: dist_xtab_kernel + tol_fixup_kernel for its doubtful points.  Workspace as lrm_launch_dist_tab.
template <int kOp, bool kAoS>
static hipError_t launch_xtab(const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L, const LrmXtabLeg& X,
                              const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz, uint32_t* workspace, hipStream_t st) {
    const size_t blocks = tab_main_blocks(n);
    uint32_t* counts = workspace;
    QueueRec* queue = reinterpret_cast<QueueRec*>(workspace + 4 * blocks);
    hipLaunchKernelGGL((dist_xtab_kernel<kOp, kAoS>), dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, X, mask, bits, dx, dy, dz, tab_dev, queue, counts, tol_selftest_env());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t stride = blocks * kBlock;
    const unsigned fblocks = (unsigned)((blocks + LRM_TAB_FIX_SEGS - 1) / LRM_TAB_FIX_SEGS);
    hipLaunchKernelGGL((tol_fixup_kernel<kOp, kAoS, LRM_TAB_FIX_SEGS>), dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts,
                       (uint32_t)blocks, (uint32_t)kTabSegCap, stride, tol_selftest_env());
    return hipGetLastError();
}
hipError_t lrm_launch_dist_xtab(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L, const LrmXtabLeg& X,
                                const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz, uint32_t* workspace, hipStream_t st) {
    return op == 2 ? launch_xtab<2, false>(x, y, z, n, L, X, tab_dev, mask, bits, dx, dy, dz, workspace, st)
                   : launch_xtab<1, false>(x, y, z, n, L, X, tab_dev, mask, bits, dx, dy, dz, workspace, st);
}
hipError_t lrm_launch_dist_xtab_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmXtabLeg& X, const uint8_t* tab_dev,
                                    uint8_t* mask, float* dxyz, uint32_t* workspace, hipStream_t st) {
    return op == 2 ? launch_xtab<2, true>(xyz, nullptr, nullptr, n, L, X, tab_dev, mask, nullptr, dxyz, nullptr, nullptr, workspace, st)
                   : launch_xtab<1, true>(xyz, nullptr, nullptr, n, L, X, tab_dev, mask, nullptr, dxyz, nullptr, nullptr, workspace, st);
}

// The same two launches on the float3 arrays of the apply_kernel boundary (cross_compiled.cu:33-79): no bit words.
hipError_t lrm_launch_dist_tol_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmTolLeg& TL, uint8_t* mask,
                                   float* dxyz, uint32_t* workspace /* lrm_tol_queue_words(n) uint32 */, uint32_t flags, hipStream_t st) {
    const size_t blocks = tol_main_blocks(n);
    uint32_t* counts = workspace;
    QueueRec* queue = reinterpret_cast<QueueRec*>(workspace + 4 * blocks); // 16-byte aligned behind the counts
    if (op == 2) hipLaunchKernelGGL((dist_tol_staged_kernel<2, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, xyz, nullptr, nullptr, n, TL, mask, nullptr, dxyz, nullptr, nullptr, queue, counts, flags | tol_selftest_env());
    else hipLaunchKernelGGL((dist_tol_staged_kernel<1, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, xyz, nullptr, nullptr, n, TL, mask, nullptr, dxyz, nullptr, nullptr, queue, counts, flags | tol_selftest_env());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned fblocks = (unsigned)((blocks + kSegPerWave - 1) / kSegPerWave);
    const size_t stride = blocks * kBlock;
    if (op == 2) hipLaunchKernelGGL((tol_fixup_kernel<2, true>), dim3(fblocks), dim3(kFixBlock), 0, st, xyz, nullptr, nullptr, n, L, mask, nullptr, dxyz, nullptr, nullptr, queue, counts, (uint32_t)blocks, (uint32_t)kSegCap, stride, flags | tol_selftest_env());
    else hipLaunchKernelGGL((tol_fixup_kernel<1, true>), dim3(fblocks), dim3(kFixBlock), 0, st, xyz, nullptr, nullptr, n, L, mask, nullptr, dxyz, nullptr, nullptr, queue, counts, (uint32_t)blocks, (uint32_t)kSegCap, stride, flags | tol_selftest_env());
    return hipGetLastError();
}

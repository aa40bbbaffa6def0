// lrm_octree.hip -- octree-culled positionability (BASELINE config 5), the MI355X re-design of
// apply_oct / branchKernel / validity_child (several_leg_octree.cu:19-488, octree_util.cu.h).
//
// The reference grows the tree from INSIDE kernels: device-side cudaMalloc of the children
// (several_leg_octree.cu:276), kernels launching kernels (:306, :372), one host iteration per level
// (:447-452).  HIP has neither device malloc nor dynamic parallelism, and neither is wanted: here
// the tree is LEVEL-SYNCHRONOUS with flat arrays.  Per level the host creates the children of every
// node that must be refined (CreateChildBox, octree_util.cu.h:105-151), ONE kernel evaluates all
// (child, foothold, orientation) work items of the level, and three flag bits per child come back.
//
// Work item semantics (validity_child, several_leg_octree.cu:19-151), body = child box centre:
//   skip   if the foothold is outside the parent-sized box grown by the leg's total length (:76-82)
//   for each leg mounted at LegMount[l] (settings.h:42): distance_global(foothold - body) -> reachable?,
//          and does the distance vector fall inside the child box (sphere for small boxes)? (:91-114)
//   edge   = more than LegCount - LegNumberForStab legs have their boundary inside the box
//   reach  = parent already valid, or at least LegNumberForStab legs reach the foothold
//   flags  |= {reach, reach and not edge ("valid leaf"), edge}
// The reference ORs these flags through racy __shared__ bools per 256-thread block and derives
// node.onEdge per block (:134-150), which makes its result depend on the launch geometry; here the
// three ORs are global per child and  onEdge = edge_any and not leaf_any  -- the evident intent.
// (The call site of apply_oct is dead code in the reference, several_leg.cpp:224; MAX_DEPTH is 1 as
// committed.  No reference run exists for this path: parity is pinned by composition of the pinned
// distance_global only, against tests/octree_oracle.py.)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lrm.h"
#include "lrm_compile.h"
#include "lrm_launch.h"
#include "lrm_point.h"
#include "lrm_point_fast.h"
#include "lrm_point_tol.h"

namespace {

constexpr int kOctBlock = 256;
#ifndef LRM_OCT_MIN_WAVES
#define LRM_OCT_MIN_WAVES 4 // 128 VGPRs (8 B/lane of scratch in the table form) instead of 153 at 3 waves: config-5 share 114 -> 100 ms; 5 waves (112 B of scratch): 116
#endif

struct OctChild {          // one child box of the level being evaluated
    float c[3], h[3];      // centre, half size ("topOffset")
    float ph[3];           // parent half size
    float margin;          // 0 when rotations are active, EnableRotBelow / 3 otherwise
    int32_t n_angles;      // 27 when parent.half.x < EnableRotBelow, else 1
    int32_t parent_valid;
    int32_t skip;          // dead quadrant or already valid
    int32_t pad;
};

// isInBox, octree_util.cu.h:153-159 (note the asymmetric comparisons)
__device__ __forceinline__ bool in_box(LrmVec3 v, float hx, float hy, float hz) {
    hx = fabsf(hx); hy = fabsf(hy); hz = fabsf(hz);
    return hx >= v.x && hy >= v.y && hz >= v.z && -hx < v.x && -hy < v.y && -hz < v.z;
}

// One (child, foothold) work item: the flags it contributes (1 reach, 2 valid leaf, 4 edge), validity_child
// several_leg_octree.cu:84-131.
// Exact cull in front of the distance evaluations: a leg can only reach the foothold, and its distance vector can only
// fall inside the child box, when the foothold is within hd = |half size| (+ margin) of the bounding sphere of everything
// that leg reaches (`spheres`, from LrmCompiledLeg::pair_center / pair_r2 turned into this frame: the nearest boundary
// point lies in the closure of the reachable set, so |distance vector| >= distance to the sphere).  With `near` such legs,
// reach_count <= near and cross_count <= near: when near < LegNumberForStab and near <= LegCount - LegNumberForStab the
// item contributes `reach = parent_valid`, no edge -- without evaluating anything; and inside an item that is evaluated, the
// legs beyond their own sphere are skipped (they neither reach nor cross).  Deep in the tree (small boxes, four
// legs mounted 45 degrees apart) that is most of the footholds inside the elongated parent box.
// `tols` (filtered mode, or null): the tolerance block of every (orientation, leg) (lrm_compile_tol).  The flags need the
// distance's validity and whether its vector falls inside the child box -- DECISIONS, not the vector's last bits: the
// contract-tolerance evaluation (lrm_dist_tol, lrm_point_tol.h: ~500 instructions instead of the filtered code's ~1000) answers
// them unless one of its own decisions is in doubt or the vector ends within its error of a face of the box (of the sphere);
// those few evaluations are redone by the filtered code.  Same flags either way (tests/test_gpu_octree.py).
// kTol: 0 the filtered code alone, 1 the tolerance evaluation without a table first, 2 through the plane tables (one instantiation
// each: with both tolerance forms inlined the kernel lost a fifth of its speed to registers)
template <bool kFast, int kTol>
__device__ __forceinline__ uint32_t oct_item_flags(const OctChild& ch, LrmVec3 vect, float h2, float hd, const LrmCompiledLeg* __restrict__ legs,
                                                   const LrmTolLeg* __restrict__ tols, const uint8_t* const* __restrict__ tabs,
                                                   const float4* __restrict__ spheres, int leg_count, int legs_for_stab, float convex_r2,
                                                   int a_begin = 0, int a_end = -1 /* orientations [a_begin, a_end); -1: all of the child's */) {
    uint32_t mine = 0;
    if (a_end < 0) a_end = ch.n_angles;
    for (int a = a_begin; a < a_end; a++) {
        int near = 0;
        uint32_t near_bits = 0;
        for (int l = 0; l < leg_count; l++) {
            const float4 sp = spheres[a * leg_count + l];
            const float ex = vect.x - sp.x, ey = vect.y - sp.y, ez = vect.z - sp.z, rr = sp.w + hd;
            const bool in = ex * ex + ey * ey + ez * ez < rr * rr;
            near += in ? 1 : 0;
            near_bits |= in ? (1u << l) : 0u;
        }
        if (near < legs_for_stab && near <= leg_count - legs_for_stab) {
            mine |= ch.parent_valid ? 3u : 0u;
            continue;
        }
        int reach_count = 0, cross_count = 0;
        for (int l = 0; l < leg_count; l++) {
            if (!((near_bits >> l) & 1u)) continue; // beyond this leg's sphere: it neither reaches nor crosses (same argument)
            const LrmCompiledLeg& L = legs[a * leg_count + l];
            LrmVec3 v = vect;
            bool sub = false, done = false;
            if (kFast && kTol != 0) {
                const LrmTolLeg& TL = tols[a * leg_count + l];
                if (TL.tol_ok) { // (wave-uniform)
                    uint32_t dbt = 0;
                    LrmVec3 tv = vect;
                    // (orientation, leg) is the same for the whole wave: the table's header, rows and bounds are read through a uniform base
                    // straight from global memory (L2-resident: a level walks 4 x 27 tables at most)
                    const uint8_t* tab = kTol == 2 ? tabs[a * leg_count + l] : nullptr;
                    bool tsub = false;
                    if (kTol == 2) {
                        if (!tab) dbt = 1u; // (a leg without a table: the filtered code below)
                        else {
                        const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab);
                        const LrmTolTabView G = lrm_toltab_view(tab, hd->rows, hd->vrows,
                                                                reinterpret_cast<const uint32_t*>(tab + sizeof(LrmTolTabHeader) + 2 * (size_t)hd->bound_off[0]), TL.r_outer);
                        tsub = lrm_tab_point(TL, G, tv, dbt);
                        }
                    } else {
                        tsub = lrm_dist_tol(TL, LrmTolTables{&TL.circ[0][0], &TL.feat[0]}, tv, dbt);
                    }
                    // the tolerance vector is within 1e-5 of max(|d|, (|p| + body) / 8) of the reference's (include/lrm.h): eps covers it twice
                    const float eps = 2.0e-5f * (fabsf(tv.x) + fabsf(tv.y) + fabsf(tv.z) + fabsf(vect.x) + fabsf(vect.y) + fabsf(vect.z) + 400.f) + 1.0e-3f;
                    bool near_face;
                    if (h2 > convex_r2) {
                        near_face = fabsf(fabsf(tv.x) - fabsf(ch.h[0])) < eps || fabsf(fabsf(tv.y) - fabsf(ch.h[1])) < eps || fabsf(fabsf(tv.z) - fabsf(ch.h[2])) < eps;
                    } else {
                        const float n = sqrtf(tv.x * tv.x + tv.y * tv.y + tv.z * tv.z), rr = sqrtf(fmaxf(h2 + ch.margin, 0.f));
                        near_face = fabsf(n - rr) < eps;
                    }
                    if ((dbt & 0xffffu) == 0u && !near_face) {
                        v = tv;
                        sub = tsub;
                        done = true;
                    }
                }
            }
            if (!done) {
                if (kFast) sub = lrm_dist_global_filtered(L, LrmDistTables{&L.lists[0][0], &L.dist_tab[0][0], &L.corner_tab[0]}, v);
                else sub = lrm_dist_global(L, &L.lists[0][0], v);
            }
            bool cross;
            if (h2 > convex_r2) cross = in_box(v, ch.h[0], ch.h[1], ch.h[2]);   // :103-107 (margin unused there)
            else cross = (v.x * v.x + v.y * v.y + v.z * v.z) < h2 + ch.margin;  // :108-109
            cross_count += cross;
            reach_count += sub;
        }
        const bool edge = cross_count > leg_count - legs_for_stab;
        const bool reach = ch.parent_valid || (reach_count >= legs_for_stab);
        mine |= (reach ? 1u : 0u) | ((reach && !edge) ? 2u : 0u) | (edge ? 4u : 0u);
    }
    return mine;
}

// ---- the table form with its doubts DEFERRED (round 4) -------------------------------------------------------------------
// Inlined next to the table evaluation, the filtered code of the doubtful evaluations (a few per cent) sets the kernel's register
// count (153 VGPRs, 3-4 waves per SIMD).  Here an (item, orientation) with a doubtful leg contributes nothing in the main
// kernel: a record {child, foothold, orientation} goes to the wave's LDS segment (count in a scalar register, no atomic), full
// segments to a global queue (one atomic per flush), and oct_deferred_kernel evaluates the queued (item, orientation) pairs
// with the filtered code for every leg and ORs their flags in.  The flags are ORs: the order changes nothing; a queue that
// overflows raises a flag and the host runs the level again with the inline form.
struct OctDeferQueue {
    uint32_t* rec;   // three words per record
    uint32_t* count; // [0] records, [1] overflowed
    uint32_t cap;
};
// records of the global queue: 2-3 % of a level's evaluated (item, orientation) pairs are deferred -- 6.9e6 at the deepest level of the
// config-5 share (1.25e7 footholds): one record per foothold, within [2^20, 2^25] (12 bytes each)
static size_t oct_defer_cap(size_t nf) {
    if (const char* e = getenv("LRM_OCT_DEFER_CAP")) return std::max<size_t>((size_t)atol(e), 64); // tests: a queue that overflows
    return std::min<size_t>(std::max<size_t>(nf, (size_t)1 << 20), (size_t)1 << 25);
}
constexpr int kOctDeferSeg = 64; // records per wave segment: one ballot's worth always fits an emptied segment

__device__ __forceinline__ void oct_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// all 64 lanes together
__device__ __forceinline__ void oct_defer_flush(uint32_t* seg, uint32_t& wq, const OctDeferQueue Q) {
    if (wq == 0u) return; // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(Q.count, wq);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base + wq <= Q.cap) {
        if (lane < wq) {
            Q.rec[3u * (base + lane)] = seg[3u * lane];
            Q.rec[3u * (base + lane) + 1u] = seg[3u * lane + 1u];
            Q.rec[3u * (base + lane) + 2u] = seg[3u * lane + 2u];
        }
    } else if (lane == 0u) {
        atomicOr(Q.count + 1, 1u);
    }
    oct_wave_fence();
    wq = 0u;
}
// oct_item_flags<true, 2> for a whole wave (every lane calls it; `act`: this lane has an item), doubts deferred
__device__ __forceinline__ uint32_t oct_item_flags_defer(const OctChild& ch, bool act, LrmVec3 vect, float h2, float hd,
                                                         const LrmTolLeg* __restrict__ tols, const uint8_t* const* __restrict__ tabs,
                                                         const float4* __restrict__ spheres, int leg_count, int legs_for_stab, float convex_r2,
                                                         uint32_t* seg, uint32_t& wq, uint32_t child, uint32_t f, const OctDeferQueue Q) {
    uint32_t mine = 0;
    for (int a = 0; a < ch.n_angles; a++) {
        int near = 0;
        uint32_t near_bits = 0;
        for (int l = 0; l < leg_count; l++) {
            const float4 sp = spheres[a * leg_count + l];
            const float ex = vect.x - sp.x, ey = vect.y - sp.y, ez = vect.z - sp.z, rr = sp.w + hd;
            const bool in = ex * ex + ey * ey + ez * ez < rr * rr;
            near += in ? 1 : 0;
            near_bits |= in ? (1u << l) : 0u;
        }
        const bool culled = near < legs_for_stab && near <= leg_count - legs_for_stab;
        if (act && culled) mine |= ch.parent_valid ? 3u : 0u;
        const bool eval = act && !culled;
        bool dfr = false;
#if defined(LRM_OCT_COUNT)
        uint32_t why = 0;
#endif
        int reach_count = 0, cross_count = 0;
        if (__ballot(eval) != 0ull) {
            for (int l = 0; l < leg_count; l++) {
                const bool on = eval && !dfr && ((near_bits >> l) & 1u) != 0u;
                if (__ballot(on) == 0ull) continue; // wave-uniform
                const LrmTolLeg& TL = tols[a * leg_count + l];
                const uint8_t* tab = tabs[a * leg_count + l];
                if (!TL.tol_ok || !tab) { // (wave-uniform) a leg without a table: its items go to the filtered code
                    dfr = dfr || on;
#if defined(LRM_OCT_COUNT)
                    if (on) why = 1u;
#endif
                    continue;
                }
                if (on) {
                    uint32_t dbt = 0;
                    LrmVec3 tv = vect;
                    const LrmTolTabHeader* hd_ = reinterpret_cast<const LrmTolTabHeader*>(tab);
                    const LrmTolTabView G = lrm_toltab_view(tab, hd_->rows, hd_->vrows,
                                                            reinterpret_cast<const uint32_t*>(tab + sizeof(LrmTolTabHeader) + 2 * (size_t)hd_->bound_off[0]), TL.r_outer);
                    const bool tsub = lrm_tab_point(TL, G, tv, dbt);
                    const float eps = 2.0e-5f * (fabsf(tv.x) + fabsf(tv.y) + fabsf(tv.z) + fabsf(vect.x) + fabsf(vect.y) + fabsf(vect.z) + 400.f) + 1.0e-3f; // as oct_item_flags
                    bool near_face, cross;
                    if (h2 > convex_r2) {
                        near_face = fabsf(fabsf(tv.x) - fabsf(ch.h[0])) < eps || fabsf(fabsf(tv.y) - fabsf(ch.h[1])) < eps || fabsf(fabsf(tv.z) - fabsf(ch.h[2])) < eps;
                        cross = in_box(tv, ch.h[0], ch.h[1], ch.h[2]);
                    } else {
                        const float n2 = tv.x * tv.x + tv.y * tv.y + tv.z * tv.z;
                        near_face = fabsf(sqrtf(n2) - sqrtf(fmaxf(h2 + ch.margin, 0.f))) < eps;
                        cross = n2 < h2 + ch.margin;
                    }
#if defined(LRM_OCT_COUNT)
                    if ((dbt & 0xffffu) != 0u) why = 2u; else if (near_face) why = 3u;
#endif
                    if ((dbt & 0xffffu) != 0u || near_face) dfr = true;
                    else {
                        cross_count += cross;
                        reach_count += tsub;
                    }
                }
            }
        }
        if (eval && !dfr) {
            const bool edge = cross_count > leg_count - legs_for_stab;
            const bool reach = ch.parent_valid || (reach_count >= legs_for_stab);
            mine |= (reach ? 1u : 0u) | ((reach && !edge) ? 2u : 0u) | (edge ? 4u : 0u);
        }
#if defined(LRM_OCT_COUNT) // counting build: evaluated (item, orientation) pairs in count[2], deferred for a missing table / a doubt bit / a near face in [3..5]
        {
            const unsigned long long em = __ballot(eval);
            if ((threadIdx.x & 63u) == 0u && em) atomicAdd(Q.count + 2, (uint32_t)__builtin_popcountll(em));
            const unsigned long long c3 = __ballot(dfr && why == 1u), c4 = __ballot(dfr && why == 2u), c5 = __ballot(dfr && why == 3u);
            if ((threadIdx.x & 63u) == 0u && c3) atomicAdd(Q.count + 3, (uint32_t)__builtin_popcountll(c3));
            if ((threadIdx.x & 63u) == 0u && c4) atomicAdd(Q.count + 4, (uint32_t)__builtin_popcountll(c4));
            if ((threadIdx.x & 63u) == 0u && c5) atomicAdd(Q.count + 5, (uint32_t)__builtin_popcountll(c5));
        }
#endif
        const unsigned long long dm = __ballot(dfr);
        if (dm != 0ull) { // wave-uniform
            const uint32_t n = (uint32_t)__builtin_popcountll(dm);
            if (wq + n > (uint32_t)kOctDeferSeg) oct_defer_flush(seg, wq, Q);
            const uint32_t pos = wq + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
            if (dfr) {
                seg[3u * pos] = child;
                seg[3u * pos + 1u] = f;
                seg[3u * pos + 2u] = (uint32_t)a;
            }
            wq += n;
            oct_wave_fence();
        }
    }
    return mine;
}

template <bool kFast, int kTol>
__global__ __launch_bounds__(kOctBlock, LRM_OCT_MIN_WAVES) void oct_validity_kernel(
    const OctChild* __restrict__ children, int n_children, const float* __restrict__ fx,
    const float* __restrict__ fy, const float* __restrict__ fz, size_t nf,
    const LrmCompiledLeg* __restrict__ legs /* [n_angles_max][leg_count] */, const LrmTolLeg* __restrict__ tols /* same shape, or null */,
    const uint8_t* const* __restrict__ tabs /* the plane tables (lrm_toltab_build.h) of the same (orientation, leg) pairs, or null */,
    const float4* __restrict__ spheres /* same shape */, int leg_count, int legs_for_stab,
    float reach_len, float convex_r2, uint32_t* __restrict__ flags /* per child: 1 reach, 2 valid leaf, 4 edge */) {
    const OctChild ch = children[blockIdx.y];
    if (ch.skip) return;
    const float h2 = ch.h[0] * ch.h[0] + ch.h[1] * ch.h[1] + ch.h[2] * ch.h[2]; // linormRaw(topOffset)
    const float hd = sqrtf(h2 + fmaxf(ch.margin, 0.f)) * 1.0001f + 0.01f; // the longest distance vector that can "cross" this child
    uint32_t mine = 0, published = 0;
    const size_t stride = (size_t)gridDim.x * kOctBlock;
    const size_t nf_pad = (nf + 63) & ~(size_t)63;
    for (size_t f = (size_t)blockIdx.x * kOctBlock + threadIdx.x; f < nf_pad; f += stride) {
        // The three flags are ORs: once all are set for this child, nothing is left to learn.  The huge boxes of
        // the first levels see every foothold and saturate after a few hundred of them -- without this exit a level
        // costs 45 ms per 1e6 footholds.  (Wave-uniform: every lane reads the same word.)
        if (*reinterpret_cast<volatile uint32_t*>(&flags[blockIdx.y]) == 7u) break;
        uint32_t wave_bits = mine;
        for (int off = 32; off > 0; off >>= 1) wave_bits |= __shfl_xor(wave_bits, off);
        if ((threadIdx.x & 63) == 0 && (wave_bits & ~published)) atomicOr(&flags[blockIdx.y], wave_bits);
        published = wave_bits;
        if (f >= nf) continue;
        const LrmVec3 vect{fx[f] - ch.c[0], fy[f] - ch.c[1], fz[f] - ch.c[2]};
        // elongated parent box, several_leg_octree.cu:76-82
        if (!in_box(vect, ch.ph[0] + reach_len, ch.ph[1] + reach_len, ch.ph[2] + reach_len)) continue;
        mine |= oct_item_flags<kFast, kTol>(ch, vect, h2, hd, legs, tols, tabs, spheres, leg_count, legs_for_stab, convex_r2);
    }
    // wave OR, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) mine |= __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicOr(&flags[blockIdx.y], mine);
}

// ---- scale: footholds binned by bounding boxes --------------------------------------------------------
// oct_validity_kernel above walks EVERY foothold for every child: fine for the few, huge boxes of the first levels
// (its grid spreads the footholds of one child over the chip), hopeless deeper down, where a level has thousands
// of small children that each see a handful of footholds.  From kOctChunkedFrom children on (every level but the first), a level runs
// oct_validity_chunked_kernel instead: the footholds are in Morton order (sorted once per call), two levels of
// axis-aligned boxes cover them in memory order (one per 1024-foothold tile, one per 64-foothold chunk, built once
// per call by oct_boxes_kernel), and a workgroup = one child tests tile boxes, then the chunk boxes of the
// surviving tiles, against the child's elongated box (several_leg_octree.cu:76-82) and evaluates only the
// footholds of surviving chunks.  A child whose three flags are all set stops at once (they are ORs).
constexpr int kOctChunkedFrom = 9; // 1.25e7 footholds, depth 6: thresholds 1 / 9 / 65 / 257 -> 198 / 188 / 193 / 226 ms of kernels
constexpr int kOctMaxChunks = 4096; // surviving chunks per round of 256 tiles

__global__ __launch_bounds__(kOctBlock) void oct_boxes_kernel(const float* __restrict__ fx, const float* __restrict__ fy,
                                                             const float* __restrict__ fz, size_t nf, size_t ntiles,
                                                             float* __restrict__ boxes) {
    __shared__ float s_red[6][16];
    const size_t t0 = (size_t)blockIdx.x * 1024;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int sub = wave; sub < 16; sub += kOctBlock / 64) {
        const size_t i = t0 + (size_t)sub * 64 + lane;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        if (i < nf) {
            lo[0] = hi[0] = fx[i];
            lo[1] = hi[1] = fy[i];
            lo[2] = hi[2] = fz[i];
        }
        for (int a = 0; a < 3; a++)
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
            }
        if (lane == 0) {
            const size_t c = (size_t)blockIdx.x * 16 + sub;
            for (int a = 0; a < 3; a++) {
                s_red[a][sub] = lo[a];
                s_red[3 + a][sub] = hi[a];
                boxes[(ntiles + c) * 6 + a] = lo[a]; // an empty chunk keeps (+big, -big): it meets nothing
                boxes[(ntiles + c) * 6 + 3 + a] = hi[a];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[threadIdx.x][0];
        for (int w = 1; w < 16; w++) v = (threadIdx.x < 3) ? fminf(v, s_red[threadIdx.x][w]) : fmaxf(v, s_red[threadIdx.x][w]);
        boxes[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// does the box [lo, hi] hold a point p with -H < p - c <= H on every axis (in_box's asymmetric test)?  Conservative.
__device__ __forceinline__ bool box_meets(const float* bb, const float* c, const float* H) {
    return bb[0] <= c[0] + H[0] && bb[3] >= c[0] - H[0] && bb[1] <= c[1] + H[1] && bb[4] >= c[1] - H[1] &&
           bb[2] <= c[2] + H[2] && bb[5] >= c[2] - H[2];
}

#ifndef LRM_OCT_DEFER_MIN_WAVES
#define LRM_OCT_DEFER_MIN_WAVES 5
#endif
template <bool kFast, int kTol, bool kDefer = false>
__global__ __launch_bounds__(kOctBlock, kDefer ? LRM_OCT_DEFER_MIN_WAVES : LRM_OCT_MIN_WAVES) void oct_validity_chunked_kernel(
    const OctChild* __restrict__ children, int n_children, const float* __restrict__ fx,
    const float* __restrict__ fy, const float* __restrict__ fz, size_t nf, const float* __restrict__ boxes, size_t ntiles,
    const LrmCompiledLeg* __restrict__ legs, const LrmTolLeg* __restrict__ tols, const uint8_t* const* __restrict__ tabs, const float4* __restrict__ spheres, int leg_count, int legs_for_stab, float reach_len,
    float convex_r2, uint32_t* __restrict__ flags /* zeroed by the host */, uint32_t splits, uint32_t tpr /* tiles per round, <= kOctBlock */, const OctDeferQueue Q /* kDefer */) {
    static_assert(!kDefer || (kFast && kTol == 2), "the deferred form is the table form");
    __shared__ uint32_t s_flags, s_ntiles, s_nchunks;
    __shared__ uint32_t s_defer[kDefer ? (kOctBlock / 64) * kOctDeferSeg * 3 : 1];
    uint32_t* dseg = s_defer + (kDefer ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (uint32_t)(kOctDeferSeg * 3) : 0u);
    uint32_t dwq = 0; // kDefer: records waiting in this wave's segment (wave-uniform)
    __shared__ uint32_t s_tiles[kOctBlock];
    __shared__ uint32_t s_chunks[kOctMaxChunks];
    // A level with few, large children would leave most of the chip idle at one workgroup per child: `splits` workgroups
    // share a child, each takes every splits-th round of 256 tiles, and the flags meet in the child's global word
    // (atomicOr; read back every round for the early exit).
    for (size_t w = blockIdx.x; w < (size_t)n_children * splits; w += gridDim.x) {
        const size_t child = w / splits;
        const uint32_t sp = (uint32_t)(w % splits);
        const OctChild ch = children[child];
        if (ch.skip) continue; // block-uniform
        const float H[3] = {fabsf(ch.ph[0] + reach_len), fabsf(ch.ph[1] + reach_len), fabsf(ch.ph[2] + reach_len)};
        const float h2 = ch.h[0] * ch.h[0] + ch.h[1] * ch.h[1] + ch.h[2] * ch.h[2];
        const float hd = sqrtf(h2 + fmaxf(ch.margin, 0.f)) * 1.0001f + 0.01f;
        if (threadIdx.x == 0) s_flags = 0;
        __syncthreads();
        for (size_t tile0 = (size_t)sp * tpr; tile0 < ntiles; tile0 += (size_t)splits * tpr) {
            if (threadIdx.x == 0) {
                s_ntiles = 0;
                s_nchunks = 0;
                if (splits > 1) { // what the other workgroups of this child have found, and ours to them
                    const uint32_t mine_so_far = s_flags;
                    const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(&flags[child]);
                    if (mine_so_far & ~seen) atomicOr(&flags[child], mine_so_far);
                    s_flags = mine_so_far | seen;
                }
            }
            __syncthreads();
            if (s_flags == 7u) break; // nothing left to learn (block-uniform: read after the barrier)
            const size_t t = tile0 + threadIdx.x;
            if (threadIdx.x < tpr && t < ntiles && box_meets(boxes + t * 6, ch.c, H)) s_tiles[atomicAdd(&s_ntiles, 1u)] = (uint32_t)t;
            __syncthreads();
            const uint32_t nt = s_ntiles;
            for (uint32_t k = threadIdx.x; k < nt * 16u; k += kOctBlock) {
                const size_t c = (size_t)s_tiles[k >> 4] * 16 + (k & 15u);
                if (box_meets(boxes + (ntiles + c) * 6, ch.c, H)) s_chunks[atomicAdd(&s_nchunks, 1u)] = (uint32_t)c;
            }
            __syncthreads();
            const uint32_t nchunks = s_nchunks;
            for (uint32_t k = threadIdx.x >> 6; k < nchunks; k += kOctBlock / 64) { // a wave per surviving chunk
                if (*reinterpret_cast<volatile uint32_t*>(&s_flags) == 7u) break;
                const size_t f = (size_t)s_chunks[k] * 64 + (threadIdx.x & 63);
                uint32_t mine = 0;
                if (kDefer) { // every lane of the wave together (ballots inside)
                    const bool live = f < nf;
                    const size_t fl = live ? f : 0;
                    const LrmVec3 vect{fx[fl] - ch.c[0], fy[fl] - ch.c[1], fz[fl] - ch.c[2]};
                    const bool act = live && in_box(vect, ch.ph[0] + reach_len, ch.ph[1] + reach_len, ch.ph[2] + reach_len);
                    mine = oct_item_flags_defer(ch, act, vect, h2, hd, tols, tabs, spheres, leg_count, legs_for_stab, convex_r2, dseg, dwq, (uint32_t)child, (uint32_t)f, Q);
                } else if (f < nf) {
                    const LrmVec3 vect{fx[f] - ch.c[0], fy[f] - ch.c[1], fz[f] - ch.c[2]};
                    if (in_box(vect, ch.ph[0] + reach_len, ch.ph[1] + reach_len, ch.ph[2] + reach_len)) {
                        mine = oct_item_flags<kFast, kTol>(ch, vect, h2, hd, legs, tols, tabs, spheres, leg_count, legs_for_stab, convex_r2);
                    }
                }
                for (int off = 32; off > 0; off >>= 1) mine |= __shfl_xor(mine, off);
                if ((threadIdx.x & 63) == 0 && mine) atomicOr(&s_flags, mine);
            }
            __syncthreads();
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_flags) atomicOr(&flags[child], s_flags);
        __syncthreads();
    }
    if (kDefer) oct_defer_flush(dseg, dwq, Q);
}

// The queued (item, orientation) pairs of oct_validity_chunked_kernel<true, 2, true>: every leg by the filtered code.
__global__ __launch_bounds__(kOctBlock) void oct_deferred_kernel( // (every lane its own child and orientation: 150 VGPRs; a few per cent of the work)
    const OctChild* __restrict__ children, const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ fz,
    const LrmCompiledLeg* __restrict__ legs, const float4* __restrict__ spheres, int leg_count, int legs_for_stab, float convex_r2,
    uint32_t* __restrict__ flags, const OctDeferQueue Q) {
    // A queue that overflowed has holes (a flush that did not fit wrote nothing, later ones may have): nothing of it is read -- the
    // host sees the flag and runs the level again with the doubts in place.
    if (Q.count[1] != 0u) return;
    const uint32_t n = Q.count[0] < Q.cap ? Q.count[0] : Q.cap;
    for (uint32_t k = blockIdx.x * kOctBlock + threadIdx.x; k < n; k += gridDim.x * kOctBlock) {
        const uint32_t child = Q.rec[3u * k], f = Q.rec[3u * k + 1u];
        const int a = (int)Q.rec[3u * k + 2u];
        if (*reinterpret_cast<volatile uint32_t*>(&flags[child]) == 7u) continue;
        const OctChild ch = children[child];
        const LrmVec3 vect{fx[f] - ch.c[0], fy[f] - ch.c[1], fz[f] - ch.c[2]};
        const float h2 = ch.h[0] * ch.h[0] + ch.h[1] * ch.h[1] + ch.h[2] * ch.h[2];
        const float hd = sqrtf(h2 + fmaxf(ch.margin, 0.f)) * 1.0001f + 0.01f;
        const uint32_t bits = oct_item_flags<true, 0>(ch, vect, h2, hd, legs, nullptr, nullptr, spheres, leg_count, legs_for_stab, convex_r2, a, a + 1);
        if (bits) atomicOr(&flags[child], bits);
    }
}

// ---- Morton order on the device ----------------------------------------------------------------------
// The chunk-culled kernel wants consecutive footholds to be neighbours in space.  (The first version sorted on the
// host: 0.35 s of the 0.44 s a 4e6-foothold call took.)  Keys as lrm_morton_order: 10 bits per axis of the cloud's
// bounding box, interleaved; hipcub radix sort of (key, index); gather.  The flags are ORs over footholds, so the
// order -- and the order among equal keys -- changes no result.
__global__ __launch_bounds__(kOctBlock) void oct_aos_to_soa_kernel(const float* __restrict__ aos, size_t nf, float* __restrict__ x,
                                                                  float* __restrict__ y, float* __restrict__ z) {
    const size_t i = (size_t)blockIdx.x * kOctBlock + threadIdx.x;
    if (i >= nf) return;
    x[i] = aos[3 * i];
    y[i] = aos[3 * i + 1];
    z[i] = aos[3 * i + 2];
}
struct OctBounds {
    float lo[3], inv[3]; // inv = 1 / span, 0 for a flat axis
};
__device__ __forceinline__ uint32_t oct_spread10(uint32_t v) { // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(kOctBlock) void oct_keys_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ z, size_t nf, OctBounds b,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * kOctBlock + threadIdx.x;
    if (i >= nf) return;
    const float p[3] = {x[i], y[i], z[i]};
    uint32_t k = 0;
    for (int a = 0; a < 3; a++) {
        float t = (p[a] - b.lo[a]) * b.inv[a];
        if (!(t >= 0.f)) t = 0.f; // nan / below
        if (t > 1.f) t = 1.f;
        k |= oct_spread10((uint32_t)(t * 1023.f)) << a;
    }
    keys[i] = k;
    idx[i] = (uint32_t)i;
}
__global__ __launch_bounds__(kOctBlock) void oct_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z, const uint32_t* __restrict__ idx, size_t nf,
                                                              float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz) {
    const size_t i = (size_t)blockIdx.x * kOctBlock + threadIdx.x;
    if (i >= nf) return;
    const uint32_t j = idx[i];
    sx[i] = x[j];
    sy[i] = y[j];
    sz[i] = z[j];
}

// ---- host side ------------------------------------------------------------------------------
struct Quat {
    float x, y, z, w;
};
Quat q_mul(Quat a, Quat b) { // qtMultiply, unified_math_cuda.cu.h:40-46
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}
Quat q_axis(float ax, float ay, float az, float angle) { // quatFromVectAngle, :48-57
    float s, c;
    sincosf(angle / 2, &s, &c);
    const float mag = sqrtf(ax * ax + ay * ay + az * az);
    return Quat{s, c * ax / mag, c * ay / mag, c * az / mag};
}
// QuaternionFromAngleIndex + RPYtoQuat, octree_util.cu.h:164-198
Quat quat_from_angle_index(unsigned idx, const LrmOctreeSettings& st) {
    float rpy[3];
    unsigned red = idx;
    for (int i = 0; i < 3; i++) {
        const unsigned char max_ind = (unsigned char)st.angle_sample[i];
        unsigned char ind = (unsigned char)(red % max_ind);
        ind = (unsigned char)((ind + (ind / 2)) % max_ind); // the reference's "starts at middle" ordering
        red = red / max_ind;
        const int den = (max_ind - 1) > 1 ? (max_ind - 1) : 1;
        const float x = (float)ind / (unsigned char)den;
        rpy[i] = (1 - x) * st.angle_minmax[i * 2] + x * st.angle_minmax[i * 2 + 1];
    }
    Quat q = q_mul(q_axis(0, 1, 0, rpy[1]), q_axis(1, 0, 0, rpy[0]));
    return q_mul(q_axis(0, 0, 1, rpy[2]), q);
}

struct Node {
    float c[3], h[3];
    bool validity = false, leaf = false, raw = false, on_edge = false, dead = false;
    int first_child = -1; // index of 8 consecutive children, -1 if none
};

// CreateChildBox (octree_util.cu.h:105-151) for SUB_QUAD = 1, quadCount = 3, no "small" dimension.
// Returns false for a dead quadrant.  *missing = number of dimensions too small to split.
bool create_child_box(const Node& parent, unsigned index, float min_box, Node* child, int* missing) {
    // the 3-bit child index, bit-reversed: bit 0 of `quadr` is bit 2 of the index, etc.
    unsigned quadr = ((index & 1u) << 2) | (index & 2u) | ((index >> 2) & 1u);
    std::memcpy(child->c, parent.c, sizeof child->c);
    std::memcpy(child->h, parent.h, sizeof child->h);
    float div[3] = {2, 2, 2};
    unsigned upper = 2;
    *missing = 0;
    for (unsigned q = 0; q < 3; q++) {
        if (child->h[q] < min_box) {
            (*missing)++;
            if ((quadr >> upper) & 1u) return false; // DEADQUADRAN: this half does not exist
            // bitShiftBetween(quadr, q, 3, 1): bits q..3 move up by one (bit 3 drops out)
            const unsigned mask = ((1u << 4) - 1u) ^ ((1u << q) - 1u);
            quadr = (quadr & ~mask) | (((quadr & mask) << 1) & mask);
            div[q] = 1;
        }
    }
    for (int q = 0; q < 3; q++) {
        const float old = child->h[q];
        child->h[q] = old / div[q];
        const float move = old - child->h[q];
        child->c[q] = child->c[q] + (((quadr >> q) & 1u) ? move * -1 : move); // flipVectorOnQuad
    }
    return true;
}

thread_local std::string g_oct_err;
// lrm_dbg_oct_trace: every evaluated child of every level of the calls that follow, with the flag bits its kernel
// returned (tests compare a sample of them with the oracle): {c[3], h[3], parent h[3], flags, parent_valid + 2 rot + 4 skip, depth}
thread_local bool g_oct_trace = false;
thread_local std::vector<float> g_oct_trace_recs;

} // namespace

extern "C" {

void lrm_octree_default_settings(LrmOctreeSettings* s) {
    const float pi = 3.14159265358979323846264338327950288419716939937510582097f;
    std::memset(s, 0, sizeof *s);
    for (int i = 0; i < 3; i++) {
        s->box_center[i] = 0.f;   // settings.h:24
        s->box_size[i] = 5000.f;  // settings.h:26
        s->angle_sample[i] = 3;   // settings.h:35
    }
    s->min_box = 100.f;           // settings.h:17
    s->enable_rot_below = 50.f;   // settings.h:33
    s->convex_radius = 100.f;     // settings.h:34
    const float mm[6] = {-pi / 4, pi / 4, -pi / 8, pi / 8, -pi / 8, pi / 8}; // settings.h:38
    std::memcpy(s->angle_minmax, mm, sizeof mm);
    s->leg_count = 4;             // settings.h:41
    for (int l = 0; l < 4; l++) s->leg_mount[l] = pi / 4 * l; // settings.h:42
    s->leg_number_for_stab = 4;   // settings.h:46
    s->max_depth = 1;             // settings.h:15 MAX_DEPTH
}

const char* lrm_octree_last_error(void) { return g_oct_err.c_str(); }

int lrm_dbg_oct_trace(int enable) {
    g_oct_trace = enable != 0;
    g_oct_trace_recs.clear();
    return 0;
}
int lrm_dbg_oct_trace_read(float* out, size_t capacity_records, size_t* n_out) {
    if (!n_out) return -1;
    *n_out = g_oct_trace_recs.size() / 12;
    if (out && capacity_records >= *n_out) std::memcpy(out, g_oct_trace_recs.data(), g_oct_trace_recs.size() * sizeof(float));
    return 0;
}

// apply_oct, several_leg_octree.cu:391-488
static int apply_oct_impl(const float* footholds, const float* dev_x, const float* dev_y, const float* dev_z, size_t nf,
                          const LrmLegDimensions* dim, const LrmOctreeSettings* st_in, float* centers_out, size_t capacity,
                          size_t* n_out, float* ms, int rank, int world, LrmOctExchange exchange, void* user);

int lrm_apply_oct(const float* footholds, size_t nf, const LrmLegDimensions* dim, const LrmOctreeSettings* st_in,
                  float* centers_out, size_t capacity, size_t* n_out, float* ms) {
    return lrm_apply_oct_sharded(footholds, nf, dim, st_in, centers_out, capacity, n_out, ms, 0, 1, nullptr, nullptr);
}
int lrm_apply_oct_sharded(const float* footholds, size_t nf, const LrmLegDimensions* dim, const LrmOctreeSettings* st_in,
                          float* centers_out, size_t capacity, size_t* n_out, float* ms, int rank, int world,
                          LrmOctExchange exchange, void* user) {
    if (nf && !footholds) { g_oct_err = "null argument"; return LRM_EINVAL; }
    return apply_oct_impl(footholds, nullptr, nullptr, nullptr, nf, dim, st_in, centers_out, capacity, n_out, ms, rank, world, exchange, user);
}
int lrm_apply_oct_dev(const float* fx, const float* fy, const float* fz, size_t nf, const LrmLegDimensions* dim,
                      const LrmOctreeSettings* st_in, float* centers_out, size_t capacity, size_t* n_out, float* ms, int rank,
                      int world, LrmOctExchange exchange, void* user) {
    if (nf && !(fx && fy && fz)) { g_oct_err = "null argument"; return LRM_EINVAL; }
    return apply_oct_impl(nullptr, fx, fy, fz, nf, dim, st_in, centers_out, capacity, n_out, ms, rank, world, exchange, user);
}

// Foothold-partitioned tree (config 5 at scale: no rank holds the whole cloud): every rank passes ITS part of the footholds
// (any disjoint split; a spatial one keeps the work local) and evaluates every child against it; the three flags of a
// child are ORs over footholds, so the OR of the ranks' flag words is exactly the flag word of the whole cloud.
int lrm_apply_oct_partitioned(const float* footholds, size_t nf_local, const LrmLegDimensions* dim, const LrmOctreeSettings* st_in,
                              float* centers_out, size_t capacity, size_t* n_out, float* ms, LrmOctExchange exchange, void* user) {
    if ((nf_local && !footholds) || !exchange) { g_oct_err = "null argument"; return LRM_EINVAL; }
    return apply_oct_impl(footholds, nullptr, nullptr, nullptr, nf_local, dim, st_in, centers_out, capacity, n_out, ms, 0, 1, exchange, user);
}
int lrm_apply_oct_partitioned_dev(const float* fx, const float* fy, const float* fz, size_t nf_local, const LrmLegDimensions* dim,
                                  const LrmOctreeSettings* st_in, float* centers_out, size_t capacity, size_t* n_out, float* ms,
                                  LrmOctExchange exchange, void* user) {
    if ((nf_local && !(fx && fy && fz)) || !exchange) { g_oct_err = "null argument"; return LRM_EINVAL; }
    return apply_oct_impl(nullptr, fx, fy, fz, nf_local, dim, st_in, centers_out, capacity, n_out, ms, 0, 1, exchange, user);
}

// The same tree on `world` GPUs (one process each).  rank / world: the children of every level are dealt round-robin to the
// ranks and each rank evaluates its share against (its copy of) all footholds; world = 1 with an exchange: every rank
// evaluates every child against its own part of the footholds (lrm_apply_oct_partitioned).  Either way `exchange` combines
// the flag words with a bitwise OR, every rank builds the identical tree and returns all valid leaves.
// Failure protocol: the ranks exchange once per level and once (one word) before the first level.  A rank that fails
// locally still takes part in the exchange its peers are waiting in, with kOctPoison in every word, and a rank that
// reads kOctPoison fails too: nobody is left waiting in a collective.
static int apply_oct_impl(const float* footholds /* host AoS, or null */, const float* dev_x, const float* dev_y, const float* dev_z,
                          size_t nf, const LrmLegDimensions* dim, const LrmOctreeSettings* st_in,
                          float* centers_out, size_t capacity, size_t* n_out, float* ms, int rank, int world,
                          LrmOctExchange exchange, void* user) {
    auto fail = [](int code, const char* w) { g_oct_err = w; return code; };
    if (!dim || !n_out || (nf && !footholds && !(dev_x && dev_y && dev_z)) || (capacity && !centers_out)) return fail(LRM_EINVAL, "null argument");
    if (nf >= ((size_t)1 << 31)) return fail(LRM_EINVAL, "more than 2^31 - 1 footholds: shard the cloud");
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !exchange)) return fail(LRM_EINVAL, "bad rank / world / exchange");
    constexpr uint32_t kOctPoison = 0xffffffffu;
    size_t peers_expect = exchange ? 1 : 0; // words the peers' next exchange carries (0: none pending)
    auto abort_peers = [&]() {              // called on a local failure: do not leave the peers waiting
        if (!exchange || peers_expect == 0) return;
        std::vector<uint32_t> poison(peers_expect, kOctPoison);
        peers_expect = 0;
        (void)exchange(poison.data(), poison.size(), user);
    };
    LrmOctreeSettings st;
    if (st_in) st = *st_in;
    else lrm_octree_default_settings(&st);
    if (st.leg_count < 1 || st.leg_count > LRM_MAX_LEGS || st.leg_number_for_stab > st.leg_count || st.max_depth < 0 ||
        st.angle_sample[0] < 1 || st.angle_sample[1] < 1 || st.angle_sample[2] < 1)
        return fail(LRM_EINVAL, "bad octree settings");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LRM_ENODEV, "no HIP device");
#define OCT_TRY(expr, where)                                                   \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            g_oct_err = std::string(where) + ": " + hipGetErrorString(e_);     \
            abort_peers();                                                     \
            cleanup();                                                         \
            return (e_ == hipErrorOutOfMemory) ? LRM_ENOMEM : LRM_ENODEV;      \
        }                                                                      \
    } while (0)
    float *d_f = nullptr, *d_boxes = nullptr, *d_tmp_aos = nullptr, *d_tmp_soa = nullptr;
    uint32_t* d_keys = nullptr;
    void* d_sort_tmp = nullptr;
    float4* d_spheres = nullptr;
    LrmTolLeg* d_tols = nullptr;
    const uint8_t** d_tabs = nullptr; // device array of the plane tables' device pointers (the tables themselves live in the cache below)
    uint32_t *d_defer = nullptr, *d_defer_count = nullptr; // the queue of deferred (item, orientation) pairs (oct_item_flags_defer)
    unsigned long long deferred_total = 0;
    LrmCompiledLeg* d_legs = nullptr;
    OctChild* d_children = nullptr;
    uint32_t* d_flags = nullptr;
    size_t children_cap = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    auto cleanup = [&]() {
        if (d_f) (void)hipFree(d_f);
        if (d_boxes) (void)hipFree(d_boxes);
        if (d_tmp_aos) (void)hipFree(d_tmp_aos);
        if (d_tmp_soa) (void)hipFree(d_tmp_soa);
        if (d_keys) (void)hipFree(d_keys);
        if (d_sort_tmp) (void)hipFree(d_sort_tmp);
        if (d_legs) (void)hipFree(d_legs);
        if (d_spheres) (void)hipFree(d_spheres);
        if (d_tols) (void)hipFree(d_tols);
        if (d_tabs) (void)hipFree(d_tabs);
        if (d_defer) (void)hipFree(d_defer);
        if (d_defer_count) (void)hipFree(d_defer_count);
        if (d_children) (void)hipFree(d_children);
        if (d_flags) (void)hipFree(d_flags);
        if (ev_a) (void)hipEventDestroy(ev_a);
        if (ev_b) (void)hipEventDestroy(ev_b);
    };

    const bool dbg = getenv("LRM_OCT_DEBUG") != nullptr; // host-side phase times on stderr
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_phase = now();
    // footholds as SoA on the device, in Morton order (the flags are ORs over footholds: the order changes nothing,
    // but consecutive footholds then fill compact boxes for oct_validity_chunked_kernel)
    const size_t ntiles = (nf + 1023) / 1024;
    OCT_TRY(hipMalloc(&d_f, 3 * (nf ? nf : 1) * sizeof(float)), "hipMalloc gpu_in.elements");
    if (nf) {
        const unsigned pb = (unsigned)((nf + kOctBlock - 1) / kOctBlock);
        const float *ux = dev_x, *uy = dev_y, *uz = dev_z;
        if (footholds) { // host AoS (the reference's Array<float3>): upload, split
            OCT_TRY(hipMalloc(&d_tmp_aos, 3 * nf * sizeof(float)), "hipMalloc footholds");
            OCT_TRY(hipMalloc(&d_tmp_soa, 3 * nf * sizeof(float)), "hipMalloc footholds");
            OCT_TRY(hipMemcpy(d_tmp_aos, footholds, 3 * nf * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy gpu_in.elements");
            hipLaunchKernelGGL(oct_aos_to_soa_kernel, dim3(pb), dim3(kOctBlock), 0, nullptr, d_tmp_aos, nf, d_tmp_soa, d_tmp_soa + nf, d_tmp_soa + 2 * nf);
            OCT_TRY(hipGetLastError(), "Kernel launch");
            ux = d_tmp_soa;
            uy = d_tmp_soa + nf;
            uz = d_tmp_soa + 2 * nf;
        }
        // bounding box of the cloud: the tile boxes of the unsorted cloud, reduced on the host
        OCT_TRY(hipMalloc(&d_boxes, ntiles * 17 * 6 * sizeof(float)), "hipMalloc foothold boxes");
        hipLaunchKernelGGL(oct_boxes_kernel, dim3((unsigned)ntiles), dim3(kOctBlock), 0, nullptr, ux, uy, uz, nf, ntiles, d_boxes);
        OCT_TRY(hipGetLastError(), "Kernel launch");
        std::vector<float> tb(ntiles * 6);
        OCT_TRY(hipMemcpy(tb.data(), d_boxes, tb.size() * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy boxes");
        OctBounds ob;
        for (int a = 0; a < 3; a++) {
            float lo = 3.0e38f, hi = -3.0e38f;
            for (size_t t = 0; t < ntiles; t++) {
                lo = std::min(lo, tb[t * 6 + a]);
                hi = std::max(hi, tb[t * 6 + 3 + a]);
            }
            ob.lo[a] = lo;
            ob.inv[a] = (hi > lo && std::isfinite(hi - lo)) ? 1.0f / (hi - lo) : 0.f;
        }
        OCT_TRY(hipMalloc(&d_keys, 4 * nf * sizeof(uint32_t)), "hipMalloc sort keys"); // keys, indices, and their sorted copies
        uint32_t *k_in = d_keys, *i_in = d_keys + nf, *k_out = d_keys + 2 * nf, *i_out = d_keys + 3 * nf;
        hipLaunchKernelGGL(oct_keys_kernel, dim3(pb), dim3(kOctBlock), 0, nullptr, ux, uy, uz, nf, ob, k_in, i_in);
        OCT_TRY(hipGetLastError(), "Kernel launch");
        size_t sort_bytes = 0;
        OCT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k_in, k_out, i_in, i_out, (int)nf, 0, 30, (hipStream_t) nullptr), "radix sort (size)");
        OCT_TRY(hipMalloc(&d_sort_tmp, sort_bytes ? sort_bytes : 16), "hipMalloc sort workspace");
        OCT_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, sort_bytes, k_in, k_out, i_in, i_out, (int)nf, 0, 30, (hipStream_t) nullptr), "radix sort");
        hipLaunchKernelGGL(oct_gather_kernel, dim3(pb), dim3(kOctBlock), 0, nullptr, ux, uy, uz, i_out, nf, d_f, d_f + nf, d_f + 2 * nf);
        OCT_TRY(hipGetLastError(), "Kernel launch");
        OCT_TRY(hipDeviceSynchronize(), "footholds in Morton order");
        (void)hipFree(d_sort_tmp); d_sort_tmp = nullptr;
        (void)hipFree(d_keys); d_keys = nullptr;
        if (d_tmp_aos) { (void)hipFree(d_tmp_aos); d_tmp_aos = nullptr; }
        if (d_tmp_soa) { (void)hipFree(d_tmp_soa); d_tmp_soa = nullptr; }
        (void)hipFree(d_boxes); d_boxes = nullptr;
    }
    if (nf) {
        OCT_TRY(hipMalloc(&d_boxes, ntiles * 17 * 6 * sizeof(float)), "hipMalloc foothold boxes");
        hipLaunchKernelGGL(oct_boxes_kernel, dim3((unsigned)ntiles), dim3(kOctBlock), 0, nullptr, d_f, d_f + nf, d_f + 2 * nf, nf, ntiles, d_boxes);
        OCT_TRY(hipGetLastError(), "Kernel launch");
    }

    if (dbg) { (void)hipDeviceSynchronize(); fprintf(stderr, "apply_oct: %zu footholds ordered on the device in %.2f ms\n", nf, ms_since(t_phase)); t_phase = now(); }
    // compiled legs for every (orientation sample, mounted leg)
    const int n_angles_max = st.angle_sample[0] * st.angle_sample[1] * st.angle_sample[2];
    std::vector<LrmCompiledLeg> legs((size_t)n_angles_max * st.leg_count);
    bool all_fast = true;
    for (int a = 0; a < n_angles_max; a++) {
        const Quat q = quat_from_angle_index((unsigned)a, st);
        const float qa[4] = {q.x, q.y, q.z, q.w};
        for (int l = 0; l < st.leg_count; l++) {
            LrmLegDimensions leg = *dim;
            leg.body_angle = st.leg_mount[l]; // several_leg_octree.cu:94
            lrm_compile_leg(leg, qa, 1, &legs[(size_t)a * st.leg_count + l]);
            all_fast = all_fast && legs[(size_t)a * st.leg_count + l].fast_ok;
        }
    }
    const bool fast = all_fast && lrm_get_mode() != LRM_MODE_STRICT;
    // the tolerance blocks (filtered mode; LRM_OCT_TOL=0: without -- A/B runs and tests): lrm_compile_tol costs ~0.3 ms of host
    // geometry per (orientation, leg), so they are worth it from a few hundred thousand footholds on and are kept across calls
    {
        const char* e = getenv("LRM_OCT_TOL");
        const bool want = fast && (e ? e[0] != '0' : nf >= 300000);
        if (want) {
            static std::mutex mu;
            struct Entry {
                LrmTolLeg tl;
                std::map<int, uint8_t*> tab_dev; // per device: the plane table (lrm_build_tol_tab_dev), nullptr = this leg has none
            };
            static std::map<std::array<float, 18>, Entry> cache;
            std::vector<LrmTolLeg> tols(legs.size());
            std::vector<const uint8_t*> tabs(legs.size(), nullptr);
            // the plane tables (LRM_OCT_TAB=0: without): built on the device on first use, ~0.4 ms each, kept across calls
            const char* et = getenv("LRM_OCT_TAB");
            const bool want_tab = et ? et[0] != '0' : true;
            int dev = 0;
            (void)hipGetDevice(&dev);
            bool any_tab = false;
            std::lock_guard<std::mutex> g(mu);
            for (int a = 0; a < n_angles_max; a++) {
                const Quat q = quat_from_angle_index((unsigned)a, st);
                for (int l = 0; l < st.leg_count; l++) {
                    LrmLegDimensions leg = *dim;
                    leg.body_angle = st.leg_mount[l];
                    std::array<float, 18> key;
                    std::memcpy(key.data(), &leg, 14 * sizeof(float));
                    key[14] = q.x; key[15] = q.y; key[16] = q.z; key[17] = q.w;
                    auto it = cache.find(key);
                    if (it == cache.end()) {
                        // A full cache takes no new entry (and frees nothing: another thread's call may be running on its tables): this
                        // pair goes without a table -- its items through the filtered code (4096 pairs: a thousand robots' worth).
                        if (cache.size() >= 4096) {
                            lrm_compile_tol(legs[(size_t)a * st.leg_count + l], &tols[(size_t)a * st.leg_count + l]);
                            continue;
                        }
                        Entry en;
                        lrm_compile_tol(legs[(size_t)a * st.leg_count + l], &en.tl);
                        it = cache.emplace(key, en).first;
                    }
                    tols[(size_t)a * st.leg_count + l] = it->second.tl;
                    if (want_tab && it->second.tl.tol_ok) {
                        auto dt = it->second.tab_dev.find(dev);
                        if (dt == it->second.tab_dev.end()) {
                            uint8_t* t = nullptr;
                            size_t bytes = 0;
                            float ms = 0.f;
                            const int rc = lrm_build_tol_tab_dev(it->second.tl, nullptr, &t, &bytes, &ms);
                            if (rc < 0) OCT_TRY((hipError_t)(-rc), "plane table (device builder)");
                            dt = it->second.tab_dev.emplace(dev, rc == 0 ? t : nullptr).first;
                        }
                        tabs[(size_t)a * st.leg_count + l] = dt->second;
                        any_tab = any_tab || dt->second != nullptr;
                    }
                }
            }
            OCT_TRY(hipMalloc(&d_tols, tols.size() * sizeof(LrmTolLeg)), "hipMalloc tolerance blocks");
            OCT_TRY(hipMemcpy(d_tols, tols.data(), tols.size() * sizeof(LrmTolLeg), hipMemcpyHostToDevice), "hipMemcpy tolerance blocks");
            if (any_tab) {
                OCT_TRY(hipMalloc(reinterpret_cast<void**>(&d_tabs), tabs.size() * sizeof(uint8_t*)), "hipMalloc table pointers");
                OCT_TRY(hipMemcpy(d_tabs, tabs.data(), tabs.size() * sizeof(uint8_t*), hipMemcpyHostToDevice), "hipMemcpy table pointers");
                const char* ed = getenv("LRM_OCT_DEFER"); // LRM_OCT_DEFER=0: the doubts of the table form inline (A/B runs and tests)
                if (!ed || ed[0] != '0') {
                    OCT_TRY(hipMalloc(reinterpret_cast<void**>(&d_defer), oct_defer_cap(nf) * 3 * sizeof(uint32_t)), "hipMalloc deferred pairs");
                    OCT_TRY(hipMalloc(reinterpret_cast<void**>(&d_defer_count), 8 * sizeof(uint32_t)), "hipMalloc deferred count");
                }
            }
        }
    }
    OCT_TRY(hipMalloc(&d_legs, legs.size() * sizeof(LrmCompiledLeg)), "hipMalloc legs");
    OCT_TRY(hipMemcpy(d_legs, legs.data(), legs.size() * sizeof(LrmCompiledLeg), hipMemcpyHostToDevice), "hipMemcpy legs");
    // bounding sphere of what each (orientation, leg) reaches, in the frame distance_global takes its point in:
    // pair_center lives in the frame of reachable_rotate_leg (the target relative to the body, before the quaternion);
    // distance_global first applies qtRotate(qtInvert(q), .), so the centre goes through qtRotate(q, .)
    std::vector<float4> spheres(legs.size());
    const bool no_cull = getenv("LRM_OCT_NOCULL") && getenv("LRM_OCT_NOCULL")[0] == '1'; // tests: every work item evaluated
    for (size_t k = 0; k < legs.size(); k++) {
        const LrmCompiledLeg& L = legs[k];
        const LrmVec3 c = lrm_qrot(L.fwd_rot, LrmVec3{L.pair_center[0], L.pair_center[1], L.pair_center[2]});
        spheres[k] = make_float4(c.x, c.y, c.z, no_cull ? 1.0e18f : std::sqrt(L.pair_r2) * 1.0001f + 0.5f);
    }
    OCT_TRY(hipMalloc(&d_spheres, spheres.size() * sizeof(float4)), "hipMalloc spheres");
    OCT_TRY(hipMemcpy(d_spheres, spheres.data(), spheres.size() * sizeof(float4), hipMemcpyHostToDevice), "hipMemcpy spheres");
    OCT_TRY(hipEventCreate(&ev_a), "hipEventCreate");
    OCT_TRY(hipEventCreate(&ev_b), "hipEventCreate");

    if (dbg) { fprintf(stderr, "apply_oct: %zu legs compiled and uploaded in %.2f ms\n", legs.size(), ms_since(t_phase)); t_phase = now(); }
    if (exchange) { // every rank got this far?
        uint32_t ready = 0;
        peers_expect = 0;
        if (exchange(&ready, 1, user) != 0 || ready == kOctPoison) {
            g_oct_err = ready == kOctPoison ? "a peer rank failed before the first level" : "the exchange callback failed";
            cleanup();
            return LRM_ENODEV;
        }
    }
    std::vector<Node> nodes(1);
    for (int i = 0; i < 3; i++) {
        nodes[0].c[i] = st.box_center[i];
        nodes[0].h[i] = st.box_size[i];
    }
    nodes[0].raw = true;
    const float reach_len = dim->body + dim->coxa_length + dim->femur_length + dim->tibia_length;
    std::vector<int> expand{0}; // raw nodes to refine in this iteration
    float total_ms = 0.f;
    for (int depth = 0; depth < st.max_depth && !expand.empty(); depth++) {
        std::vector<OctChild> level;
        std::vector<int> level_nodes;
        for (int pi : expand) {
            const int first = (int)nodes.size();
            nodes.resize(nodes.size() + 8);
            Node& parent = nodes[pi];
            parent.first_child = first;
            const bool rot = parent.h[0] < st.enable_rot_below; // several_leg_octree.cu:52
            for (unsigned ci = 0; ci < 8; ci++) {
                Node& n = nodes[first + ci];
                int missing = 0;
                if (!create_child_box(parent, ci, st.min_box, &n, &missing)) { // several_leg_octree.cu:331-339
                    n.dead = n.leaf = n.validity = n.on_edge = true;
                    n.raw = false;
                    std::memset(n.c, 0, sizeof n.c);
                    std::memset(n.h, 0, sizeof n.h);
                } else if (3 - missing <= 0) { // :342-347
                    n.leaf = true;
                    n.raw = false;
                } else {                       // :348-353
                    n.leaf = false;
                    n.raw = true;
                }
                OctChild oc;
                std::memcpy(oc.c, n.c, sizeof oc.c);
                std::memcpy(oc.h, n.h, sizeof oc.h);
                std::memcpy(oc.ph, parent.h, sizeof oc.ph);
                oc.margin = rot ? 0.f : st.enable_rot_below / 3;
                oc.n_angles = rot ? n_angles_max : 1;
                oc.parent_valid = parent.validity;
                oc.skip = n.validity; // dead quadrants are "valid" and skipped (:58-61)
                oc.pad = 0;
                level.push_back(oc);
                level_nodes.push_back(first + ci);
            }
            nodes[pi].raw = false;
        }
        const size_t nc = level.size();
        peers_expect = exchange ? nc : 0;
        if (nc > children_cap) {
            if (d_children) (void)hipFree(d_children);
            if (d_flags) (void)hipFree(d_flags);
            d_children = nullptr;
            d_flags = nullptr;
            OCT_TRY(hipMalloc(&d_children, nc * sizeof(OctChild)), "hipMalloc children");
            OCT_TRY(hipMalloc(&d_flags, nc * sizeof(uint32_t)), "hipMalloc flags");
            children_cap = nc;
        }
        if (world > 1) { // this rank's share of the level: the others' children are skipped on the device only
            std::vector<OctChild> mine(level);
            for (size_t k = 0; k < nc; k++)
                if ((int)(k % (size_t)world) != rank) mine[k].skip = 1;
            OCT_TRY(hipMemcpy(d_children, mine.data(), nc * sizeof(OctChild), hipMemcpyHostToDevice), "hipMemcpy children");
        } else {
            OCT_TRY(hipMemcpy(d_children, level.data(), nc * sizeof(OctChild), hipMemcpyHostToDevice), "hipMemcpy children");
        }
        OCT_TRY(hipMemset(d_flags, 0, nc * sizeof(uint32_t)), "hipMemset flags");
        std::vector<uint32_t> flags(nc, 0);
        if (nf) {
            OCT_TRY(hipEventRecord(ev_a, nullptr), "hipEventRecord");
            const float cr2 = st.convex_radius * st.convex_radius;
            // LRM_OCT_BRUTE=1 (tests): every level with the every-foothold kernel, for comparison
            const bool brute = getenv("LRM_OCT_BRUTE") && getenv("LRM_OCT_BRUTE")[0] == '1' && nc <= 65535;
            size_t chunked_from = (size_t)kOctChunkedFrom;
            if (const char* e = getenv("LRM_OCT_CHUNKED_FROM")) chunked_from = (size_t)atol(e); // experiments
            // (the first level's 8 huge children too when the cloud has enough tiles to spread them over the chip: 6.4 -> 4.0 ms at 1.25e7 footholds)
            if ((nc >= chunked_from || (!getenv("LRM_OCT_CHUNKED_FROM") && nc * ((ntiles + 31) / 32) >= 2048)) && !brute) {
                // many small children: one workgroup per child, only the footholds of nearby chunks
                // workgroups per child: at least ~4096 workgroups in flight, at most one per round of 256 tiles
                // (levels of few, huge children: rounds of fewer tiles, so that more workgroups share a child and look at its flags more often)
                // config-5 share, levels of 64 / 256 / 976 / 4112 children: 11.6 / 9.2 / 9.9 / 14.3 ms with rounds of 256 tiles and 4096
                // workgroups -> 5.3 / 6.1 / 8.4 / 13.4 ms (profiles/r04_octree.txt); the deepest level (15 856 children) is best left alone
                size_t tpr = nc < 2048 ? 32 : (nc < 8192 ? 128 : kOctBlock);
                if (const char* e = getenv("LRM_OCT_TPR")) tpr = std::min<size_t>(std::max<size_t>((size_t)atol(e), 1), kOctBlock);
                size_t want_wgs = nc < 2048 ? 32768 : (nc < 8192 ? 8192 : 4096);
                if (const char* e = getenv("LRM_OCT_WGS")) want_wgs = (size_t)atol(e);
                size_t splits = nc >= want_wgs ? 1 : (want_wgs + nc - 1) / nc;
                splits = std::min(splits, std::max<size_t>(1, (ntiles + tpr - 1) / tpr));
                splits = std::min<size_t>(splits, 256);
                const dim3 grid((unsigned)std::min<size_t>(nc * splits, (size_t)256 * 64));
#define LRM_OCT_CHUNKED(FAST, TOL) hipLaunchKernelGGL((oct_validity_chunked_kernel<FAST, TOL>), grid, dim3(kOctBlock), 0, nullptr, d_children, (int)nc, d_f, d_f + nf, \
                                       d_f + 2 * nf, nf, d_boxes, ntiles, d_legs, d_tols, d_tabs, d_spheres, st.leg_count, st.leg_number_for_stab, reach_len, cr2, d_flags, (uint32_t)splits, (uint32_t)tpr, OctDeferQueue{nullptr, nullptr, 0u})
                bool deferred_done = false;
                // the table form with its doubts queued for a second launch (see oct_item_flags_defer) -- from 256 children on: the few huge
                // children of the first levels saturate their flags early, and later when some of their items wait in the queue
                const size_t defer_from = getenv("LRM_OCT_DEFER_FROM") ? (size_t)atol(getenv("LRM_OCT_DEFER_FROM")) : (size_t)256;
                if (fast && d_tabs && d_defer && nc >= defer_from) {
                    const OctDeferQueue Q{d_defer, d_defer_count, (uint32_t)oct_defer_cap(nf)};
                    OCT_TRY(hipMemsetAsync(d_defer_count, 0, 8 * sizeof(uint32_t), nullptr), "hipMemsetAsync");
                    hipLaunchKernelGGL((oct_validity_chunked_kernel<true, 2, true>), grid, dim3(kOctBlock), 0, nullptr, d_children, (int)nc, d_f, d_f + nf,
                                       d_f + 2 * nf, nf, d_boxes, ntiles, d_legs, d_tols, d_tabs, d_spheres, st.leg_count, st.leg_number_for_stab, reach_len, cr2, d_flags, (uint32_t)splits, (uint32_t)tpr, Q);
                    OCT_TRY(hipGetLastError(), "Kernel launch");
                    hipLaunchKernelGGL(oct_deferred_kernel, dim3(2048), dim3(kOctBlock), 0, nullptr, d_children, d_f, d_f + nf, d_f + 2 * nf, d_legs, d_spheres,
                                       st.leg_count, st.leg_number_for_stab, cr2, d_flags, Q);
                    OCT_TRY(hipGetLastError(), "Kernel launch");
                    uint32_t qc[8] = {0};
                    OCT_TRY(hipMemcpy(qc, d_defer_count, sizeof qc, hipMemcpyDeviceToHost), "hipMemcpy queue count");
#if defined(LRM_OCT_COUNT)
                    fprintf(stderr, "apply_oct: level %d: %u pairs evaluated, %u queued (overflow %u): no table %u, doubt %u, near face %u\n", depth, qc[2], qc[0], qc[1], qc[3], qc[4], qc[5]);
#endif
                    deferred_total += qc[0];
                    if (qc[1] == 0) deferred_done = true;
                    else OCT_TRY(hipMemset(d_flags, 0, nc * sizeof(uint32_t)), "hipMemset flags"); // the queue overflowed: the level again, doubts inline
                }
                if (deferred_done) {}
                else if (fast && d_tabs) LRM_OCT_CHUNKED(true, 2);
                else if (fast && d_tols) LRM_OCT_CHUNKED(true, 1);
                else if (fast) LRM_OCT_CHUNKED(true, 0);
                else LRM_OCT_CHUNKED(false, 0);
#undef LRM_OCT_CHUNKED
            } else {
                // few, huge children (the first levels): every foothold, spread over the chip; grid.y = children < 65
                size_t gx = (nf + kOctBlock - 1) / kOctBlock;
                if (gx > 1024) gx = 1024;
                const dim3 grid((unsigned)gx, (unsigned)nc);
#define LRM_OCT_EVERY(FAST, TOL) hipLaunchKernelGGL((oct_validity_kernel<FAST, TOL>), grid, dim3(kOctBlock), 0, nullptr, d_children, (int)nc, d_f, d_f + nf, \
                                       d_f + 2 * nf, nf, d_legs, d_tols, d_tabs, d_spheres, st.leg_count, st.leg_number_for_stab, reach_len, cr2, d_flags)
                if (fast && d_tabs) LRM_OCT_EVERY(true, 2);
                else if (fast && d_tols) LRM_OCT_EVERY(true, 1);
                else if (fast) LRM_OCT_EVERY(true, 0);
                else LRM_OCT_EVERY(false, 0);
#undef LRM_OCT_EVERY
            }
            OCT_TRY(hipGetLastError(), "Kernel launch");
            OCT_TRY(hipEventRecord(ev_b, nullptr), "hipEventRecord");
            OCT_TRY(hipMemcpy(flags.data(), d_flags, nc * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy flags");
            float e = 0.f;
            OCT_TRY(hipEventElapsedTime(&e, ev_a, ev_b), "hipEventElapsedTime");
            total_ms += e;
        }
        if (dbg) { fprintf(stderr, "apply_oct: level %d, %zu children: %.2f ms (host + kernel)\n", depth, nc, ms_since(t_phase)); t_phase = now(); }
        if (exchange) {
            peers_expect = 0;
            bool poisoned = exchange(flags.data(), nc, user) != 0;
            const bool cb_failed = poisoned;
            for (size_t k = 0; k < nc && !poisoned; k++) poisoned = flags[k] == kOctPoison;
            if (poisoned) {
                g_oct_err = cb_failed ? "the exchange callback failed" : "a peer rank failed";
                cleanup();
                return LRM_ENODEV;
            }
        }
        if (g_oct_trace)
            for (size_t k = 0; k < nc; k++) {
                const OctChild& oc = level[k];
                const float rec[12] = {oc.c[0], oc.c[1], oc.c[2], oc.h[0], oc.h[1], oc.h[2], oc.ph[0], oc.ph[1], oc.ph[2], (float)(flags[k] & 7u),
                                       (float)((oc.parent_valid ? 1 : 0) + (oc.n_angles > 1 ? 2 : 0) + (oc.skip ? 4 : 0)), (float)depth};
                g_oct_trace_recs.insert(g_oct_trace_recs.end(), rec, rec + 12);
            }
        // several_leg_octree.cu:134-150, with global ORs
        std::vector<int> next;
        for (size_t k = 0; k < nc; k++) {
            Node& n = nodes[level_nodes[k]];
            if (level[k].skip) continue;
            if (flags[k] & 1u) n.validity = true;
            if (flags[k] & 2u) n.leaf = true;
            if ((flags[k] & 4u) && !(flags[k] & 2u)) n.on_edge = true;
        }
        // the next host iteration descends (branchKernel "goDeeper", :296-312): a child that is not on
        // an edge becomes a leaf, the others are refined
        if (depth + 1 < st.max_depth) {
            for (size_t k = 0; k < nc; k++) {
                Node& n = nodes[level_nodes[k]];
                if (!n.on_edge) n.leaf = true;
                if (!n.leaf) next.push_back(level_nodes[k]);
            }
        }
        expand.swap(next);
    }
    // extractValidAsArray / fill_recus, octree_util.cu:128-180: depth first, child order
    size_t count = 0;
    std::vector<int> stack{0};
    // iterative DFS that visits children in index order
    struct Frame { int node; int next; };
    std::vector<Frame> st_frames{{0, 0}};
    while (!st_frames.empty()) {
        Frame& fr = st_frames.back();
        const Node& n = nodes[fr.node];
        if (n.first_child < 0 || fr.next >= 8) {
            st_frames.pop_back();
            continue;
        }
        const int ci = n.first_child + fr.next++;
        const Node& c = nodes[ci];
        const bool endpoint = !(c.leaf || c.raw || c.dead);
        const bool valid = !c.dead && (c.leaf || c.raw) && c.validity;
        if (endpoint) st_frames.push_back(Frame{ci, 0});
        else if (valid) {
            if (count < capacity) std::memcpy(centers_out + 3 * count, c.c, 3 * sizeof(float));
            count++;
        }
    }
    *n_out = count;
    if (ms) *ms = total_ms;
    if (dbg) { fprintf(stderr, "apply_oct: %zu nodes, leaves extracted in %.2f ms; %llu (item, orientation) pairs went through the deferred queue\n", nodes.size(), ms_since(t_phase), deferred_total); t_phase = now(); }
    cleanup();
    if (dbg) fprintf(stderr, "apply_oct: device memory released in %.2f ms\n", ms_since(t_phase));
#undef OCT_TRY
    if (count > capacity) return fail(LRM_EINVAL, "output capacity too small (n_out holds the required count)");
    return LRM_OK;
}

} // extern "C"

// lrm_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the reach / distance path.
//
// Layout in HBM: coordinates SoA (x[], y[], z[] float32), mask one byte per point (the
// reference's Array<bool>), optional ballot bit mask (uint64 per 64 points), distance field
// SoA.  AoS variants exist only for the drop-in apply_kernel boundary.
//
// The per-(leg, orientation) constants travel as ONE kernel argument (LrmCompiledLeg, by
// value): wave-uniform scalars are read through the scalar cache into SGPRs, and the only
// per-lane-indexed table (the 4 region circle lists, 256 B) is staged in LDS so that a lane
// fetches a circle with one ds_read_b128.  No MFMA: this is elementwise geometry.
//
// Compiled with -ffp-contract=off (see lrm_point.h).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include "lrm_launch.h"
#include "lrm_types.h"
#define LRM_FRESH(L) lrm_fresh(L)
#include "lrm_point.h"
#include "lrm_point_fast.h"

namespace {

constexpr int kBlock = 256;
// The kernels below take the compiled leg BY VALUE (it travels in the kernarg segment, no device
// allocation, graph-capturable) but read it through lrm_kernarg<>(offset) rather than through the
// parameter: the parameter's loads are hoisted to the top of the kernel, overflow the SGPR file
// (98 SGPR spills, ~150 v_readlane per point in the fused kernel), and its address cannot be
// taken without a 2.7 KB/lane scratch copy.  The kernarg segment is laid out like a C struct of
// the parameters in order (natural alignment), mirrored by KernargSoA / KernargAoS.
struct KernargSoA { // the leading parameters of the SoA kernels, in order
    const float *x, *y, *z;
    size_t n;
    LrmCompiledLeg L;
};
struct KernargAoS { // ... of the AoS kernels
    const float* xyz;
    size_t n;
    LrmCompiledLeg L;
};
constexpr unsigned kLegArgSoA = (unsigned)offsetof(KernargSoA, L), kLegArgAoS = (unsigned)offsetof(KernargAoS, L);
static_assert(kLegArgSoA == 32 && kLegArgAoS == 16, "kernarg layout");
#ifndef LRM_DIST_MIN_WAVES
// Waves per SIMD the filtered distance kernels are compiled for.  They are VALU-issue bound and the
// full issue rate needs many waves: 4 / 5 waves (128 / 96 VGPRs) -> 0.239 / 0.216 ms at steady clocks; with
// the circle loop of the filter unrolled by 2 instead of 4 (LRM_CIRCLE_UNROLL: fewer table values
// live at once) 6 / 7 / 8 waves -> 0.207 / 0.203 / 0.205 ms; at 7 waves (72 VGPRs) the compiler reports ScratchSize 0
// for all four distance kernels (`make resource-usage`; the fused one needs the opaque re-load of dist_soa_kernel).
#define LRM_DIST_MIN_WAVES 7
#endif
#ifndef LRM_DIST_STRICT_MIN_WAVES
#define LRM_DIST_STRICT_MIN_WAVES 5 // the strict kernels keep all four circles of a list in flight
#endif
#ifndef LRM_DIST_GRID_MULT
#define LRM_DIST_GRID_MULT 8 // workgroups launched per resident workgroup (see lrm_launch_dist_soa): 4 / 8 / 16 / 32 -> 0.220 / 0.216 / 0.218 / 0.223 ms
#endif
#ifndef LRM_REACH_GRID
#define LRM_REACH_GRID (256 * 32) // workgroups at most: 2048 / 8192 / 16384 / one quad per lane -> 27.0 / 24.4 / 25.3 / 25.3 us at 1e7 points
#endif
#ifndef LRM_REACH_MIN_WAVES
#define LRM_REACH_MIN_WAVES 8 // 64 VGPRs, no scratch: 24.4 us at 1e7 points against 26.0 at the compiler's choice (75 VGPRs, 6 waves)
#endif

// LDS image of the per-lane-indexed tables: circle lists, filter records, corner points (~2 KB)
struct LdsTables {
    LrmCircle lists[16];
    LrmCompiledLeg::LeanCircle lean[16];
    LrmCompiledLeg::DistCircle dist[16];
    LrmCircle corners[LRM_N_CORNERS];
};

__device__ __forceinline__ void stage_lists(const LrmCompiledLeg& L, LdsTables* t) {
    // 2 x 64 floats; one float of each table per thread of the first wave
    const float* src = reinterpret_cast<const float*>(&L.lists[0][0]);
    const float* lsrc = reinterpret_cast<const float*>(&L.lean[0][0]);
    const float* dsrc = reinterpret_cast<const float*>(&L.dist_tab[0][0]);
    const float* csrc = reinterpret_cast<const float*>(&L.corner_tab[0]);
    if (threadIdx.x < 64) {
        reinterpret_cast<float*>(t->lists)[threadIdx.x] = src[threadIdx.x];
        reinterpret_cast<float*>(t->lean)[threadIdx.x] = lsrc[threadIdx.x];
    }
    for (int i = threadIdx.x; i < (int)(sizeof(t->dist) / 4); i += kBlock) reinterpret_cast<float*>(t->dist)[i] = dsrc[i];
    if (threadIdx.x < (int)(sizeof(t->corners) / 4)) reinterpret_cast<float*>(t->corners)[threadIdx.x] = csrc[threadIdx.x];
    __syncthreads();
}

// mode dispatch: kFast selects the filtered evaluation (bit-identical, see lrm_point_fast.h)
template <bool kFast>
__device__ __forceinline__ bool eval_reach(const LrmCompiledLeg& L, const LdsTables* t, LrmVec3 p) {
    if (kFast) return lrm_reach_global_filtered(L, t->lists, t->lean, p);
    return lrm_reach_global(L, t->lists, p);
}
template <bool kFast>
__device__ __forceinline__ bool eval_dist(const LrmCompiledLeg& L, const LdsTables* t, LrmVec3& p) {
    if (kFast) return lrm_dist_global_filtered(L, LrmDistTables{t->lists, t->dist, t->corners}, p);
    return lrm_dist_global(L, t->lists, p);
}
// kOp 1: distance only; kOp 2: reach mask + distance (filtered mode: the mask is a by-product of
// the distance evaluation, lrm_reach_from_dist)
template <int kOp, bool kFast>
__device__ __forceinline__ bool eval_reach_dist(const LrmCompiledLeg& L, const LdsTables* t, LrmVec3& p, bool& reach) {
    if (kOp == 2 && kFast)
        return lrm_reach_dist_global_filtered(L, LrmDistTables{t->lists, t->dist, t->corners}, p, reach);
    if (kOp == 2) reach = eval_reach<kFast>(L, t, p);
    return eval_dist<kFast>(L, t, p);
}
template <bool kFast>
__device__ __forceinline__ bool eval_pair(const LrmCompiledLeg& L, const LdsTables* t, LrmVec3 tg, LrmVec3 body) {
    if (kFast) return lrm_reachable_rotate_leg_filtered(L, t->lists, t->lean, tg, body);
    return lrm_reachable_rotate_leg(L, t->lists, tg, body);
}

// ------------------------------------------------------------------------------------
// reachability_global_kernel (one_leg_global.cu:149-156), 4 consecutive points per lane:
// three 16-byte loads in, one 4-byte store out; in the filtered mode the (rare) doubtful points are
// re-evaluated by the strict code in the same launch, the bit words come from a 16-lane shuffle.
// (A two-launch variant -- filter only, then a fix-up pass over the mask -- was the faster one
// while the strict code in the loop cost SGPR spills and occupancy; since the leg is read through
// the kernarg pointer this single pass at 8 waves/SIMD takes 24.4 us at 1e7 points against 30.0.)
// ------------------------------------------------------------------------------------
template <bool kBits, bool kFast>
__global__ __launch_bounds__(kBlock, LRM_REACH_MIN_WAVES) void reach_soa_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ z, size_t n,
                                                           const LrmCompiledLeg L_kernarg,
                                                           uint8_t* __restrict__ mask,
                                                           uint64_t* __restrict__ bits) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgSoA);
    stage_lists(L, &s_tab);
    const size_t nquad = n >> 2;
    // every wave runs the same number of iterations so that the shuffles below see all lanes
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t nquad_pad = (nquad + 63) & ~(size_t)63;
    for (size_t qd = (size_t)blockIdx.x * kBlock + threadIdx.x; qd < nquad_pad; qd += stride) {
        uint32_t packed = 0;
        if (qd < nquad) {
            const float4 vx = reinterpret_cast<const float4*>(x)[qd];
            const float4 vy = reinterpret_cast<const float4*>(y)[qd];
            const float4 vz = reinterpret_cast<const float4*>(z)[qd];
            if (kFast) {
                // filter on all four points first; the (rare) strict re-evaluations share one copy
                // of the strict code, kept out of the hot path
                uint32_t u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                const bool r0 = lrm_reach_global_fast(L, s_tab.lean, LrmVec3{vx.x, vy.x, vz.x}, u0);
                const bool r1 = lrm_reach_global_fast(L, s_tab.lean, LrmVec3{vx.y, vy.y, vz.y}, u1);
                const bool r2 = lrm_reach_global_fast(L, s_tab.lean, LrmVec3{vx.z, vy.z, vz.z}, u2);
                const bool r3 = lrm_reach_global_fast(L, s_tab.lean, LrmVec3{vx.w, vy.w, vz.w}, u3);
                packed = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
                uint32_t redo = (u0 ? 1u : 0u) | (u1 ? 2u : 0u) | (u2 ? 4u : 0u) | (u3 ? 8u : 0u);
                if (__builtin_expect(redo != 0, 0)) {
#pragma unroll 1
                    for (int j = 0; j < 4; j++) {
                        if (!((redo >> j) & 1u)) continue;
                        const LrmVec3 p{j == 0 ? vx.x : j == 1 ? vx.y : j == 2 ? vx.z : vx.w,
                                        j == 0 ? vy.x : j == 1 ? vy.y : j == 2 ? vy.z : vy.w,
                                        j == 0 ? vz.x : j == 1 ? vz.y : j == 2 ? vz.z : vz.w};
                        const uint32_t r = lrm_reach_global(L, s_tab.lists, p) ? 1u : 0u;
                        packed = (packed & ~(0xffu << (8 * j))) | (r << (8 * j));
                    }
                }
            } else {
                const bool r0 = lrm_reach_global(L, s_tab.lists, LrmVec3{vx.x, vy.x, vz.x});
                const bool r1 = lrm_reach_global(L, s_tab.lists, LrmVec3{vx.y, vy.y, vz.y});
                const bool r2 = lrm_reach_global(L, s_tab.lists, LrmVec3{vx.z, vy.z, vz.z});
                const bool r3 = lrm_reach_global(L, s_tab.lists, LrmVec3{vx.w, vy.w, vz.w});
                packed = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
            }
            if (mask) reinterpret_cast<uint32_t*>(mask)[qd] = packed;
        }
        if (kBits) {
            // 16 lanes x 4 points = one 64-bit word
            // bit 0 of the four mask bytes gathered at bits 21..24 by one multiplication
            const uint32_t nib = (((packed & 0x01010101u) * 0x10204081u) >> 21) & 0xfu;
            // lanes 0-7 of a 16-lane group make the low half of the word, lanes 8-15 the high half:
            // three 32-bit exchanges inside the 8-lane groups, one across
            const int sub = threadIdx.x & 15;
            uint32_t h = nib << (4 * (sub & 7));
            h |= __shfl_xor(h, 1);
            h |= __shfl_xor(h, 2);
            h |= __shfl_xor(h, 4);
            const uint32_t hi = __shfl_down(h, 8);
            if (sub == 0 && (qd >> 4) < (n >> 6)) bits[qd >> 4] = (uint64_t)h | ((uint64_t)hi << 32); // full words only
        }
    }
    // tail: n % 4 points, handled by the first lanes of block 0
    const size_t tail0 = nquad << 2;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
        const size_t i = tail0 + threadIdx.x;
        const bool r = eval_reach<kFast>(L, &s_tab, LrmVec3{x[i], y[i], z[i]});
        if (mask) mask[i] = r;
    }
    if (kBits && blockIdx.x == 0 && threadIdx.x == 0 && (n & 63)) {
        // the last (partial) word: rebuild it from scratch so that unused bits are 0
        const size_t w0 = n & ~(size_t)63;
        uint64_t w = 0;
        for (size_t i = w0; i < n; i++)
            w |= (uint64_t)eval_reach<kFast>(L, &s_tab, LrmVec3{x[i], y[i], z[i]}) << (i - w0);
        bits[w0 >> 6] = w;
    }
}

// Same computation, one point per lane: used when the arrays are not 16-byte aligned (views
// into a larger allocation).  The wave ballot is the bit word.
template <bool kFast>
__global__ __launch_bounds__(kBlock) void reach_soa_scalar_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ z, size_t n,
                                                                  const LrmCompiledLeg L_kernarg,
                                                                  uint8_t* __restrict__ mask,
                                                                  uint64_t* __restrict__ bits) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgSoA);
    stage_lists(L, &s_tab);
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_pad = (n + 63) & ~(size_t)63;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
        bool m = false;
        if (i < n) {
            m = eval_reach<kFast>(L, &s_tab, LrmVec3{x[i], y[i], z[i]});
            if (mask) mask[i] = m;
        }
        if (bits) {
            const uint64_t w = __ballot(m);
            if ((threadIdx.x & 63) == 0) bits[i >> 6] = w;
        }
    }
}

// ------------------------------------------------------------------------------------
// distance_global_kernel (one_leg_global.cu:157-166) and the fused reach+distance kernel:
// one point per lane per iteration (the distance code is long: keep one copy of it).
// kOp: 1 = distance (+ optional validity byte), 2 = reach mask + distance.
// ------------------------------------------------------------------------------------
template <int kOp, bool kFast>
__global__ __launch_bounds__(kBlock, kFast ? LRM_DIST_MIN_WAVES : LRM_DIST_STRICT_MIN_WAVES) void dist_soa_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ z, size_t n,
                                                          const LrmCompiledLeg L_kernarg,
                                                          uint8_t* __restrict__ mask,
                                                          uint64_t* __restrict__ bits,
                                                          float* __restrict__ dx,
                                                          float* __restrict__ dy,
                                                          float* __restrict__ dz) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgSoA);
    stage_lists(L, &s_tab);
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_pad = (n + 63) & ~(size_t)63; // whole waves iterate together (ballot below)
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
        bool m = false;
        if (i < n) {
            LrmVec3 p{x[i], y[i], z[i]};
            bool v;
            if (kOp == 2 && kFast) {
                // as lrm_reach_dist_global_filtered, but the (rare) strict re-evaluation of the mask re-loads the
                // point instead of keeping it live across the whole distance evaluation (3 VGPRs: the difference
                // between 72 registers with one spilled and 72 without)
                uint32_t stat = 0;
                LrmDistByproduct by;
                v = lrm_dist_global_fast(L, LrmDistTables{s_tab.lists, s_tab.dist, s_tab.corners}, p, stat, &by);
                bool doubt;
                m = lrm_reach_from_dist(L, by, doubt);
                if (doubt) {
                    size_t ii = i;
                    asm volatile("" : "+v"(ii)); // opaque: a real re-load, or the compiler keeps the point (and |x|, ...) live for this path
                    m = lrm_reach_global(L, s_tab.lists, LrmVec3{x[ii], y[ii], z[ii]});
                }
            } else {
                v = eval_reach_dist<kOp, kFast>(L, &s_tab, p, m);
            }
            dx[i] = p.x;
            dy[i] = p.y;
            dz[i] = p.z;
            if (kOp != 2) m = v;
            if (mask) mask[i] = m;
        }
        if (bits) { // wave64 ballot: lane l of this wave holds point (i & ~63) + l
            const uint64_t w = __ballot(m);
            if ((threadIdx.x & 63) == 0) bits[i >> 6] = w;
        }
    }
}

// AoS variants for the apply_kernel boundary (cross_compiled.cu:33-79)
template <bool kFast>
__global__ __launch_bounds__(kBlock) void reach_aos_kernel(const float* __restrict__ xyz, size_t n,
                                                           const LrmCompiledLeg L_kernarg,
                                                           uint8_t* __restrict__ mask) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgAoS);
    stage_lists(L, &s_tab);
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        mask[i] = eval_reach<kFast>(L, &s_tab, p);
    }
}

// The same on 4 consecutive points per lane (16-byte aligned arrays): three 16-byte loads = four float3, one 4-byte store,
// the strict re-evaluations of the rare doubtful points out of the hot path -- the structure of reach_soa_kernel.
// (One point per lane: three strided 4-byte loads and a byte store, 36 us per 1e7 points against 24 for the SoA kernel.)
template <bool kFast>
__global__ __launch_bounds__(kBlock, LRM_REACH_MIN_WAVES) void reach_aos4_kernel(const float* __restrict__ xyz, size_t n,
                                                                                 const LrmCompiledLeg L_kernarg,
                                                                                 uint8_t* __restrict__ mask) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgAoS);
    stage_lists(L, &s_tab);
    const size_t nquad = n >> 2;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t qd = (size_t)blockIdx.x * kBlock + threadIdx.x; qd < nquad; qd += stride) {
        const float4 a = reinterpret_cast<const float4*>(xyz)[3 * qd];
        const float4 b = reinterpret_cast<const float4*>(xyz)[3 * qd + 1];
        const float4 c = reinterpret_cast<const float4*>(xyz)[3 * qd + 2];
        const LrmVec3 p0{a.x, a.y, a.z}, p1{a.w, b.x, b.y}, p2{b.z, b.w, c.x}, p3{c.y, c.z, c.w};
        uint32_t packed;
        if (kFast) {
            uint32_t u0 = 0, u1 = 0, u2 = 0, u3 = 0;
            const bool r0 = lrm_reach_global_fast(L, s_tab.lean, p0, u0);
            const bool r1 = lrm_reach_global_fast(L, s_tab.lean, p1, u1);
            const bool r2 = lrm_reach_global_fast(L, s_tab.lean, p2, u2);
            const bool r3 = lrm_reach_global_fast(L, s_tab.lean, p3, u3);
            packed = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
            const uint32_t redo = (u0 ? 1u : 0u) | (u1 ? 2u : 0u) | (u2 ? 4u : 0u) | (u3 ? 8u : 0u);
            if (__builtin_expect(redo != 0, 0)) {
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    if (!((redo >> j) & 1u)) continue;
                    const LrmVec3 p = j == 0 ? p0 : j == 1 ? p1 : j == 2 ? p2 : p3;
                    const uint32_t r = lrm_reach_global(L, s_tab.lists, p) ? 1u : 0u;
                    packed = (packed & ~(0xffu << (8 * j))) | (r << (8 * j));
                }
            }
        } else {
            packed = (uint32_t)lrm_reach_global(L, s_tab.lists, p0) | ((uint32_t)lrm_reach_global(L, s_tab.lists, p1) << 8) |
                     ((uint32_t)lrm_reach_global(L, s_tab.lists, p2) << 16) | ((uint32_t)lrm_reach_global(L, s_tab.lists, p3) << 24);
        }
        reinterpret_cast<uint32_t*>(mask)[qd] = packed;
    }
    const size_t tail0 = nquad << 2; // n % 4 points
    if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
        const size_t i = tail0 + threadIdx.x;
        mask[i] = eval_reach<kFast>(L, &s_tab, LrmVec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    }
}

template <int kOp, bool kFast>
__global__ __launch_bounds__(kBlock, kFast ? LRM_DIST_MIN_WAVES : LRM_DIST_STRICT_MIN_WAVES) void dist_aos_kernel(const float* __restrict__ xyz, size_t n,
                                                          const LrmCompiledLeg L_kernarg,
                                                          uint8_t* __restrict__ mask,
                                                          float* __restrict__ dxyz) {
    __shared__ LdsTables s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kLegArgAoS);
    stage_lists(L, &s_tab);
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        LrmVec3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        bool m = false;
        bool v;
        if (kOp == 2 && kFast) { // as dist_soa_kernel: the rare strict re-evaluation of the mask re-loads its point (16 B per lane of scratch otherwise)
            uint32_t stat = 0;
            LrmDistByproduct by;
            v = lrm_dist_global_fast(L, LrmDistTables{s_tab.lists, s_tab.dist, s_tab.corners}, p, stat, &by);
            bool doubt;
            m = lrm_reach_from_dist(L, by, doubt);
            if (doubt) {
                size_t ii = i;
                asm volatile("" : "+v"(ii));
                m = lrm_reach_global(L, s_tab.lists, LrmVec3{xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2]});
            }
        } else {
            v = eval_reach_dist<kOp, kFast>(L, &s_tab, p, m);
        }
        dxyz[3 * i] = p.x;
        dxyz[3 * i + 1] = p.y;
        dxyz[3 * i + 2] = p.z;
        if (mask) mask[i] = (kOp == 2) ? m : v;
    }
}

// empty_kernel, cuda_util.cu:3 (the warm-up launch of apply_kernel, cross_compiled.cu:52)
__global__ void warmup_kernel() {}

// ------------------------------------------------------------------------------------
// Body x target aggregation: reach_mem_kernel (several_leg.cu:92-129) for all legs and all
// targets in ONE launch; the AND over legs (agregateReachability, several_leg.cu:681-697, any
// number of legs) is written by the same kernel.
//   block = 4 waves; each wave owns one body at a time and ALL legs of it; the four waves walk
//   the same 1024-target LDS tile.
//   Stage 1 (every target): lane = target, conservative sphere test |t - b|^2 <= reach^2 (pairs
//   outside cannot be reachable: skipping them changes no output), __ballot + mbcnt prefix
//   append the survivors to the wave's LDS queue.
//   Stage 2 (full 64-lane batches of survivors): for every leg that has not found a target yet,
//   lane = queued target -> filtered reachable_rotate_leg -> __ballot is the "any".
//   A body whose legs have all found a target stops consuming work.
// The reference launches Nt/512 kernels per leg and re-tests every pair (several_leg.cu:131-157).
// ------------------------------------------------------------------------------------
constexpr int kWaves = kBlock / 64;
constexpr int kTargetTile = 1024;
constexpr int kQueue = 128;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two levels of axis-aligned bounding boxes over the target cloud in memory order: one per
// 1024-target tile (boxes[t*6..]) and one per 64-target chunk (boxes[(ntiles + c)*6..]); one block
// per tile.  reach_any_kernel / any_in_shape_kernel skip, per group of bodies, the tiles and then
// the chunks that are out of reach.  Pays off when the cloud has spatial locality in memory order
// (Morton-sorted clouds: lrm_morton_order; terrain rasters); costs one cheap pass otherwise.
__global__ __launch_bounds__(kBlock) void tile_aabb_kernel(const float* __restrict__ tx, const float* __restrict__ ty,
                                                           const float* __restrict__ tz, size_t nt, size_t ntiles,
                                                           float* __restrict__ boxes) {
    __shared__ float s_red[6][16];
    const size_t t0 = (size_t)blockIdx.x * 1024;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int sub = wave; sub < 16; sub += kBlock / 64) { // each wave: four 64-target chunks
        const size_t i = t0 + (size_t)sub * 64 + lane;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        if (i < nt) {
            lo[0] = hi[0] = tx[i];
            lo[1] = hi[1] = ty[i];
            lo[2] = hi[2] = tz[i];
        }
#pragma unroll
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
                boxes[(ntiles + c) * 6 + a] = lo[a];     // an empty chunk keeps (+big, -big): never "near"
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

// squared distance from a point to a box (lower bound of the distance to every member)
__device__ __forceinline__ float box_dist2(const float* bb, float x, float y, float z) {
    const float ex = fmaxf(fmaxf(bb[0] - x, x - bb[3]), 0.f);
    const float ey = fmaxf(fmaxf(bb[1] - y, y - bb[4]), 0.f);
    const float ez = fmaxf(fmaxf(bb[2] - z, z - bb[5]), 0.f);
    return ex * ex + ey * ey + ez * ez;
}

#ifndef LRM_ANY_MIN_WAVES
#define LRM_ANY_MIN_WAVES 6 // barrier-heavy (4 bodies share each staged tile): occupancy hides the waits; 4 -> 6 waves/SIMD: -17 %, 7: no further gain
#endif
template <bool kFast>
__global__ __launch_bounds__(kBlock, LRM_ANY_MIN_WAVES) void reach_any_kernel(
    const float* __restrict__ bx, const float* __restrict__ by, const float* __restrict__ bz, size_t nb,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz, size_t nt,
    const LrmCompiledLeg* __restrict__ legs, int nlegs, const float* __restrict__ boxes,
    const uint8_t* __restrict__ body_active /* null = all */, uint8_t* __restrict__ out, uint8_t* __restrict__ all_out) {
    __shared__ float s_tx[kTargetTile], s_ty[kTargetTile], s_tz[kTargetTile];
    __shared__ float s_qx[kWaves][kQueue], s_qy[kWaves][kQueue], s_qz[kWaves][kQueue];
    __shared__ LrmCompiledLeg::LeanCircle s_lean[LRM_MAX_LEGS][16];
    __shared__ int s_todo;
    __shared__ unsigned s_near[2];

    for (int i = threadIdx.x; i < nlegs * 64; i += kBlock)
        reinterpret_cast<float*>(&s_lean[i >> 6][0])[i & 63] = reinterpret_cast<const float*>(&legs[i >> 6].lean[0][0])[i & 63];
    float r2max = 0.f;
    for (int l = 0; l < nlegs; l++) r2max = fmaxf(r2max, legs[l].reach_r2_max);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t all_found = (1u << nlegs) - 1u;
    float* qx = s_qx[wave];
    float* qy = s_qy[wave];
    float* qz = s_qz[wave];

    for (size_t group = blockIdx.x; group * kWaves < nb; group += gridDim.x) {
        const size_t b = group * kWaves + wave;
        const bool in_range = b < nb;
        const bool live = in_range && (!body_active || body_active[b] != 0); // inactive bodies answer 0
        const LrmVec3 body = live ? LrmVec3{bx[b], by[b], bz[b]} : LrmVec3{0.f, 0.f, 0.f};
        uint32_t found = live ? 0u : all_found; // bit l: leg l has a reachable target
        int count = 0;                           // survivors waiting in this wave's queue

        // stage 2: the first `m` queue entries against every leg still searching
        auto process = [&](int m) {
            LrmVec3 t{0.f, 0.f, 0.f};
            if (lane < m) t = LrmVec3{qx[lane], qy[lane], qz[lane]};
            for (int l = 0; l < nlegs; l++) {
                if ((found >> l) & 1u) continue; // wave-uniform
                bool hit = false;
                if (lane < m) {
                    if (kFast) hit = lrm_reachable_rotate_leg_filtered(legs[l], &legs[l].lists[0][0], s_lean[l], t, body);
                    else hit = lrm_reachable_rotate_leg(legs[l], &legs[l].lists[0][0], t, body);
                }
                if (__ballot(hit) != 0ull) found |= 1u << l;
            }
        };

        const size_t ntiles = (nt + kTargetTile - 1) / kTargetTile;
        for (size_t tw0 = 0; tw0 < ntiles; tw0 += 64) {
          // which of the next 64 tiles can any of the four bodies reach?  (lane = tile)
          unsigned long long near = ~0ull;
          if (boxes) {
              __syncthreads();
              if (threadIdx.x < 2) s_near[threadIdx.x] = 0u;
              __syncthreads();
              bool mine = false;
              const size_t tl = tw0 + lane;
              if (tl < ntiles && found != all_found) {
                  // box distance is a lower bound of every member's distance; 1e-3 relative slack
                  // for the rounding of the bound itself
                  mine = box_dist2(boxes + tl * 6, body.x, body.y, body.z) * 0.999f <= r2max;
              }
              const unsigned long long mm = __ballot(mine);
              if (lane == 0 && mm) {
                  atomicOr(&s_near[0], (unsigned)mm);
                  atomicOr(&s_near[1], (unsigned)(mm >> 32));
              }
              __syncthreads();
              near = (unsigned long long)s_near[0] | ((unsigned long long)s_near[1] << 32);
          }
          bool all_done = false;
          for (int tb = 0; tb < 64 && tw0 + tb < ntiles; tb++) {
            if (!((near >> tb) & 1ull)) continue; // block-uniform
            const size_t t0 = (tw0 + tb) * kTargetTile;
            const int tile_n = (int)((nt - t0 < (size_t)kTargetTile) ? (nt - t0) : (size_t)kTargetTile);
            __syncthreads(); // previous tile fully consumed (and s_todo read)
            if (threadIdx.x == 0) s_todo = 0;
            for (int i = threadIdx.x; i < tile_n; i += kBlock) {
                s_tx[i] = tx[t0 + i];
                s_ty[i] = ty[t0 + i];
                s_tz[i] = tz[t0 + i];
            }
            __syncthreads();
            if (found != all_found) { // wave-uniform
                for (int s = 0; s < tile_n; s += 64) {
                    // (the chunk-level boxes are not used here: against a 0.5 m reach sphere they skip too
                    // little to pay for their test; the sphere / cylinder reductions do use them)
                    const int i = s + lane;
                    bool keep = false;
                    LrmVec3 t{0.f, 0.f, 0.f};
                    if (i < tile_n) {
                        t = LrmVec3{s_tx[i], s_ty[i], s_tz[i]};
                        const float ddx = t.x - body.x, ddy = t.y - body.y, ddz = t.z - body.z;
                        keep = __builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx)) <= r2max;
                    }
                    const unsigned long long m = __ballot(keep);
                    if (m == 0ull) continue;
                    if (keep) {
                        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        qx[pos] = t.x;
                        qy[pos] = t.y;
                        qz[pos] = t.z;
                    }
                    count += __builtin_popcountll(m);
                    wave_lds_fence();
                    if (count >= 64) {
                        process(64);
                        count -= 64;
                        // the (< 64) entries behind the processed batch move to the front
                        float mx = 0.f, my = 0.f, mz = 0.f;
                        if (lane < count) { mx = qx[64 + lane]; my = qy[64 + lane]; mz = qz[64 + lane]; }
                        wave_lds_fence();
                        if (lane < count) { qx[lane] = mx; qy[lane] = my; qz[lane] = mz; }
                        wave_lds_fence();
                        if (found == all_found) break;
                    }
                }
            }
            if (lane == 0 && found != all_found) atomicAdd(&s_todo, 1);
            __syncthreads();
            if (s_todo == 0) { all_done = true; break; } // block-uniform: all four bodies are done
          }
          if (all_done) break;
        }
        if (count > 0 && found != all_found) process(count);
        if (in_range && lane < nlegs) out[(size_t)lane * nb + b] = live ? ((found >> lane) & 1u) : 0;
        if (in_range && lane == 0 && all_out) all_out[b] = (live && found == all_found) ? 1 : 0;
    }
}

// The same reduction with fully independent waves (used when the two-level boxes are available):
// a wave owns one body, walks the tile boxes (lane = tile) and, inside a near tile, the chunk boxes
// (lane = chunk), and reads only the near 64-target chunks straight from global memory (the cloud
// is L2-resident: 1.2 MB for 1e5 footholds) with the next chunk's loads in flight while the
// current one is tested.  No tile staging in LDS, no block barriers (reach_any_kernel waits on
// them half of the time: four bodies advance in lockstep through every tile any of them needs),
// and a finished body leaves at once.
// -DLRM_PAIR_COUNT: a counting build (tools/c3_evidence.py; never the shipped library): how many (leg, target) pairs
// the wave-per-body kernel really evaluates, against the nb * nt * nlegs pairs it answers.
//   [0] full evaluations (lrm_reachable_rotate_leg*), [1] leg bounding-sphere tests, [2] footholds inside a body's
//   reach sphere (queued), [3] footholds loaded
#if defined(LRM_PAIR_COUNT)
__device__ unsigned long long g_pair_count[4];
#define LRM_COUNT_PAIRS(k, ballot_mask) do { const unsigned long long bm_ = (ballot_mask); /* the ballot with every lane active */ \
        if (lane == 0) atomicAdd(&g_pair_count[k], (unsigned long long)__builtin_popcountll(bm_)); } while (0)
#else
#define LRM_COUNT_PAIRS(k, ballot_mask) do {} while (0)
#endif
#ifndef LRM_ANY_WAVE_MIN_WAVES
#define LRM_ANY_WAVE_MIN_WAVES 8 // latency-bound on its L2 loads: 4 / 5 / 6 / 8 waves per SIMD -> 1.10 / 1.10 / 1.03 / 0.98 ms
#endif
template <bool kFast>
__global__ __launch_bounds__(kBlock, LRM_ANY_WAVE_MIN_WAVES) void reach_any_wave_kernel(
    const float* __restrict__ bx, const float* __restrict__ by, const float* __restrict__ bz, size_t nb,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz, size_t nt,
    const LrmCompiledLeg* __restrict__ legs, int nlegs, const float* __restrict__ boxes,
    const uint8_t* __restrict__ body_active /* null = all */, uint8_t* __restrict__ out, uint8_t* __restrict__ all_out) {
    __shared__ float s_qx[kWaves][kQueue], s_qy[kWaves][kQueue], s_qz[kWaves][kQueue];
    __shared__ LrmCompiledLeg::LeanCircle s_lean[LRM_MAX_LEGS][16];
    __shared__ float s_sphere[LRM_MAX_LEGS][4]; // per-leg bounding sphere: centre (relative to the body), r^2
    for (int i = threadIdx.x; i < nlegs * 64; i += kBlock)
        reinterpret_cast<float*>(&s_lean[i >> 6][0])[i & 63] = reinterpret_cast<const float*>(&legs[i >> 6].lean[0][0])[i & 63];
    if (threadIdx.x < nlegs * 4)
        s_sphere[threadIdx.x >> 2][threadIdx.x & 3] =
            (threadIdx.x & 3) < 3 ? legs[threadIdx.x >> 2].pair_center[threadIdx.x & 3] : legs[threadIdx.x >> 2].pair_r2;
    __syncthreads(); // the only one
    float r2max = 0.f;
    for (int l = 0; l < nlegs; l++) r2max = fmaxf(r2max, legs[l].reach_r2_max);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t all_found = (1u << nlegs) - 1u;
    float* qx = s_qx[wave];
    float* qy = s_qy[wave];
    float* qz = s_qz[wave];
    const size_t ntiles = (nt + kTargetTile - 1) / kTargetTile;

    for (size_t b = (size_t)blockIdx.x * kWaves + wave; b < nb; b += (size_t)gridDim.x * kWaves) {
        const bool live = !body_active || body_active[b] != 0; // inactive bodies answer 0
        const LrmVec3 body{bx[b], by[b], bz[b]};
        uint32_t found = live ? 0u : all_found;
        int count = 0;

        auto process = [&](int m) {
            LrmVec3 t{0.f, 0.f, 0.f};
            if (lane < m) t = LrmVec3{qx[lane], qy[lane], qz[lane]};
            const float rx = t.x - body.x, ry = t.y - body.y, rz = t.z - body.z;
            for (int l = 0; l < nlegs; l++) {
                if ((found >> l) & 1u) continue; // wave-uniform
                // the leg's own bounding sphere: a batch is a patch of neighbouring footholds and
                // mostly lies outside the spheres of all but two or three legs
                const float ex = rx - legs[l].pair_center[0], ey = ry - legs[l].pair_center[1], ez = rz - legs[l].pair_center[2];
                const bool inside = (lane < m) && __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) <= legs[l].pair_r2;
                LRM_COUNT_PAIRS(1, __ballot(lane < m));
                LRM_COUNT_PAIRS(0, __ballot(inside));
                if (__ballot(inside) == 0ull) continue;
                bool hit = false;
                if (inside) {
                    if (kFast) hit = lrm_reachable_rotate_leg_filtered(legs[l], &legs[l].lists[0][0], s_lean[l], t, body);
                    else hit = lrm_reachable_rotate_leg(legs[l], &legs[l].lists[0][0], t, body);
                }
                if (__ballot(hit) != 0ull) found |= 1u << l;
            }
        };

        for (size_t tw0 = 0; tw0 < ntiles && found != all_found; tw0 += 64) {
            // lane = tile: box distance is a lower bound of every member's distance; 1e-3 relative
            // slack for the rounding of the bound itself
            const size_t tl = tw0 + lane;
            unsigned long long near =
                __ballot(tl < ntiles && box_dist2(boxes + tl * 6, body.x, body.y, body.z) * 0.999f <= r2max);
            while (near != 0ull && found != all_found) {
                const int tb = __builtin_ctzll(near);
                near &= near - 1ull;
                const size_t tile = tw0 + tb;
                const size_t t0 = tile * kTargetTile;
                // lane = (chunk of this tile, one of four legs): a chunk is read when its box touches the
                // bounding sphere of a leg that is still searching (empty chunks carry an inverted box)
                uint32_t cnear = 0u;
                {
                    const float* cb = boxes + (ntiles + tile * 16 + (lane & 15)) * 6;
                    for (int l0 = 0; l0 < nlegs; l0 += 4) { // wave-uniform
                        const int l = l0 + (lane >> 4);
                        bool touch = false;
                        if (l < nlegs && !((found >> l) & 1u))
                            touch = box_dist2(cb, body.x + s_sphere[l][0], body.y + s_sphere[l][1], body.z + s_sphere[l][2]) * 0.999f <=
                                    s_sphere[l][3];
                        const unsigned long long mm = __ballot(touch);
                        cnear |= (uint32_t)((mm | (mm >> 16) | (mm >> 32) | (mm >> 48)) & 0xffffull);
                    }
                }
                // software pipeline: the next near chunk's loads are issued before this one is tested
                LrmVec3 nxt{0.f, 0.f, 0.f};
                bool nxt_ok = false;
                auto fetch = [&](int chunk) {
                    const size_t i = t0 + (size_t)chunk * 64 + lane;
                    nxt_ok = i < nt;
                    if (nxt_ok) nxt = LrmVec3{tx[i], ty[i], tz[i]};
                };
                if (cnear) {
                    fetch(__builtin_ctz(cnear));
                    cnear &= cnear - 1u;
                }
                bool more = true;
                while (more) {
                    const LrmVec3 t = nxt;
                    const bool ok = nxt_ok;
                    more = cnear != 0u;
                    if (more) {
                        fetch(__builtin_ctz(cnear));
                        cnear &= cnear - 1u;
                    }
                    const float ddx = t.x - body.x, ddy = t.y - body.y, ddz = t.z - body.z;
                    const bool keep = ok && __builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx)) <= r2max;
                    const unsigned long long m = __ballot(keep);
                    LRM_COUNT_PAIRS(3, __ballot(ok));
                    LRM_COUNT_PAIRS(2, m);
                    if (m == 0ull) continue;
                    if (keep) {
                        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        qx[pos] = t.x;
                        qy[pos] = t.y;
                        qz[pos] = t.z;
                    }
                    count += __builtin_popcountll(m);
                    wave_lds_fence();
                    if (count >= 64) {
                        process(64);
                        count -= 64;
                        float mx = 0.f, my = 0.f, mz = 0.f;
                        if (lane < count) { mx = qx[64 + lane]; my = qy[64 + lane]; mz = qz[64 + lane]; }
                        wave_lds_fence();
                        if (lane < count) { qx[lane] = mx; qy[lane] = my; qz[lane] = mz; }
                        wave_lds_fence();
                        if (found == all_found) break;
                    }
                }
            }
        }
        if (count > 0 && found != all_found) process(count);
        if (lane < nlegs) out[(size_t)lane * nb + b] = live ? ((found >> lane) & 1u) : 0;
        if (lane == 0 && all_out) all_out[b] = (live && found == all_found) ? 1 : 0;
    }
}

// rotateData (several_leg.cu:401-411): dst = qtRotate(q, src) with the strict coefficient form
__global__ __launch_bounds__(kBlock) void rotate_soa_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                            const float* __restrict__ sz, size_t n, const LrmCompiledLeg* rot,
                                                            float* __restrict__ dx, float* __restrict__ dy,
                                                            float* __restrict__ dz) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const LrmVec3 r = lrm_qrot(rot->fwd_rot, LrmVec3{sx[i], sy[i], sz[i]});
        dx[i] = r.x;
        dy[i] = r.y;
        dz[i] = r.z;
    }
}

// One orientation of the estimator's pipeline folded into the running state (several_leg.cu:
// 504-559 bring_together, :698-706 cleanAgregated, :396-399 flipWorkingSide): a body is accepted
// when it is still active, passes the cylinder pair (if used) and every leg found a target.
__global__ __launch_bounds__(kBlock) void sweep_update_kernel(const uint8_t* __restrict__ all_legs,
                                                              const uint8_t* __restrict__ cyl_validate,
                                                              const uint8_t* __restrict__ cyl_eliminate, int use_culls,
                                                              size_t nb, uint8_t* __restrict__ active,
                                                              uint8_t* __restrict__ accepted) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t b = (size_t)blockIdx.x * kBlock + threadIdx.x; b < nb; b += stride) {
        bool ok = active[b] != 0 && all_legs[b] != 0;
        if (use_culls) ok = ok && cyl_validate[b] == 1 && cyl_eliminate[b] != 1;
        if (ok) {
            accepted[b] = 1;
            active[b] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------
// in_sphere_mem_kernel / in_cylinder_mem_kernel (collision.cu:40-66, :119-146) in one
// launch each: out[c] = any target inside the sphere / cylinder centred on c.
// kShape: 0 sphere (collision.cu.h:5-10), 1 cylinder (collision.cu.h:12-23).
// ------------------------------------------------------------------------------------
template <int kShape>
__global__ __launch_bounds__(kBlock) void any_in_shape_kernel(
    const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz, size_t nc,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz, size_t nt,
    float radius, float plus_z, float minus_z, const float* __restrict__ boxes, uint8_t* __restrict__ out) {
    __shared__ float s_tx[kTargetTile], s_ty[kTargetTile], s_tz[kTargetTile];
    __shared__ float s_red[6][kBlock / 64];
    __shared__ float s_cbox[6];
    __shared__ float s_chunk_box[16 * 6];
    // one centre per thread, the whole block sweeps the same LDS tile (broadcast reads)
    const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = c < nc;
    const float px = live ? cx[c] : 0.f, py = live ? cy[c] : 0.f, pz = live ? cz[c] : 0.f;
    if (boxes) {
        // bounding box of this block's centres: a target tile whose box is out of reach of that
        // box cannot contain a hit for any of them (block-uniform skip)
        float lo[3] = {live ? px : 3.0e38f, live ? py : 3.0e38f, live ? pz : 3.0e38f};
        float hi[3] = {live ? px : -3.0e38f, live ? py : -3.0e38f, live ? pz : -3.0e38f};
#pragma unroll
        for (int a = 0; a < 3; a++)
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
            }
        if ((threadIdx.x & 63) == 0)
            for (int a = 0; a < 3; a++) {
                s_red[a][threadIdx.x >> 6] = lo[a];
                s_red[3 + a][threadIdx.x >> 6] = hi[a];
            }
        __syncthreads();
        if (threadIdx.x < 6) {
            float v = s_red[threadIdx.x][0];
            for (int w = 1; w < kBlock / 64; w++) v = (threadIdx.x < 3) ? fminf(v, s_red[threadIdx.x][w]) : fmaxf(v, s_red[threadIdx.x][w]);
            s_cbox[threadIdx.x] = v;
        }
        __syncthreads();
    }
    bool found = false;
    const size_t ntiles = (nt + kTargetTile - 1) / kTargetTile;
    for (size_t t0 = 0; t0 < nt; t0 += kTargetTile) {
        if (boxes) { // block-uniform
            const float* bb = boxes + (t0 / kTargetTile) * 6;
            const float gx = fmaxf(fmaxf(bb[0] - s_cbox[3], s_cbox[0] - bb[3]), 0.f); // gaps between the two boxes
            const float gy = fmaxf(fmaxf(bb[1] - s_cbox[4], s_cbox[1] - bb[4]), 0.f);
            const float gz = fmaxf(fmaxf(bb[2] - s_cbox[5], s_cbox[2] - bb[5]), 0.f);
            const float slack = 0.999f; // the gaps are lower bounds; keep a margin for their rounding
            bool skip;
            if (kShape == 0) skip = (gx * gx + gy * gy + gz * gz) * slack >= radius * radius;
            else // cylinder: radial gap, or the z windows (target.z - centre.z in (minus_z, plus_z)) cannot overlap
                skip = (gx * gx + gy * gy) * slack >= radius * radius || (bb[2] - s_cbox[5]) >= plus_z + 1e-3f * fabsf(plus_z) + 1e-3f ||
                       (bb[5] - s_cbox[2]) <= minus_z - 1e-3f * fabsf(minus_z) - 1e-3f;
            if (skip) continue;
        }
        const int tile_n = (int)((nt - t0 < (size_t)kTargetTile) ? (nt - t0) : (size_t)kTargetTile);
        __syncthreads();
        for (int i = threadIdx.x; i < tile_n; i += kBlock) {
            s_tx[i] = tx[t0 + i];
            s_ty[i] = ty[t0 + i];
            s_tz[i] = tz[t0 + i];
        }
        if (boxes && threadIdx.x < 96) s_chunk_box[threadIdx.x] = boxes[(ntiles + t0 / 64) * 6 + threadIdx.x];
        __syncthreads();
        if (live && !found) {
            for (int i = 0; i < tile_n; i++) {
                if (boxes && (i & 63) == 0) { // second level: skip a 64-target chunk out of this centre's reach
                    const float* cb = s_chunk_box + (i >> 6) * 6;
                    const float gx = fmaxf(fmaxf(cb[0] - px, px - cb[3]), 0.f), gy = fmaxf(fmaxf(cb[1] - py, py - cb[4]), 0.f);
                    bool far;
                    if (kShape == 0) {
                        const float gz = fmaxf(fmaxf(cb[2] - pz, pz - cb[5]), 0.f);
                        far = (gx * gx + gy * gy + gz * gz) * 0.999f >= radius * radius;
                    } else {
                        far = (gx * gx + gy * gy) * 0.999f >= radius * radius || (cb[2] - pz) >= plus_z + 1e-3f * fabsf(plus_z) + 1e-3f ||
                              (cb[5] - pz) <= minus_z - 1e-3f * fabsf(minus_z) - 1e-3f;
                    }
                    if (far) { i += 63; continue; }
                }
                bool in;
                if (kShape == 0) {
                    const float ax = px - s_tx[i], ay = py - s_ty[i], az = pz - s_tz[i];
                    in = sqrtf(ax * ax + ay * ay + az * az) < radius;
                } else {
                    const float dzz = s_tz[i] - pz;
                    const float ax = s_tx[i] - px, ay = s_ty[i] - py;
                    in = (sqrtf(ax * ax + ay * ay + 0.f) < radius) && (dzz < plus_z) && (dzz > minus_z);
                }
                if (in) { found = true; break; }
            }
        }
    }
    if (live) out[c] = found ? 1 : 0;
}

// diagnostic: the device build of lrm_exact_math.h on arrays (tests compare it with glibc)
__global__ __launch_bounds__(kBlock) void exact_math_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            size_t n, float* __restrict__ at2,
                                                            float* __restrict__ sn, float* __restrict__ cs) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        at2[i] = lrm_atan2f(a[i], b[i]);
        float s, c;
        lrm_sincosf(a[i], &s, &c);
        sn[i] = s;
        cs[i] = c;
    }
}

// diagnostic: lrm_sqrtf against the compiler's IEEE sqrtf on every float bit pattern
__global__ __launch_bounds__(kBlock) void sqrt_check_kernel(unsigned long long* __restrict__ counters) {
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t first = blockIdx.x * kBlock + threadIdx.x;
    unsigned long long bad = 0, first_bad = ~0ull;
    // 2^32 patterns: every thread walks its residue class (stride divides 2^32)
    for (uint64_t u = first; u < (1ull << 32); u += stride) {
        const float x = lrm_u2f((uint32_t)u);
        const uint32_t got = lrm_f2u(lrm_sqrtf(x)), want = lrm_f2u(sqrtf(x));
        if (got != want) {
            bad++;
            if (u + 1 < first_bad) first_bad = u + 1;
        }
    }
    if (bad) {
        atomicAdd(&counters[0], bad);
        atomicMin(&counters[1], first_bad);
    }
}

inline int grid_for(size_t work_items, size_t cap = 256 * 8) {
    // memory-streaming launches: enough workgroups to fill 256 CUs x 8, grid-stride the rest
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

// The same reduction with one centre per WAVE (used when the two-level boxes are available): lane =
// tile box test, lane = chunk box test, then lane = target over the near chunks only, read from
// global memory (L2) with the next chunk in flight; the first hit ends the centre.  The
// thread-per-centre kernel above walks its near chunks serially in every lane and diverges; on the
// estimator's cylinder culls (1e5 centres x 1e5 footholds) it took 650 us, this one takes 63.
template <int kShape>
__global__ __launch_bounds__(kBlock) void any_in_shape_wave_kernel(
    const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz, size_t nc,
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz, size_t nt,
    float radius, float plus_z, float minus_z, const float* __restrict__ boxes, uint8_t* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t ntiles = (nt + kTargetTile - 1) / kTargetTile;
    const float r2 = radius * radius;
    const float zhi = plus_z + 1e-3f * fabsf(plus_z) + 1e-3f, zlo = minus_z - 1e-3f * fabsf(minus_z) - 1e-3f;
    // is the box out of reach of the centre?  (the gaps are lower bounds; 1e-3 margin for their rounding)
    auto far = [&](const float* bb, float px, float py, float pz) {
        const float gx = fmaxf(fmaxf(bb[0] - px, px - bb[3]), 0.f), gy = fmaxf(fmaxf(bb[1] - py, py - bb[4]), 0.f);
        if (kShape == 0) {
            const float gz = fmaxf(fmaxf(bb[2] - pz, pz - bb[5]), 0.f);
            return (gx * gx + gy * gy + gz * gz) * 0.999f >= r2;
        }
        return (gx * gx + gy * gy) * 0.999f >= r2 || (bb[2] - pz) >= zhi || (bb[5] - pz) <= zlo;
    };
    for (size_t c = (size_t)blockIdx.x * kWaves + wave; c < nc; c += (size_t)gridDim.x * kWaves) {
        const float px = cx[c], py = cy[c], pz = cz[c];
        bool found = false;
        for (size_t tw0 = 0; tw0 < ntiles && !found; tw0 += 64) {
            const size_t tl = tw0 + lane;
            unsigned long long near = __ballot(tl < ntiles && !far(boxes + tl * 6, px, py, pz));
            while (near != 0ull && !found) {
                const size_t tile = tw0 + __builtin_ctzll(near);
                near &= near - 1ull;
                const size_t t0 = tile * kTargetTile;
                uint32_t cnear = (uint32_t)__ballot(lane < 16 && !far(boxes + (ntiles + tile * 16 + lane) * 6, px, py, pz));
                float nx = 0.f, ny = 0.f, nz = 0.f;
                bool nxt_ok = false;
                auto fetch = [&](int chunk) {
                    const size_t i = t0 + (size_t)chunk * 64 + lane;
                    nxt_ok = i < nt;
                    if (nxt_ok) { nx = tx[i]; ny = ty[i]; nz = tz[i]; }
                };
                if (cnear) {
                    fetch(__builtin_ctz(cnear));
                    cnear &= cnear - 1u;
                }
                bool more = true;
                while (more) {
                    const float ux = nx, uy = ny, uz = nz;
                    const bool ok = nxt_ok;
                    more = cnear != 0u;
                    if (more) {
                        fetch(__builtin_ctz(cnear));
                        cnear &= cnear - 1u;
                    }
                    bool in;
                    if (kShape == 0) { // in_sphere, collision.cu.h:5-10
                        const float ax = px - ux, ay = py - uy, az = pz - uz;
                        in = lrm_sqrtf(ax * ax + ay * ay + az * az) < radius;
                    } else { // in_cylinder, collision.cu.h:12-23
                        const float dzz = uz - pz;
                        const float ax = ux - px, ay = uy - py;
                        in = (lrm_sqrtf(ax * ax + ay * ay + 0.f) < radius) && (dzz < plus_z) && (dzz > minus_z);
                    }
                    if (__ballot(ok && in) != 0ull) {
                        found = true;
                        break;
                    }
                }
            }
        }
        if (lane == 0) out[c] = found ? 1 : 0;
    }
}

// ---- launch functions (declared in lrm_launch.h) ---------------------------------------
hipError_t lrm_launch_warmup(size_t n, hipStream_t st) {
    hipLaunchKernelGGL(warmup_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st);
    return hipGetLastError();
}

hipError_t lrm_launch_reach_soa(const float* x, const float* y, const float* z, size_t n,
                                const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, bool fast, hipStream_t st) {
    fast = fast && L.fast_ok;
    const uintptr_t align = (uintptr_t)x | (uintptr_t)y | (uintptr_t)z;
    if ((align & 15) || ((uintptr_t)mask & 3)) {
        if (fast) hipLaunchKernelGGL(reach_soa_scalar_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
        else hipLaunchKernelGGL(reach_soa_scalar_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
        return hipGetLastError();
    }
    const int grid = grid_for((n + 3) / 4, LRM_REACH_GRID);
    if (bits && fast) hipLaunchKernelGGL((reach_soa_kernel<true, true>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
    else if (bits) hipLaunchKernelGGL((reach_soa_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
    else if (fast) hipLaunchKernelGGL((reach_soa_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
    else hipLaunchKernelGGL((reach_soa_kernel<false, false>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits);
    return hipGetLastError();
}

hipError_t lrm_launch_dist_soa(int op, const float* x, const float* y, const float* z, size_t n,
                               const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                               float* dz, bool fast, hipStream_t st) {
    fast = fast && L.fast_ok;
    // Compute-bound with a data-dependent iteration time: 8x more workgroups than are resident
    // (256 CUs x LRM_DIST_MIN_WAVES) lets the dispatcher even out the tail -- 7 % faster than one
    // grid-stride wave set (sweep: 2048 -> 0.289 ms, 5120 -> 0.273, 10240..20480 -> 0.262-0.266).
    const int grid = grid_for(n, 256 * LRM_DIST_MIN_WAVES * LRM_DIST_GRID_MULT);
    if (op == 2 && fast) hipLaunchKernelGGL((dist_soa_kernel<2, true>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz);
    else if (op == 2) hipLaunchKernelGGL((dist_soa_kernel<2, false>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz);
    else if (fast) hipLaunchKernelGGL((dist_soa_kernel<1, true>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz);
    else hipLaunchKernelGGL((dist_soa_kernel<1, false>), dim3(grid), dim3(kBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz);
    return hipGetLastError();
}

hipError_t lrm_launch_reach_aos(const float* xyz, size_t n, const LrmCompiledLeg& L, uint8_t* mask, bool fast,
                                hipStream_t st) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(xyz) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
    if (aligned && n >= 4) { // device allocations are: four points per lane, 16-byte loads
        const int grid = grid_for(n >> 2);
        if (fast && L.fast_ok) hipLaunchKernelGGL(reach_aos4_kernel<true>, dim3(grid), dim3(kBlock), 0, st, xyz, n, L, mask);
        else hipLaunchKernelGGL(reach_aos4_kernel<false>, dim3(grid), dim3(kBlock), 0, st, xyz, n, L, mask);
        return hipGetLastError();
    }
    if (fast && L.fast_ok) hipLaunchKernelGGL(reach_aos_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask);
    else hipLaunchKernelGGL(reach_aos_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask);
    return hipGetLastError();
}

hipError_t lrm_launch_dist_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, uint8_t* mask,
                               float* dxyz, bool fast, hipStream_t st) {
    fast = fast && L.fast_ok;
    if (op == 2 && fast) hipLaunchKernelGGL((dist_aos_kernel<2, true>), dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask, dxyz);
    else if (op == 2) hipLaunchKernelGGL((dist_aos_kernel<2, false>), dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask, dxyz);
    else if (fast) hipLaunchKernelGGL((dist_aos_kernel<1, true>), dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask, dxyz);
    else hipLaunchKernelGGL((dist_aos_kernel<1, false>), dim3(grid_for(n)), dim3(kBlock), 0, st, xyz, n, L, mask, dxyz);
    return hipGetLastError();
}

hipError_t lrm_launch_reach_any(const float* bx, const float* by, const float* bz, size_t nb, const float* tx,
                                const float* ty, const float* tz, size_t nt, const LrmCompiledLeg* legs_dev,
                                int nlegs, float* tile_boxes, bool boxes_ready, const uint8_t* body_active,
                                uint8_t* out_leg_body, uint8_t* all_legs_out, bool fast, hipStream_t st) {
    if (tile_boxes && nt && !boxes_ready) {
        hipLaunchKernelGGL(tile_aabb_kernel, dim3((unsigned)((nt + 1023) / 1024)), dim3(kBlock), 0, st, tx, ty, tz, nt, (nt + 1023) / 1024, tile_boxes);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    size_t groups = (nb + kWaves - 1) / kWaves;
    if (tile_boxes && nt) { // config 3: 1.0 ms against 1.9 ms (Morton order), 1.4 against 2.5 (raster), 7.4 against 10.4 (shuffled)
        const dim3 grid((unsigned)groups);
        if (fast) hipLaunchKernelGGL(reach_any_wave_kernel<true>, grid, dim3(kBlock), 0, st, bx, by, bz, nb, tx, ty, tz, nt, legs_dev, nlegs, tile_boxes, body_active, out_leg_body, all_legs_out);
        else hipLaunchKernelGGL(reach_any_wave_kernel<false>, grid, dim3(kBlock), 0, st, bx, by, bz, nb, tx, ty, tz, nt, legs_dev, nlegs, tile_boxes, body_active, out_leg_body, all_legs_out);
        return hipGetLastError();
    }
    if (groups > 256 * 16) groups = 256 * 16; // persistent over body groups beyond that
    const dim3 grid((unsigned)groups);
    if (fast) hipLaunchKernelGGL(reach_any_kernel<true>, grid, dim3(kBlock), 0, st, bx, by, bz, nb, tx, ty, tz, nt, legs_dev, nlegs, tile_boxes, body_active, out_leg_body, all_legs_out);
    else hipLaunchKernelGGL(reach_any_kernel<false>, grid, dim3(kBlock), 0, st, bx, by, bz, nb, tx, ty, tz, nt, legs_dev, nlegs, tile_boxes, body_active, out_leg_body, all_legs_out);
    return hipGetLastError();
}

hipError_t lrm_launch_any_in_shape(int shape, const float* cx, const float* cy, const float* cz, size_t nc,
                                   const float* tx, const float* ty, const float* tz, size_t nt, float radius,
                                   float plus_z, float minus_z, float* tile_boxes, bool boxes_ready, uint8_t* out,
                                   hipStream_t st) {
    if (tile_boxes && nt && !boxes_ready) {
        hipLaunchKernelGGL(tile_aabb_kernel, dim3((unsigned)((nt + 1023) / 1024)), dim3(kBlock), 0, st, tx, ty, tz, nt, (nt + 1023) / 1024, tile_boxes);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (tile_boxes && nt) {
        const dim3 wgrid((unsigned)std::min<size_t>((nc + kWaves - 1) / kWaves, (size_t)256 * 64));
        if (shape == 0)
            hipLaunchKernelGGL(any_in_shape_wave_kernel<0>, wgrid, dim3(kBlock), 0, st, cx, cy, cz, nc, tx, ty, tz, nt, radius,
                               plus_z, minus_z, tile_boxes, out);
        else
            hipLaunchKernelGGL(any_in_shape_wave_kernel<1>, wgrid, dim3(kBlock), 0, st, cx, cy, cz, nc, tx, ty, tz, nt, radius,
                               plus_z, minus_z, tile_boxes, out);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((nc + kBlock - 1) / kBlock));
    if (shape == 0)
        hipLaunchKernelGGL(any_in_shape_kernel<0>, grid, dim3(kBlock), 0, st, cx, cy, cz, nc, tx, ty, tz, nt, radius,
                           plus_z, minus_z, tile_boxes, out);
    else
        hipLaunchKernelGGL(any_in_shape_kernel<1>, grid, dim3(kBlock), 0, st, cx, cy, cz, nc, tx, ty, tz, nt, radius,
                           plus_z, minus_z, tile_boxes, out);
    return hipGetLastError();
}

hipError_t lrm_launch_sqrt_check(unsigned long long* counters_dev, hipStream_t st) {
    hipLaunchKernelGGL(sqrt_check_kernel, dim3(2048), dim3(kBlock), 0, st, counters_dev);
    return hipGetLastError();
}

hipError_t lrm_launch_exact_math(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs,
                                 hipStream_t st) {
    hipLaunchKernelGGL(exact_math_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, a, b, n, at2, sn, cs);
    return hipGetLastError();
}

hipError_t lrm_launch_rotate_soa(const float* sx, const float* sy, const float* sz, size_t n, const LrmCompiledLeg* rot_dev,
                                 float* dx, float* dy, float* dz, hipStream_t st) {
    hipLaunchKernelGGL(rotate_soa_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, sx, sy, sz, n, rot_dev, dx, dy, dz);
    return hipGetLastError();
}

hipError_t lrm_launch_sweep_update(const uint8_t* all_legs, const uint8_t* cyl_validate, const uint8_t* cyl_eliminate,
                                   int use_culls, size_t nb, uint8_t* active, uint8_t* accepted, hipStream_t st) {
    hipLaunchKernelGGL(sweep_update_kernel, dim3(grid_for(nb)), dim3(kBlock), 0, st, all_legs, cyl_validate, cyl_eliminate,
                       use_culls, nb, active, accepted);
    return hipGetLastError();
}

// counters of a -DLRM_PAIR_COUNT build (read and reset); an ordinary build has none
hipError_t lrm_pair_counts(unsigned long long out[4]) {
#if defined(LRM_PAIR_COUNT)
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_count), 4 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    const unsigned long long zero[4] = {0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pair_count), zero, sizeof zero);
#else
    (void)out;
    return hipErrorNotSupported;
#endif
}

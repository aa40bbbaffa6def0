// lrm_tol_kernels.hip -- gfx950 kernels of the contract-tolerance mode (LRM_MODE_TOL, lrm_point_tol.h).
//
// Two launches per call, no host synchronisation between them and NO global atomics (a first version appended
// the doubtful points to one queue with one atomicAdd per wave: 42 000 same-address atomics per 1e7 points cost
// 0.42 ms, five times the arithmetic):
//   dist_tol_kernel   every point: FP32-FMA / v_rsq_f32 evaluation (no trigonometry, no IEEE sqrt / div), reach
//                     mask + ballot bit words + distance vector; a point with any decision inside its error band
//                     is appended to the SEGMENT of its workgroup (kSegCap slots per workgroup, slot numbers from
//                     an LDS counter); the workgroup stores its count at the end.
//   tol_fixup_kernel  one wave per kSegPerWave segments: prefix sum of their counts, then the queued points
//                     (a few 1e-3 of the cloud), 64 at a time, through the bit-exact filtered code of
//                     LRM_MODE_FAST, overwriting their outputs.  A segment that overflowed (a cloud hugging a
//                     decision boundary) has ALL the points of its workgroup re-evaluated: slow, never wrong.
//                     Counts are rewritten by every call: nothing to reset.
// Layout as lrm_kernels.hip: SoA coordinates, byte mask, ballot words, SoA distance field.  The per-leg block
// (LrmTolLeg, 1.5 KB) travels by value in the kernarg segment; its per-lane tables are staged in LDS.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "lrm_launch.h"
#include "lrm_types.h"
#define LRM_FRESH(L) lrm_fresh(L)
#include "lrm_point.h"
#include "lrm_point_fast.h"
#include "lrm_point_tol.h"

#ifndef LRM_TOL_MIN_WAVES
#define LRM_TOL_MIN_WAVES 6
#endif
#ifndef LRM_TOL_SEG_PER_WAVE
#define LRM_TOL_SEG_PER_WAVE 8 // ~30 queued points per fix-up wave at the usual 0.45 % of doubt: one batch
#endif
#ifndef LRM_TOL_SEG_CAP
#define LRM_TOL_SEG_CAP 32
#endif
#ifndef LRM_TOL_GRID_MULT
#define LRM_TOL_GRID_MULT 8
#endif

namespace {

constexpr int kBlock = 256;
constexpr int kFixBlock = 64;
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

struct TolLds {
    LrmTolLeg::Circle circ[16];
    LrmCircle feat[LRM_TOL_FEATS];
};

// kOp 1: distance + optional validity byte; kOp 2: reach mask (+ bit words) + distance.  In this mode the two
// flags are the same function of the point wherever no decision is in doubt.
template <int kOp>
__global__ __launch_bounds__(kBlock, LRM_TOL_MIN_WAVES) void dist_tol_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmTolLeg L_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, uint32_t* __restrict__ queue, uint32_t* __restrict__ counts) {
    __shared__ TolLds s_tab;
    __shared__ uint32_t s_qn;
    const LrmTolLeg& L = lrm_kernarg<LrmTolLeg>(kTolLegArg);
    {
        const float* csrc = reinterpret_cast<const float*>(&L.circ[0][0]);
        const float* fsrc = reinterpret_cast<const float*>(&L.feat[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.circ) / 4); i += kBlock) reinterpret_cast<float*>(s_tab.circ)[i] = csrc[i];
        if (threadIdx.x < (int)(sizeof(s_tab.feat) / 4)) reinterpret_cast<float*>(s_tab.feat)[threadIdx.x] = fsrc[threadIdx.x];
        if (threadIdx.x == 0) s_qn = 0;
        __syncthreads();
    }
    const LrmTolTables T{s_tab.circ, s_tab.feat};
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_pad = (n + 63) & ~(size_t)63; // whole waves iterate together (ballots below)
    uint32_t* seg = queue + (size_t)blockIdx.x * kSegCap;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
        bool m = false;
        uint32_t doubt = 0;
        if (i < n) {
            LrmVec3 p{x[i], y[i], z[i]};
            m = lrm_dist_tol(L, T, p, doubt);
            doubt &= 0xffffu; // the statistics bits do not queue a point
            dx[i] = p.x;
            dy[i] = p.y;
            dz[i] = p.z;
            if (mask) mask[i] = m;
        }
        if (bits) {
            const uint64_t w = __ballot(m);
            if ((threadIdx.x & 63) == 0) bits[i >> 6] = w;
        }
#if !defined(LRM_TOL_NOQUEUE)
        const uint64_t dm = __ballot(doubt != 0);
        if (dm) { // rare (a quarter of the waves): one LDS atomic per wave with a doubtful lane
            const int lane = threadIdx.x & 63;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_qn, (uint32_t)__popcll(dm));
            base = __shfl(base, 0);
            if (doubt) {
                const uint32_t slot = base + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (slot < (uint32_t)kSegCap) seg[slot] = (uint32_t)i; // beyond: the count tells the fix-up to redo the workgroup
            }
        }
#endif
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_qn;
}

struct FixLds {
    LrmCircle lists[16];
    LrmCompiledLeg::DistCircle dist[16];
    LrmCircle corners[LRM_N_CORNERS];
};

// One wave: the segments [blockIdx.x * kSegPerWave, +kSegPerWave) of a dist_tol_kernel launch of `nseg` workgroups
// with grid stride `main_stride` points.
template <int kOp>
__global__ __launch_bounds__(kFixBlock) void tol_fixup_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmCompiledLeg L_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const uint32_t* __restrict__ queue,
    const uint32_t* __restrict__ counts, uint32_t nseg, size_t main_stride) {
    __shared__ FixLds s_tab;
    __shared__ uint32_t s_pre[kSegPerWave + 1], s_cnt[kSegPerWave];
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kFixLegArg);
    const uint32_t seg0 = blockIdx.x * kSegPerWave;
    const int lane = threadIdx.x;
    if (lane < kSegPerWave) s_cnt[lane] = (seg0 + lane < nseg) ? counts[seg0 + lane] : 0u;
    __syncthreads();
    if (lane == 0) {
        uint32_t acc = 0;
        for (int j = 0; j < kSegPerWave; j++) {
            s_pre[j] = acc;
            acc += s_cnt[j] <= (uint32_t)kSegCap ? s_cnt[j] : 0u; // an overflowed segment is redone as a whole below
        }
        s_pre[kSegPerWave] = acc;
    }
    uint32_t any = 0;
    for (int j = 0; j < kSegPerWave; j++) any |= s_cnt[j];
    if (any == 0) return; // nothing in doubt in these workgroups (wave-uniform)
    {
        const float* src = reinterpret_cast<const float*>(&L.lists[0][0]);
        const float* dsrc = reinterpret_cast<const float*>(&L.dist_tab[0][0]);
        const float* csrc = reinterpret_cast<const float*>(&L.corner_tab[0]);
        for (int i = lane; i < (int)(sizeof(s_tab.lists) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.lists)[i] = src[i];
        for (int i = lane; i < (int)(sizeof(s_tab.dist) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.dist)[i] = dsrc[i];
        for (int i = lane; i < (int)(sizeof(s_tab.corners) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.corners)[i] = csrc[i];
    }
    __syncthreads();
    const LrmDistTables T{s_tab.lists, s_tab.dist, s_tab.corners};
    auto redo = [&](size_t i) {
        LrmVec3 p{x[i], y[i], z[i]};
        bool m = false;
        if (kOp == 2) lrm_reach_dist_global_filtered(L, T, p, m);
        else m = lrm_dist_global_filtered(L, T, p);
        dx[i] = p.x;
        dy[i] = p.y;
        dz[i] = p.z;
        if (mask) mask[i] = m;
        if (bits) { // one bit of a word the tolerance kernel wrote: distinct addresses, no contention
            unsigned long long* w = reinterpret_cast<unsigned long long*>(bits) + (i >> 6);
            const unsigned long long bit = 1ull << (i & 63);
            if (m) atomicOr(w, bit);
            else atomicAnd(w, ~bit);
        }
    };
    const uint32_t total = s_pre[kSegPerWave];
    for (uint32_t k = lane; k < total; k += kFixBlock) {
        int j = 0;
#pragma unroll
        for (int t = 1; t < kSegPerWave; t++) j += (s_pre[t] <= k) ? 1 : 0; // segments with nothing queued share a prefix
        redo((size_t)queue[(size_t)(seg0 + j) * kSegCap + (k - s_pre[j])]);
    }
    for (int j = 0; j < kSegPerWave; j++) {
        if (s_cnt[j] <= (uint32_t)kSegCap) continue; // wave-uniform
        // every point of workgroup seg0 + j: i = (seg0 + j) * kBlock + t + round * main_stride
        for (size_t base = (size_t)(seg0 + j) * kBlock; base < n; base += main_stride)
            for (int t = lane; t < kBlock; t += kFixBlock)
                if (base + t < n) redo(base + t);
    }
}

} // namespace

size_t lrm_tol_queue_words(void) { return (size_t)256 * LRM_TOL_MIN_WAVES * LRM_TOL_GRID_MULT * (kSegCap + 1); }

hipError_t lrm_launch_dist_tol(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz,
                               uint32_t* workspace /* lrm_tol_queue_words() uint32 */, hipStream_t st) {
    const size_t cap = (size_t)256 * LRM_TOL_MIN_WAVES * LRM_TOL_GRID_MULT;
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    uint32_t* counts = workspace;
    uint32_t* queue = workspace + cap;
    if (op == 2) hipLaunchKernelGGL(dist_tol_kernel<2>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counts);
    else hipLaunchKernelGGL(dist_tol_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned fblocks = (unsigned)((blocks + kSegPerWave - 1) / kSegPerWave);
    const size_t stride = blocks * kBlock;
    if (op == 2) hipLaunchKernelGGL(tol_fixup_kernel<2>, dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts, (uint32_t)blocks, stride);
    else hipLaunchKernelGGL(tol_fixup_kernel<1>, dim3(fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counts, (uint32_t)blocks, stride);
    return hipGetLastError();
}

// lrm_tol_kernels.hip -- gfx950 kernels of the contract-tolerance mode (LRM_MODE_TOL, lrm_point_tol.h).
//
// Two launches per call, no host synchronisation between them:
//   dist_tol_kernel   every point: FP32-FMA / v_rsq_f32 evaluation (no trigonometry, no IEEE sqrt / div), reach
//                     mask + ballot bit words + distance vector; a point with any decision inside its error band
//                     is appended to a device queue (one atomic per wave that has such a lane).
//   tol_fixup_kernel  the queued points (a few 1e-3 of the cloud) once more with the bit-exact filtered code of
//                     LRM_MODE_FAST, overwriting their outputs.  If the queue overflowed (a cloud hugging a
//                     decision boundary) it re-evaluates EVERY point: slow, never wrong.  The last block to
//                     finish resets the queue counter for the next call.
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

namespace {

constexpr int kBlock = 256;
constexpr int kFixBlock = 64;

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

#ifndef LRM_TOL_MIN_WAVES
#define LRM_TOL_MIN_WAVES 6
#endif
#ifndef LRM_TOL_GRID_MULT
#define LRM_TOL_GRID_MULT 8
#endif

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
    float* __restrict__ dy, float* __restrict__ dz, uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount,
    uint32_t qcap) {
    __shared__ TolLds s_tab;
    const LrmTolLeg& L = lrm_kernarg<LrmTolLeg>(kTolLegArg);
    {
        const float* csrc = reinterpret_cast<const float*>(&L.circ[0][0]);
        const float* fsrc = reinterpret_cast<const float*>(&L.feat[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.circ) / 4); i += kBlock) reinterpret_cast<float*>(s_tab.circ)[i] = csrc[i];
        if (threadIdx.x < (int)(sizeof(s_tab.feat) / 4)) reinterpret_cast<float*>(s_tab.feat)[threadIdx.x] = fsrc[threadIdx.x];
        __syncthreads();
    }
    const LrmTolTables T{s_tab.circ, s_tab.feat};
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_pad = (n + 63) & ~(size_t)63; // whole waves iterate together (ballots below)
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
        const uint64_t dm = __ballot(doubt != 0);
        if (dm) { // rare: one atomic per wave with a doubtful lane
            const int lane = threadIdx.x & 63;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(qcount, (uint32_t)__popcll(dm));
            base = __shfl(base, 0);
            if (doubt) {
                const uint32_t slot = base + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (slot < qcap) queue[slot] = (uint32_t)i;
            }
        }
    }
}

struct FixLds {
    LrmCircle lists[16];
    LrmCompiledLeg::DistCircle dist[16];
    LrmCircle corners[LRM_N_CORNERS];
};

template <int kOp>
__global__ __launch_bounds__(kFixBlock) void tol_fixup_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const LrmCompiledLeg L_kernarg, uint8_t* __restrict__ mask, uint64_t* __restrict__ bits, float* __restrict__ dx,
    float* __restrict__ dy, float* __restrict__ dz, const uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount,
    uint32_t qcap, uint32_t* __restrict__ done) {
    __shared__ FixLds s_tab;
    const LrmCompiledLeg& L = lrm_kernarg<LrmCompiledLeg>(kFixLegArg);
    const uint32_t count = *reinterpret_cast<volatile uint32_t*>(qcount);
    const bool overflow = count > qcap;
    const size_t work = overflow ? n : (size_t)count;
    const size_t stride = (size_t)gridDim.x * kFixBlock;
    const size_t first = (size_t)blockIdx.x * kFixBlock + threadIdx.x;
    if ((size_t)blockIdx.x * kFixBlock < work) { // this block has something to do
        const float* src = reinterpret_cast<const float*>(&L.lists[0][0]);
        const float* dsrc = reinterpret_cast<const float*>(&L.dist_tab[0][0]);
        const float* csrc = reinterpret_cast<const float*>(&L.corner_tab[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.lists) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.lists)[i] = src[i];
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.dist) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.dist)[i] = dsrc[i];
        for (int i = threadIdx.x; i < (int)(sizeof(s_tab.corners) / 4); i += kFixBlock) reinterpret_cast<float*>(s_tab.corners)[i] = csrc[i];
        __syncthreads();
        const LrmDistTables T{s_tab.lists, s_tab.dist, s_tab.corners};
        for (size_t k = first; k < work; k += stride) {
            const size_t i = overflow ? k : (size_t)queue[k];
            LrmVec3 p{x[i], y[i], z[i]};
            bool m = false;
            if (kOp == 2) {
                lrm_reach_dist_global_filtered(L, T, p, m);
            } else {
                m = lrm_dist_global_filtered(L, T, p);
            }
            dx[i] = p.x;
            dy[i] = p.y;
            dz[i] = p.z;
            if (mask) mask[i] = m;
            if (bits) {
                unsigned long long* w = reinterpret_cast<unsigned long long*>(bits) + (i >> 6);
                const unsigned long long bit = 1ull << (i & 63);
                if (m) atomicOr(w, bit);
                else atomicAnd(w, ~bit);
            }
        }
    }
    // the last block to arrive resets the queue for the next call (every block has read `count` by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t prev = atomicAdd(done, 1u);
        if (prev == gridDim.x - 1) {
            *qcount = 0;
            *done = 0;
            __threadfence();
        }
    }
}

} // namespace

hipError_t lrm_launch_dist_tol(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz,
                               uint32_t* queue, uint32_t qcap, uint32_t* counters /* [0] queue length, [1] blocks done */,
                               hipStream_t st) {
    size_t blocks = (n + kBlock - 1) / kBlock;
    const size_t cap = (size_t)256 * LRM_TOL_MIN_WAVES * LRM_TOL_GRID_MULT;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    if (op == 2) hipLaunchKernelGGL(dist_tol_kernel<2>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counters, qcap);
    else hipLaunchKernelGGL(dist_tol_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, y, z, n, TL, mask, bits, dx, dy, dz, queue, counters, qcap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // enough 64-lane blocks for a queue of n/16 points in one round; longer queues (and the overflow path) stride
    size_t fblocks = (n / 16 + kFixBlock - 1) / kFixBlock;
    if (fblocks < 1) fblocks = 1;
    if (fblocks > 8192) fblocks = 8192;
    if (op == 2) hipLaunchKernelGGL(tol_fixup_kernel<2>, dim3((unsigned)fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counters, qcap, counters + 1);
    else hipLaunchKernelGGL(tol_fixup_kernel<1>, dim3((unsigned)fblocks), dim3(kFixBlock), 0, st, x, y, z, n, L, mask, bits, dx, dy, dz, queue, counters, qcap, counters + 1);
    return hipGetLastError();
}

// lrm_toltab_dev.hip -- the plane table with deferred decisions (LrmTolTabHeader, lrm_types.h) built ON THE DEVICE.
//
// Round 3 built it on eight host threads in ~30 ms per (leg, orientation) -- 340 steps of the kernel it serves, paid by the first
// call for every new orientation.  The cells are independent, so here they are classified by the GPU, with the very functions the
// host builder runs (lrm_toltab_build.h: double arithmetic made of correctly rounded operations only): the two builders give the
// same table byte for byte (tests/test_gpu_toltab.py).
//
//   classify_coarse_kernel  one lane per coarse cell of both grids (2 x 128 x 128): the cell's answer (six bytes + its lower
//                           bound); an unanswered cell takes a slot in the grid's refinement list
//   classify_fine_kernel    one workgroup per listed cell, one lane per 1-mm (8-mm) sub-cell: 256 answers, whether any of them
//                           is one, the cell's bound as the minimum of theirs; the canonical rows in use are OR-ed into two words
//   -- one small read-back: the counts, the rows in use, which cells are refined (32 KB) -> the host numbers the rows and the
//      refined cells (in cell order, as the host builder does), sizes the table and uploads its header --
//   emit_kernel             cell codes from the canonical ids and the row numbers
//   bounds_kernel           one workgroup per bound cell (64 x 64): its 16 x 16 own bounds in parallel, the plane fit by one lane
//                           in the host builder's summation order
// Everything runs on the caller's stream; the read-back is the only synchronisation (the first call for a (leg, orientation)
// allocates the table anyway).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "lrm_compile.h"
#include "lrm_toltab_build.h"

namespace {

constexpr int kN = LRM_TT_N, kCells = LRM_TT_N * LRM_TT_N, kSub2 = LRM_TT_SUB * LRM_TT_SUB;
constexpr int kMaxRefine = 2048; // unanswered coarse cells per grid the device builder takes (the reference robots: ~300); beyond: the host builder

struct DevScratch { // one per device, kept (17 MB)
    LrmTbInput* in = nullptr;
    LrmTbCell* coarse = nullptr;   // [2][kCells]
    LrmTbCell* fine = nullptr;     // [2][kMaxRefine][kSub2]
    uint32_t* list = nullptr;      // [2][kMaxRefine] cell numbers of the unanswered coarse cells
    int32_t* slot_of = nullptr;    // [2][kCells] slot in the list, or -1
    uint8_t* state = nullptr;      // [2][kCells] 0 answered, 1 unanswered, 2 refined (some sub-cell has an answer)
    uint32_t* status = nullptr;    // [8]: unanswered count per grid (2), used rows lo / hi, used validity rows
    uint16_t* fine_of = nullptr;   // [2][kCells] number of the cell's fine block (host numbering)
    uint8_t* row_num = nullptr;    // [LRM_TB_ROWS + LRM_TB_VROWS]
};
DevScratch g_scratch[64];

// the rows a lane uses, OR-ed over the wave: one lane issues the three atomics (180 000 same-address atomics would take milliseconds)
__device__ __forceinline__ void mark_rows(uint32_t* status, bool ok, const LrmTbCell& c) {
    unsigned long long rows = ok ? ((1ull << c.t[0]) | (1ull << c.t[1])) : 0ull;
    uint32_t vr = ok ? (1u << c.v) : 0u;
    for (int off = 32; off >= 1; off >>= 1) {
        rows |= (unsigned long long)__shfl_xor((long long)rows, off);
        vr |= (uint32_t)__shfl_xor((int)vr, off);
    }
    if ((threadIdx.x & 63) == 0 && (rows | vr)) {
        atomicOr(&status[2], (uint32_t)rows);
        atomicOr(&status[3], (uint32_t)(rows >> 32));
        atomicOr(&status[4], vr);
    }
}

__global__ __launch_bounds__(64) void classify_coarse_kernel(const LrmTbInput* __restrict__ in, LrmTbCell* __restrict__ coarse, uint32_t* __restrict__ list,
                                                              int32_t* __restrict__ slot_of, uint8_t* __restrict__ state, uint32_t* __restrict__ status) {
    const int gid = blockIdx.x * 64 + threadIdx.x; // (2 kCells is a multiple of 64: every lane has a cell)
    const int g = gid / kCells, cell = gid % kCells;
    const double H = g ? (double)LRM_TT_H_OUTER : (double)LRM_TT_H_INNER;
    const LrmTbCell c = lrm_tb_coarse_cell(*in, g, H, cell % kN, cell / kN);
    coarse[gid] = c;
    int32_t slot = -1;
    if (!c.ok) {
        const uint32_t s = atomicAdd(&status[g], 1u);
        if (s < (uint32_t)kMaxRefine) {
            list[g * kMaxRefine + s] = (uint32_t)cell;
            slot = (int32_t)s;
        }
    }
    mark_rows(status, c.ok != 0, c);
    slot_of[gid] = slot;
    state[gid] = c.ok ? 0 : 1;
}

__global__ __launch_bounds__(256) void classify_fine_kernel(const LrmTbInput* __restrict__ in, LrmTbCell* __restrict__ coarse, LrmTbCell* __restrict__ fine,
                                                            const uint32_t* __restrict__ list, uint8_t* __restrict__ state, uint32_t* __restrict__ status) {
    const int g = blockIdx.x / kMaxRefine, slot = blockIdx.x % kMaxRefine;
    const uint32_t count = status[g] < (uint32_t)kMaxRefine ? status[g] : (uint32_t)kMaxRefine;
    if ((uint32_t)slot >= count) return; // workgroup-uniform
    const uint32_t cell = list[g * kMaxRefine + slot];
    const double H = g ? (double)LRM_TT_H_OUTER : (double)LRM_TT_H_INNER;
    const int t = threadIdx.x;
    const LrmTbCell f = lrm_tb_fine_cell(*in, g, H, (int)(cell % kN), (int)(cell / kN), t % LRM_TT_SUB, t / LRM_TT_SUB);
    fine[((size_t)g * kMaxRefine + slot) * kSub2 + t] = f;
    __shared__ double s_lb[256];
    __shared__ uint32_t s_any;
    if (t == 0) s_any = 0;
    s_lb[t] = f.lb;
    __syncthreads();
    if (f.ok) atomicOr(&s_any, 1u);
    mark_rows(status, f.ok != 0, f);
    for (int off = 128; off >= 1; off >>= 1) { // the minimum: exact in any order
        if (t < off) s_lb[t] = lrm_tb_min(s_lb[t], s_lb[t + off]);
        __syncthreads();
    }
    if (t == 0) {
        LrmTbCell& c = coarse[(size_t)g * kCells + cell];
        c.lb = lrm_tb_max(s_lb[0], c.lb); // the sub-cells' bounds are tighter than the coarse cell's own: their minimum holds for the whole cell
        state[(size_t)g * kCells + cell] = s_any ? 2 : 1;
    }
}

struct TabLayout {
    uint32_t coarse_off[2], fine_off[2], n_fine[2];
};
__device__ __forceinline__ uint16_t code_of(const LrmTbCell& c, const uint8_t* row_num) {
    if (!c.ok) return (uint16_t)LRM_TT_UNANSWERED;
    return (uint16_t)((unsigned)row_num[c.t[0]] | ((unsigned)row_num[c.t[1]] << 5) | ((unsigned)row_num[LRM_TB_ROWS + c.v] << 10));
}

// cells: the uint16 arrays behind the table's header.  blocks [0, 2 kCells / 256): coarse entries; then one block per list slot: fine blocks
__global__ __launch_bounds__(256) void emit_kernel(const LrmTbCell* __restrict__ coarse, const LrmTbCell* __restrict__ fine, const uint32_t* __restrict__ list,
                                                   const uint8_t* __restrict__ state, const uint16_t* __restrict__ fine_of, const uint8_t* __restrict__ row_num,
                                                   const uint32_t* __restrict__ status, const TabLayout hd, uint16_t* __restrict__ cells) {
    const int nb_coarse = 2 * kCells / 256;
    if ((int)blockIdx.x < nb_coarse) {
        const int gid = blockIdx.x * 256 + threadIdx.x;
        const int g = gid / kCells, cell = gid % kCells;
        cells[hd.coarse_off[g] + cell] = state[gid] == 2 ? (uint16_t)(0x8000u | (unsigned)fine_of[gid]) : code_of(coarse[gid], row_num);
        if (cell < kSub2 && hd.n_fine[g] == 0) cells[hd.fine_off[g] + cell] = (uint16_t)LRM_TT_UNANSWERED; // the spare block of a grid without refined cells
        return;
    }
    const int b = blockIdx.x - nb_coarse;
    const int g = b / kMaxRefine, slot = b % kMaxRefine;
    const uint32_t count = status[g] < (uint32_t)kMaxRefine ? status[g] : (uint32_t)kMaxRefine;
    if ((uint32_t)slot >= count) return;
    const uint32_t cell = list[g * kMaxRefine + slot];
    if (state[(size_t)g * kCells + cell] != 2) return;
    const size_t at = (size_t)hd.fine_off[g] + (size_t)fine_of[(size_t)g * kCells + cell] * kSub2 + threadIdx.x;
    cells[at] = code_of(fine[((size_t)g * kMaxRefine + slot) * kSub2 + threadIdx.x], row_num);
}

// the inner grid's bounds: one workgroup per bound cell, lane = sub-cell (sz * 16 + sx)
__global__ __launch_bounds__(256) void bounds_kernel(const LrmTbInput* __restrict__ in, const LrmTbCell* __restrict__ coarse, const LrmTbCell* __restrict__ fine,
                                                     const int32_t* __restrict__ slot_of, const uint8_t* __restrict__ state, uint32_t bound_off, uint16_t* __restrict__ cells) {
    constexpr int S = LRM_TB_S, NB = LRM_TT_NB;
    const int bx = blockIdx.x % NB, bz = blockIdx.x / NB;
    const int sx = threadIdx.x % S, sz = threadIdx.x / S;
    const int ix = 2 * bx + sx / (S / 2), iz = 2 * bz + sz / (S / 2);
    const size_t ci = (size_t)iz * kN + ix;
    const LrmTbCell c = coarse[ci];
    const LrmTbCell* f = (state[ci] == 2) ? fine + (size_t)slot_of[ci] * kSub2 : nullptr; // (grid 0: slots [0, kMaxRefine))
    __shared__ double s_lb[S * S];
    s_lb[sz * S + sx] = lrm_tb_subcell_lb(*in, (double)LRM_TT_H_INNER, bx, bz, sx, sz, c, f);
    __syncthreads();
    if (threadIdx.x == 0) { // the fit's sums in the host builder's order: they enter a rounding decision (the quantised gradient)
        const uint32_t w = lrm_tb_bound_word(lrm_tb_fit_bound(s_lb, (double)LRM_TT_H_INNER));
        const size_t at = (size_t)bound_off + 2 * ((size_t)bz * NB + bx);
        cells[at] = (uint16_t)(w & 0xffffu);
        cells[at + 1] = (uint16_t)(w >> 16);
    }
}

hipError_t ensure_scratch(int dev, DevScratch** out) {
    DevScratch& S = g_scratch[dev];
    if (!S.in) {
        hipError_t e;
        auto A = [&](void** p, size_t bytes) { return hipMalloc(p, bytes); };
        if ((e = A((void**)&S.in, sizeof(LrmTbInput))) != hipSuccess) return e;
        if ((e = A((void**)&S.coarse, sizeof(LrmTbCell) * 2 * kCells)) != hipSuccess) return e;
        if ((e = A((void**)&S.fine, sizeof(LrmTbCell) * 2 * (size_t)kMaxRefine * kSub2)) != hipSuccess) return e;
        if ((e = A((void**)&S.list, sizeof(uint32_t) * 2 * kMaxRefine)) != hipSuccess) return e;
        if ((e = A((void**)&S.slot_of, sizeof(int32_t) * 2 * kCells)) != hipSuccess) return e;
        if ((e = A((void**)&S.state, 2 * kCells)) != hipSuccess) return e;
        if ((e = A((void**)&S.status, 8 * sizeof(uint32_t))) != hipSuccess) return e;
        if ((e = A((void**)&S.fine_of, sizeof(uint16_t) * 2 * kCells)) != hipSuccess) return e;
        if ((e = A((void**)&S.row_num, LRM_TB_ROWS + LRM_TB_VROWS)) != hipSuccess) return e;
    }
    *out = &S;
    return hipSuccess;
}

} // namespace

void lrm_toltab_dev_release() {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < 64; d++) {
        DevScratch& S = g_scratch[d];
        if (!S.in) continue;
        (void)hipSetDevice(d);
        for (void* p : {(void*)S.in, (void*)S.coarse, (void*)S.fine, (void*)S.list, (void*)S.slot_of, (void*)S.state, (void*)S.status, (void*)S.fine_of, (void*)S.row_num})
            if (p) (void)hipFree(p);
        S = DevScratch{};
    }
    if (have) (void)hipSetDevice(cur);
}

// Builds the table of L on the current device, on `st`.  *tab_dev_out: a fresh hipMalloc-ed table (the caller owns it), *bytes_out
// its size; *ms_out (optional) the device time of the build (HIP events around it, the read-back included).
// Returns 0 ok; 1 this leg has no table (more rows than a cell code can name); 2 the device builder does not take this leg (more
// unanswered cells than its scratch holds): use the host builder; < 0 a HIP error (hipError_t negated).
int lrm_build_tol_tab_dev(const LrmTolLeg& L, hipStream_t st, uint8_t** tab_dev_out, size_t* bytes_out, float* ms_out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    if (dev < 0 || dev >= 64) return 2;
    DevScratch* S = nullptr;
    if ((e = ensure_scratch(dev, &S)) != hipSuccess) return -(int)e;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ms_out) {
        if ((e = hipEventCreate(&ev0)) != hipSuccess) return -(int)e;
        if ((e = hipEventCreate(&ev1)) != hipSuccess) { (void)hipEventDestroy(ev0); return -(int)e; }
        (void)hipEventRecord(ev0, st);
    }
    auto done = [&](int rc) {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        return rc;
    };
    LrmTbInput in;
    lrm_tb_make_input(L, &in);
    if ((e = hipMemcpyAsync(S->in, &in, sizeof in, hipMemcpyHostToDevice, st)) != hipSuccess) return done(-(int)e);
    if ((e = hipMemsetAsync(S->status, 0, 8 * sizeof(uint32_t), st)) != hipSuccess) return done(-(int)e);
    hipLaunchKernelGGL(classify_coarse_kernel, dim3(2 * kCells / 64), dim3(64), 0, st, S->in, S->coarse, S->list, S->slot_of, S->state, S->status);
    hipLaunchKernelGGL(classify_fine_kernel, dim3(2 * kMaxRefine), dim3(256), 0, st, S->in, S->coarse, S->fine, S->list, S->state, S->status);
    if ((e = hipGetLastError()) != hipSuccess) return done(-(int)e);
    // ---- the one read-back ----
    uint32_t status[8];
    std::vector<uint8_t> state(2 * kCells);
    if ((e = hipMemcpyAsync(status, S->status, sizeof status, hipMemcpyDeviceToHost, st)) != hipSuccess) return done(-(int)e);
    if ((e = hipMemcpyAsync(state.data(), S->state, state.size(), hipMemcpyDeviceToHost, st)) != hipSuccess) return done(-(int)e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return done(-(int)e);
    if (status[0] > (uint32_t)kMaxRefine || status[1] > (uint32_t)kMaxRefine) return done(2);
    // rows in use; row 0 / validity rows 0 and 1 always exist
    const uint64_t used_rows = ((uint64_t)status[2] | ((uint64_t)status[3] << 32)) | 1ull;
    const uint32_t used_vrows = status[4] | 3u;
    LrmTolTabHeader hd;
    memset(&hd, 0, sizeof hd);
    uint8_t nums[LRM_TB_ROWS + LRM_TB_VROWS];
    if (!lrm_tb_number_rows(in, used_rows, used_vrows, nums, nums + LRM_TB_ROWS, &hd)) return done(1);
    // refined cells numbered in cell order, as the host builder numbers them
    std::vector<uint16_t> fine_of(2 * kCells, 0);
    uint32_t n_fine[2] = {0, 0};
    for (int g = 0; g < 2; g++)
        for (int i = 0; i < kCells; i++)
            if (state[(size_t)g * kCells + i] == 2) fine_of[(size_t)g * kCells + i] = (uint16_t)n_fine[g]++;
    if (n_fine[0] > 0x7fff || n_fine[1] > 0x7fff) return done(1);
    size_t n_cells = 0;
    lrm_tb_layout(n_fine, &hd, &n_cells);
    hd.band_max = (float)in.band[0];
    hd.band_max_outer = (float)in.band[1];
    hd.far_limit = (float)(0.5 * kN * (double)LRM_TT_H_INNER - 1.0);
    const size_t bytes = sizeof hd + n_cells * 2;
    void* tab = nullptr;
    if ((e = hipMalloc(&tab, bytes)) != hipSuccess) return done(-(int)e);
    auto fail = [&](hipError_t err) { (void)hipFree(tab); return done(-(int)err); };
    if ((e = hipMemsetAsync(tab, 0, bytes, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(tab, &hd, sizeof hd, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(S->fine_of, fine_of.data(), fine_of.size() * 2, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(S->row_num, nums, sizeof nums, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    uint16_t* cells = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(tab) + sizeof hd);
    const TabLayout lay{{hd.coarse_off[0], hd.coarse_off[1]}, {hd.fine_off[0], hd.fine_off[1]}, {hd.n_fine[0], hd.n_fine[1]}};
    hipLaunchKernelGGL(emit_kernel, dim3(2 * kCells / 256 + 2 * kMaxRefine), dim3(256), 0, st, S->coarse, S->fine, S->list, S->state, S->fine_of, S->row_num, S->status, lay, cells);
    hipLaunchKernelGGL(bounds_kernel, dim3(LRM_TT_NB * LRM_TT_NB), dim3(256), 0, st, S->in, S->coarse, S->fine, S->slot_of, S->state, hd.bound_off[0], cells);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    // (the copies above read hd, fine_of, nums: the call returns after the stream has drained -- once per (leg, orientation))
    if (ms_out) (void)hipEventRecord(ev1, st);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
    if (ms_out) (void)hipEventElapsedTime(ms_out, ev0, ev1);
    *tab_dev_out = static_cast<uint8_t*>(tab);
    *bytes_out = bytes;
    return done(0);
}

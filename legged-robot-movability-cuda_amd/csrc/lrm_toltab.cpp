// lrm_toltab.cpp -- builds the plane table with deferred decisions of the tolerance mode (LrmTolTabHeader,
// lrm_types.h) on the host, once per (leg, orientation).
//
// For a cell (centre c, half-diagonal rho) and each region list a point of the cell can be evaluated with:
//   point validity v_j = |q - c_j|^2 gs_j + c_j, Lipschitz 2 |gs_j| (|c - c_j| + rho): IN / OUT with margin, or OPEN.
//       One OUT: invalid all over the cell.  All IN: valid.  Exactly one OPEN (and no OUT): that circle decides at run time.
//   clamp targets: circle i with its clamp validity ALWAYS / NEVER / MAYBE over the cell (w_i = (q - c_i) . m_i - chw_i |q - c_i|,
//       Lipschitz |m_i - chw_i u| + |chw_i| rho / (|c - c_i| - rho)); corner points wherever the point can be invalid.
//   A target k can be dropped when a target a that is available all over the cell (ALWAYS circle; corner point only
//       in an all-invalid cell) beats it by more than the tie band everywhere: d_k - d_a > tau + lip rho, the
//       difference of two distances varying by at most rho (|g_k - g_a| + turn_k + turn_a) over the cell.
//   What remains must be one or two targets; the per-point code (lrm_tol_plane_tab) ranks those two, runs the arc test
//   of a MAYBE circle and the validity of the OPEN circle with the doubt bands of the full evaluation.
// Where find_region's rays cross the cell every region in reach must give the same rows.  Everything in double, from the
// float tables the device reads.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "lrm_compile.h"
#include "lrm_toltab_build.h"

// Host builder.  The per-cell arithmetic lives in lrm_toltab_build.h, shared with the device builder (lrm_toltab_dev.hip): both
// produce the same bytes.  Rows of cells are classified on a few host threads (the cells are independent); nothing in the result
// depends on the thread count or on the order in which cells are visited (row numbers = ranks of the canonical row ids in use).

namespace {

struct GridCells {
    std::vector<LrmTbCell> coarse;            // N * N
    std::vector<std::vector<LrmTbCell>> fine; // per refined cell: SUB * SUB
    std::vector<int> fine_of;                 // N * N: index into fine, or -1
};

void classify_grid(const LrmTbInput& in, int g, double H, int threads, GridCells* out) {
    constexpr int N = LRM_TT_N, kSub = LRM_TT_SUB;
    out->coarse.assign((size_t)N * N, LrmTbCell{});
    out->fine_of.assign((size_t)N * N, -1);
    std::vector<std::vector<LrmTbCell>> fine_rows((size_t)N * N);
    auto work = [&](int t) {
        for (int iz = t; iz < N; iz += threads)
            for (int ix = 0; ix < N; ix++) {
                LrmTbCell c = lrm_tb_coarse_cell(in, g, H, ix, iz);
                out->coarse[(size_t)iz * N + ix] = c;
                if (c.ok) continue;
                std::vector<LrmTbCell> sub((size_t)kSub * kSub);
                bool any = false;
                double lb = 1.0e30; // the sub-cells' bounds are tighter than the coarse cell's own: their minimum holds for the whole cell
                for (int sz = 0; sz < kSub; sz++)
                    for (int sx = 0; sx < kSub; sx++) {
                        const LrmTbCell f = lrm_tb_fine_cell(in, g, H, ix, iz, sx, sz);
                        sub[(size_t)sz * kSub + sx] = f;
                        any = any || f.ok;
                        lb = lrm_tb_min(lb, f.lb);
                    }
                out->coarse[(size_t)iz * N + ix].lb = lrm_tb_max(lb, c.lb);
                if (any) fine_rows[(size_t)iz * N + ix] = std::move(sub);
            }
    };
    if (threads <= 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (size_t i = 0; i < fine_rows.size(); i++)
        if (!fine_rows[i].empty()) {
            out->fine_of[i] = (int)out->fine.size();
            out->fine.push_back(std::move(fine_rows[i]));
        }
}

} // namespace

// The table's row numbers from the canonical ids in use (bit i of used_rows / used_vrows): row 0 = none, validity rows 0 / 1 =
// false / true, then the ids in use in ascending order.  false: more rows than a cell code can name.
bool lrm_tb_number_rows(const LrmTbInput& in, uint64_t used_rows, uint32_t used_vrows, uint8_t* row_num, uint8_t* vrow_num, LrmTolTabHeader* hd) {
    const LrmTabRow none{0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 2.f, 0.f};
    const LrmTabVRow vfalse{0.f, 0.f, 0.f, 1.0e30f};
    for (int i = 0; i < 32; i++) { hd->rows[i] = none; hd->vrows[i] = vfalse; }
    uint32_t nr = 1, nv = 2;
    bool ok = true;
    std::memset(row_num, 0, LRM_TB_ROWS);
    std::memset(vrow_num, 0, LRM_TB_VROWS);
    for (int i = 1; i < LRM_TB_ROWS; i++)
        if (used_rows & (1ull << i)) {
            if (nr > LRM_TT_MAX_ROWS - 1) { ok = false; break; }
            row_num[i] = (uint8_t)nr;
            hd->rows[nr++] = in.rows[i];
        }
    hd->vrows[0] = in.vrows[0];
    hd->vrows[1] = in.vrows[1];
    vrow_num[1] = 1;
    for (int i = 2; i < LRM_TB_VROWS && ok; i++)
        if (used_vrows & (1u << i)) {
            if (nv > LRM_TT_MAX_ROWS - 1) { ok = false; break; }
            vrow_num[i] = (uint8_t)nv;
            hd->vrows[nv++] = in.vrows[i];
        }
    hd->n_rows = nr;
    hd->n_vrows = nv;
    // LRM_TT_UNANSWERED names row 31 three times: its validity is nan, which no test passes (lrm_tol_plane_tab: doubt)
    hd->vrows[31] = LrmTabVRow{0.f, 0.f, 0.f, std::nanf("")};
    return ok;
}

// Offsets of the table's arrays (uint16 units behind the header) for n_fine refined cells per grid; returns the number of uint16
void lrm_tb_layout(const uint32_t n_fine[2], LrmTolTabHeader* hd, size_t* n_cells) {
    constexpr size_t N = LRM_TT_N, kBlk = (size_t)LRM_TT_SUB * LRM_TT_SUB;
    const double Hs[2] = {LRM_TT_H_INNER, LRM_TT_H_OUTER};
    size_t at = 0;
    for (int g = 0; g < 2; g++) {
        hd->coarse_off[g] = (uint32_t)at;
        at += N * N;
        hd->fine_off[g] = (uint32_t)at;
        hd->n_fine[g] = n_fine[g];
        at += kBlk * (n_fine[g] ? n_fine[g] : 1); // (the look-up reads block 0 for unrefined cells: a spare block where there is none)
        hd->inv_h[g] = (float)(1.0 / Hs[g]);
    }
    hd->lb_unit = (float)(2.0 * Hs[0] / 16.0 / 64.0);
    at = (at + 7) & ~(size_t)7; // the kernels stage the bounds with 16-byte loads
    hd->bound_off[0] = hd->bound_off[1] = (uint32_t)at;
    at += 2 * (size_t)LRM_TT_NB * LRM_TT_NB;
    *n_cells = at;
}
static_assert(sizeof(LrmTolTabHeader) % 16 == 0, "the cell arrays and the bounds behind the header are read with 16-byte loads");

// -> false when the leg needs more distinct rows than a cell code can name (the caller then uses the kernels without a table)
bool lrm_build_tol_tab(const LrmTolLeg& L, std::vector<uint8_t>* out) {
    LrmTbInput in;
    lrm_tb_make_input(L, &in);
    constexpr int N = LRM_TT_N, kSub = LRM_TT_SUB, NB = LRM_TT_NB, S = LRM_TB_S;
    int threads = (int)std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : (threads > 8 ? 8 : threads);
    if (const char* e = std::getenv("LRM_TOLTAB_THREADS")) threads = std::max(1, std::atoi(e));
    const double Hs[2] = {LRM_TT_H_INNER, LRM_TT_H_OUTER};
    GridCells grids[2];
    for (int g = 0; g < 2; g++) classify_grid(in, g, Hs[g], threads, &grids[g]);
    // the rows in use: every cell that gets a code of its own
    uint64_t used_rows = 1;
    uint32_t used_vrows = 3;
    auto mark = [&](const LrmTbCell& c) {
        if (!c.ok) return;
        used_rows |= (1ull << c.t[0]) | (1ull << c.t[1]);
        used_vrows |= 1u << c.v;
    };
    size_t n_un[2] = {0, 0};
    uint32_t n_fine[2];
    for (int g = 0; g < 2; g++) {
        if (grids[g].fine.size() > 0x7fff) return false;
        n_fine[g] = (uint32_t)grids[g].fine.size();
        for (size_t i = 0; i < (size_t)N * N; i++)
            if (grids[g].fine_of[i] < 0) { mark(grids[g].coarse[i]); n_un[g] += !grids[g].coarse[i].ok; }
        for (const auto& blk : grids[g].fine)
            for (const LrmTbCell& c : blk) mark(c);
    }
    LrmTolTabHeader hd;
    std::memset(&hd, 0, sizeof hd);
    uint8_t row_num[LRM_TB_ROWS], vrow_num[LRM_TB_VROWS];
    if (!lrm_tb_number_rows(in, used_rows, used_vrows, row_num, vrow_num, &hd)) return false;
    size_t n_cells = 0;
    lrm_tb_layout(n_fine, &hd, &n_cells);
    std::vector<uint16_t> cells(n_cells, 0);
    auto code_of = [&](const LrmTbCell& c) -> uint16_t {
        if (!c.ok) return (uint16_t)LRM_TT_UNANSWERED;
        return (uint16_t)((unsigned)row_num[c.t[0]] | ((unsigned)row_num[c.t[1]] << 5) | ((unsigned)vrow_num[c.v] << 10));
    };
    for (int g = 0; g < 2; g++) {
        const GridCells& G = grids[g];
        for (size_t i = 0; i < (size_t)N * N; i++)
            cells[hd.coarse_off[g] + i] = G.fine_of[i] >= 0 ? (uint16_t)(0x8000u | (unsigned)G.fine_of[i]) : code_of(G.coarse[i]);
        size_t at = hd.fine_off[g];
        for (const auto& blk : G.fine)
            for (const LrmTbCell& c : blk) cells[at++] = code_of(c);
        if (G.fine.empty())
            for (int k = 0; k < kSub * kSub; k++) cells[at++] = (uint16_t)LRM_TT_UNANSWERED;
    }
    {   // 32-bit bounds of the inner grid (the outer grid's lanes use the outer circle)
        const GridCells& G = grids[0];
        auto work = [&](int t) {
            for (int bz = t; bz < NB; bz += threads)
                for (int bx = 0; bx < NB; bx++) {
                    double lb[S * S];
                    for (int sz = 0; sz < S; sz++)
                        for (int sx = 0; sx < S; sx++) {
                            const int ix = 2 * bx + sx / (S / 2), iz = 2 * bz + sz / (S / 2);
                            const size_t ci = (size_t)iz * N + ix;
                            lb[sz * S + sx] = lrm_tb_subcell_lb(in, Hs[0], bx, bz, sx, sz, G.coarse[ci], G.fine_of[ci] >= 0 ? G.fine[(size_t)G.fine_of[ci]].data() : nullptr);
                        }
                    const uint32_t w = lrm_tb_bound_word(lrm_tb_fit_bound(lb, Hs[0]));
                    const size_t at = hd.bound_off[0] + 2 * ((size_t)bz * NB + bx);
                    cells[at] = (uint16_t)(w & 0xffffu);
                    cells[at + 1] = (uint16_t)(w >> 16);
                }
        };
        if (threads <= 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
            for (auto& th : pool) th.join();
        }
    }
    hd.band_max = (float)in.band[0];
    hd.band_max_outer = (float)in.band[1];
    // both plane points of a point lie within max(r + coxa_length, |z|) of the femur joint (|u| <= r)
    hd.far_limit = (float)(0.5 * N * Hs[0] - 1.0);
    out->resize(sizeof hd + cells.size() * 2);
    std::memcpy(out->data(), &hd, sizeof hd);
    std::memcpy(out->data() + sizeof hd, cells.data(), cells.size() * 2);
    if (std::getenv("LRM_TOL_DEBUG"))
        std::fprintf(stderr, "tol tab: %u rows, %u validity rows; inner grid %u refined, %zu coarse cells unanswered; outer grid %u refined, %zu unanswered; %zu bytes\n",
                     hd.n_rows, hd.n_vrows, hd.n_fine[0], n_un[0], hd.n_fine[1], n_un[1], out->size());
    return true;
}

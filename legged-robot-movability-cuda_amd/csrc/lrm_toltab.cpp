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
#include "lrm_point_tol.h"

namespace {

thread_local int g_reason = 0; // why the last cell went unanswered (statistics): 1 regions differ, 2 two open validities, 3 a centre nearby, 4 more than two targets, 5 none
struct Rows {
    std::vector<LrmTabRow> rows;
    std::vector<LrmTabVRow> vrows;
    int row(const LrmTabRow& r) {
        for (size_t i = 0; i < rows.size(); i++)
            if (std::memcmp(&rows[i], &r, sizeof r) == 0) return (int)i;
        rows.push_back(r);
        return (int)rows.size() - 1;
    }
    int vrow(const LrmTabVRow& r) {
        for (size_t i = 0; i < vrows.size(); i++)
            if (std::memcmp(&vrows[i], &r, sizeof r) == 0) return (int)i;
        vrows.push_back(r);
        return (int)vrows.size() - 1;
    }
};

struct CellCode {
    bool ok = false;
    int n = 0;
    LrmTabRow t[2];
    LrmTabVRow v;
    double lb = 0.0; // lower bound of sqrt(du^2 + dz^2) over the cell (set whether or not the cell has an answer)
    bool all_invalid = false; // every point of the cell is invalid, whatever region list it is evaluated with
};

constexpr LrmTabRow kNoneRow{0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 2.f, 0.f};
constexpr LrmTabVRow kFalseRow{0.f, 0.f, 0.f, 1.0e30f}, kTrueRow{0.f, 0.f, 0.f, -1.0e30f};

CellCode classify_reg(const LrmTolLeg& L, unsigned reg, double cx, double cz, double rho, double band, double tau) {
    CellCode out;
    const LrmTolLeg::Circle* ct = &L.circ[reg][0];
    double mag[LRM_N_CIRCLES];
    int n_open = 0, open_j = -1;
    bool one_out = false;
    for (int j = 0; j < LRM_N_CIRCLES; j++) {
        const double vx = cx - ct[j].x, vy = cz - ct[j].y;
        mag[j] = std::hypot(vx, vy);
        const double v = (vx * vx + vy * vy) * (double)ct[j].gs + (double)ct[j].c;
        const double lip = 2.0 * std::fabs((double)ct[j].gs) * (mag[j] + rho) * rho;
        if (v - lip - band > 0) one_out = true;
        else if (!(v + lip + band < 0)) { n_open++; open_j = j; }
    }
    int vstate; // 0: invalid all over the cell, 1: valid all over it, 2: one circle decides
    if (one_out) vstate = 0;
    else if (n_open == 0) vstate = 1;
    else if (n_open == 1) vstate = 2;
    else { g_reason = 2; return out; }
    struct Cand { double d, gx, gy, turn; LrmTabRow row; bool flips, always; } cand[LRM_N_CIRCLES + LRM_N_CORNERS];
    int nc = 0;
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        if (!(mag[i] > 2.0 * rho)) { g_reason = 3; return out; } // the centre of a circle in or next to the cell: directions turn freely
        const double vx = cx - ct[i].x, vy = cz - ct[i].y;
        const double ux = vx / mag[i], uy = vy / mag[i];
        const double w = vx * (double)ct[i].mx + vy * (double)ct[i].my - (double)ct[i].chw * mag[i];
        bool ok, maybe = false;
        if (std::fabs((double)ct[i].chw) > 1.0) ok = ct[i].chw < 0; // always / never
        else {
            const double gx = (double)ct[i].mx - (double)ct[i].chw * ux, gy = (double)ct[i].my - (double)ct[i].chw * uy;
            const double lip = std::hypot(gx, gy) + std::fabs((double)ct[i].chw) * rho / (mag[i] - rho);
            maybe = !(std::fabs(w) - (double)ct[i].bw * (mag[i] + rho) > tau + lip * rho);
            ok = w >= 0;
        }
        if (!ok && !maybe) continue;
        const double s = (mag[i] >= ct[i].r) ? 1.0 : -1.0;
        LrmTabRow row{ct[i].x, ct[i].y, ct[i].r, 0.f, ct[i].mx, ct[i].my, ct[i].chw, ct[i].bw};
        if (!maybe) { row.mx = 1.f; row.my = 0.f; row.chw = -2.f; row.bw = 0.f; } // valid all over the cell: no arc test
        cand[nc++] = Cand{std::fabs((double)ct[i].r - mag[i]), s * ux, s * uy, rho / (mag[i] - rho), row,
                          std::fabs((double)ct[i].r - mag[i]) <= rho, !maybe};
    }
    if (vstate != 1)
        for (int i = 0; i < L.n_corners; i++) {
            const LrmCircle& f = L.feat[4 * LRM_N_CIRCLES + i];
            const double vx = cx - f.x, vy = cz - f.y, d = std::hypot(vx, vy);
            if (!(d > 2.0 * rho)) { g_reason = 3; return out; }
            cand[nc++] = Cand{d, vx / d, vy / d, rho / (d - rho), LrmTabRow{f.x, f.y, 0.f, 3.0e38f, 1.f, 0.f, -2.f, 0.f}, false, vstate == 0};
        }
    if (nc == 0) { g_reason = 5; return out; }
    bool excl[LRM_N_CIRCLES + LRM_N_CORNERS] = {false};
    for (int k = 0; k < nc; k++)
        for (int a = 0; a < nc && !excl[k]; a++) {
            if (a == k || !cand[a].always) continue;
            double lip = std::hypot(cand[k].gx - cand[a].gx, cand[k].gy - cand[a].gy) + cand[k].turn + cand[a].turn;
            if (cand[k].flips || cand[a].flips || lip > 2.0) lip = 2.0; // a circle crossing the cell: the sign of its gradient is open
            // + the 4 mantissa bits the full evaluation's ranking keys drop and its relative tie allowance
            if (cand[k].d - cand[a].d > tau + lip * rho + 8.0e-6 * (cand[k].d + 1.0)) excl[k] = true;
        }
    for (int k = 0; k < nc; k++)
        if (!excl[k]) {
            if (out.n >= 2) { out.n = 0; g_reason = 4; return out; }
            out.t[out.n++] = cand[k].row;
        }
    if (out.n == 0) { g_reason = 5; return out; }
    if (out.n == 1) out.t[1] = kNoneRow;
    // Lower bound of the distance to the chosen target: the choice is one of the survivors, each distance 1-Lipschitz.  A
    // point that may be valid gets 0 (a candidate on a yaw-limit plane then collapses to its offset).  When no survivor is
    // available all over the cell the evaluation may find no target at all and return the raw point (one_leg.cu:141-142).
    {
        double m = 1.0e30;
        bool any_always = false;
        for (int k = 0; k < nc; k++)
            if (!excl[k]) {
                m = std::min(m, cand[k].d);
                any_always = any_always || cand[k].always;
            }
        if (!any_always) m = std::min(m, std::hypot(cx, cz));
        out.lb = vstate == 0 ? std::max(0.0, m - rho) : 0.0;
        out.all_invalid = vstate == 0;
    }
    out.v = vstate == 0 ? kFalseRow : (vstate == 1 ? kTrueRow : LrmTabVRow{ct[open_j].x, ct[open_j].y, ct[open_j].gs, ct[open_j].c});
    out.ok = true;
    return out;
}

bool same_row(const LrmTabRow& a, const LrmTabRow& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// The bound of a cell without an answer, for the region lists `regs` (bit set): every circle whose clamp point is not
// proven invalid all over the cell, every corner point, the raw point; 0 where a point may be valid.
double generic_lb(const LrmTolLeg& L, unsigned regs, double cx, double cz, double rho, double band, double tau) {
    double lb = std::hypot(cx, cz);
    bool maybe_valid = false;
    for (unsigned reg = 0; reg < 4; reg++) {
        if (!(regs & (1u << reg))) continue;
        const LrmTolLeg::Circle* ct = &L.circ[reg][0];
        bool one_out = false;
        for (int j = 0; j < LRM_N_CIRCLES; j++) {
            const double vx = cx - ct[j].x, vy = cz - ct[j].y, mag = std::hypot(vx, vy);
            bool never = false;
            if (mag > 2.0 * rho) {
                if (std::fabs((double)ct[j].chw) > 1.0) never = !(ct[j].chw < 0);
                else {
                    const double ux = vx / mag, uy = vy / mag;
                    const double w = vx * (double)ct[j].mx + vy * (double)ct[j].my - (double)ct[j].chw * mag;
                    const double gx = (double)ct[j].mx - (double)ct[j].chw * ux, gy = (double)ct[j].my - (double)ct[j].chw * uy;
                    const double lip = std::hypot(gx, gy) + std::fabs((double)ct[j].chw) * rho / (mag - rho);
                    never = !(w >= 0) && (std::fabs(w) - (double)ct[j].bw * (mag + rho) > tau + lip * rho);
                }
            }
            if (!never) lb = std::min(lb, std::fabs((double)ct[j].r - mag));
            const double v = (vx * vx + vy * vy) * (double)ct[j].gs + (double)ct[j].c;
            const double lip = 2.0 * std::fabs((double)ct[j].gs) * (mag + rho) * rho;
            if (v - lip - band > 0) one_out = true;
        }
        if (!one_out) maybe_valid = true;
    }
    for (int i = 0; i < L.n_corners; i++) {
        const LrmCircle& f = L.feat[4 * LRM_N_CIRCLES + i];
        lb = std::min(lb, std::hypot(cx - f.x, cz - f.y));
    }
    return maybe_valid ? 0.0 : std::max(0.0, lb - rho);
}

// what the cell's points may be evaluated with, or !ok.  Where find_region's rays cross the cell, every region in
// reach must give the same rows.
CellCode classify_cell(const LrmTolLeg& L, double cx, double cz, double rho, double band, double tau) {
    double v[4];
    for (int i = 0; i < 3; i++) v[i] = (double)L.dir_cos[i] * cz - (double)L.dir_sin[i] * cx;
    v[3] = cz; // the atan2f wrap ray (x < 0, z = +-0) is the sign of z itself
    unsigned open_bits = 0, base = 0;
    for (int i = 0; i < 4; i++) {
        if (!(std::fabs(v[i]) > band + rho)) open_bits |= 1u << i;
        if (v[i] < 0) base |= 1u << i;
    }
    CellCode code;
    unsigned seen = 0, reach = 0;
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // the region lists in reach
        reach |= 1u << ((L.region_lut >> (((base & ~open_bits) | sub) << 1)) & 3u);
        if (sub == 0) break;
    }
    const double lb_generic = generic_lb(L, reach, cx, cz, rho, band, tau);
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // every assignment of the open signs
        const unsigned pat = (base & ~open_bits) | sub;
        const unsigned reg = (L.region_lut >> (pat << 1)) & 3u;
        if (!(seen & (1u << reg))) {
            CellCode c = classify_reg(L, reg, cx, cz, rho, band, tau);
            if (!c.ok) {
                c.lb = lb_generic;
                return c;
            }
            if (!seen) code = c;
            else {
                code.lb = std::min(code.lb, c.lb);
                code.all_invalid = code.all_invalid && c.all_invalid;
                bool same = c.n == code.n && std::memcmp(&c.v, &code.v, sizeof c.v) == 0;
                if (same && c.n == 1) same = same_row(c.t[0], code.t[0]);
                if (same && c.n == 2)
                    same = (same_row(c.t[0], code.t[0]) && same_row(c.t[1], code.t[1])) ||
                           (same_row(c.t[0], code.t[1]) && same_row(c.t[1], code.t[0]));
                if (!same) {
                    g_reason = 1;
                    code.ok = false;
                    code.lb = lb_generic;
                    return code;
                }
            }
            seen |= 1u << reg;
        }
        if (sub == 0) break;
    }
    return code;
}

// Lower bound of the distance to the target the evaluation picks at a point within rho of (cx, cz), for a cell with an answer
// whose points are all invalid: the choice is one of the cell's targets (or, when none of them is available all over the cell,
// possibly none: the raw point, one_leg.cu:141-142), each distance 1-Lipschitz.
double survivor_lb(const CellCode& c, double cx, double cz, double rho) {
    double m = 1.0e30;
    bool any_always = false;
    for (int k = 0; k < c.n; k++) {
        m = std::min(m, std::fabs((double)c.t[k].r - std::hypot(cx - c.t[k].x, cz - c.t[k].y)));
        any_always = any_always || c.t[k].chw == -2.f;
    }
    if (!any_always) m = std::min(m, std::hypot(cx, cz));
    return std::max(0.0, m - rho);
}

// One grid: N x N cells of H mm around the femur joint, unanswered cells refined into LRM_TT_SUB^2 sub-cells.
// Rows of cells are classified on `threads` host threads (the cells are independent); row numbers are given
// afterwards, serially, so that the table does not depend on the thread count.
struct GridCells {
    std::vector<CellCode> coarse;             // N * N
    std::vector<std::vector<CellCode>> fine;  // per refined cell: SUB * SUB
    std::vector<int> fine_of;                 // N * N: index into fine, or -1
};
void classify_grid(const LrmTolLeg& L, double H, double band, double tau, int threads, GridCells* out) {
    constexpr int N = LRM_TT_N, kSub = LRM_TT_SUB;
    const double half = 0.5 * N * H, h = H / kSub;
    // The per-point code finds its cell from float arithmetic (one FMA + floor: off by at most 2^-17 of a cell) on a plane
    // point that itself carries a few 1e-5 mm of rounding: the classification holds a margin around the cell.
    const double slack = H * 1.6e-5 + 1.0e-3;
    out->coarse.assign((size_t)N * N, CellCode());
    out->fine_of.assign((size_t)N * N, -1);
    std::vector<std::vector<CellCode>> fine_rows((size_t)N * N);
    auto work = [&](int t) {
        for (int iz = t; iz < N; iz += threads)
            for (int ix = 0; ix < N; ix++) {
                const double x0 = -half + ix * H, z0 = -half + iz * H;
                CellCode c = classify_cell(L, x0 + 0.5 * H, z0 + 0.5 * H, 0.5 * H * 1.41421357 + slack, band, tau);
                out->coarse[(size_t)iz * N + ix] = c;
                if (c.ok) continue;
                std::vector<CellCode> sub((size_t)kSub * kSub);
                bool any = false;
                double lb = 1.0e30; // the sub-cells' bounds are tighter than the coarse cell's own: their minimum holds for the whole cell
                for (int sz = 0; sz < kSub; sz++)
                    for (int sx = 0; sx < kSub; sx++) {
                        sub[(size_t)sz * kSub + sx] = classify_cell(L, x0 + (sx + 0.5) * h, z0 + (sz + 0.5) * h, 0.5 * h * 1.41421357 + slack, band, tau);
                        any = any || sub[(size_t)sz * kSub + sx].ok;
                        lb = std::min(lb, sub[(size_t)sz * kSub + sx].lb);
                    }
                out->coarse[(size_t)iz * N + ix].lb = std::max(lb, c.lb);
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

// The bounds of one grid: LRM_TT_NB^2 bound cells of 2 H, each a plane over its 16 x 16 sub-cells (size hs = H / 8),
//     lb(sx, sz) = d0 + unit (gx sx + gz sz),  unit = hs / 64 mm, gx, gz integers of 8 bits.
// A sub-cell's own bound comes from the coarse cell it lies in: the distance to that cell's targets at the sub-cell's centre
// minus its half-diagonal (cells with an answer and no valid point), the minimum over the fine sub-cells it covers (refined
// cells), else the coarse cell's constant.  The plane: least-squares gradient, quantised, then lowered until no sub-cell's
// bound lies below it -- nothing to prove about curvature.
struct BoundEntry {
    double d0;
    int gx, gz;
};
void build_bounds(const GridCells& G, double H, int threads, std::vector<BoundEntry>* out) {
    constexpr int N = LRM_TT_N, NB = LRM_TT_NB, kSub = LRM_TT_SUB, S = 16;
    static_assert(N == 2 * NB, "a bound cell is 2 x 2 coarse cells");
    const double half = 0.5 * N * H, hs = 2.0 * H / S, slack = H * 1.6e-5 + 1.0e-3, rho = 0.5 * hs * 1.41421357 + slack, unit = hs / 64.0;
    out->assign((size_t)NB * NB, BoundEntry{0.0, 0, 0});
    auto work = [&](int t) {
        for (int bz = t; bz < NB; bz += threads)
            for (int bx = 0; bx < NB; bx++) {
                double lb[S][S];
                for (int sz = 0; sz < S; sz++)
                    for (int sx = 0; sx < S; sx++) {
                        const double cx = -half + bx * 2.0 * H + (sx + 0.5) * hs, cz = -half + bz * 2.0 * H + (sz + 0.5) * hs;
                        const int ix = 2 * bx + sx / (S / 2), iz = 2 * bz + sz / (S / 2);
                        const size_t ci = (size_t)iz * N + ix;
                        const CellCode& c = G.coarse[ci];
                        double v;
                        if (c.ok) v = c.all_invalid ? survivor_lb(c, cx, cz, rho) : 0.0;
                        else if (G.fine_of[ci] >= 0) { // the 2 x 2 fine sub-cells this sub-cell covers
                            const std::vector<CellCode>& f = G.fine[(size_t)G.fine_of[ci]];
                            const int fx = (sx % (S / 2)) * (kSub / (S / 2)), fz = (sz % (S / 2)) * (kSub / (S / 2));
                            v = 1.0e30;
                            for (int a = 0; a < kSub / (S / 2); a++)
                                for (int b = 0; b < kSub / (S / 2); b++) v = std::min(v, f[(size_t)(fz + a) * kSub + fx + b].lb);
                        } else v = c.lb;
                        lb[sz][sx] = v;
                    }
                double mean = 0, gx = 0, gz = 0;
                for (int sz = 0; sz < S; sz++)
                    for (int sx = 0; sx < S; sx++) mean += lb[sz][sx];
                mean /= S * S;
                const double m = 0.5 * (S - 1), var = S * (S * S - 1.0) / 12.0 * S; // sum over the grid of (s - m)^2
                for (int sz = 0; sz < S; sz++)
                    for (int sx = 0; sx < S; sx++) {
                        gx += (sx - m) * (lb[sz][sx] - mean);
                        gz += (sz - m) * (lb[sz][sx] - mean);
                    }
                BoundEntry e;
                e.gx = (int)std::lround(std::max(-127.0, std::min(127.0, gx / var / unit)));
                e.gz = (int)std::lround(std::max(-127.0, std::min(127.0, gz / var / unit)));
                e.d0 = 1.0e30;
                for (int sz = 0; sz < S; sz++)
                    for (int sx = 0; sx < S; sx++) e.d0 = std::min(e.d0, lb[sz][sx] - unit * (e.gx * sx + e.gz * sz));
                (*out)[(size_t)bz * NB + bx] = e;
            }
    };
    if (threads <= 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
}

// IEEE half (bits) of a bound, rounded DOWN (towards -inf); normal halves and zero only: a positive value below the smallest
// normal half becomes 0, a negative one above its negative becomes that
uint16_t half_floor(double v) {
    const bool neg = v < 0;
    double a = std::fabs(v);
    if (!(a >= 6.103515625e-5)) return neg ? 0x8400 : 0;
    if (a >= 65504.0) a = 65504.0; // (a bound that large does not occur: the grids end at 8192 mm)
    int e;
    const double m = std::frexp(a, &e); // a = m 2^e, m in [0.5, 1)
    const double f = (2.0 * m - 1.0) * 1024.0; // 10 mantissa bits
    unsigned frac = (unsigned)(neg ? std::ceil(f) : std::floor(f)), ex = (unsigned)(e - 1 + 15);
    if (frac > 1023u) { frac = 0; ex++; }
    if (ex > 30u) { ex = 30u; frac = 1023u; }
    return (uint16_t)((neg ? 0x8000u : 0u) | (ex << 10) | frac);
}

} // namespace

// -> false when the leg needs more distinct rows than a cell code can name (the caller then uses the kernels without a table)
bool lrm_build_tol_tab(const LrmTolLeg& L, std::vector<uint8_t>* out) {
    // the largest decision bands the grids are built for: points up to |p|_1 = 4096 mm on the inner grid (whatever lies further
    // out is beyond it), 16384 mm on the outer one (its cells are 8 times as large: the band stays the same fraction of a sub-cell)
    const double bands[2] = {(double)L.band_base + (double)L.band_slope * 4096.0, (double)L.band_base + (double)L.band_slope * 16384.0};
    constexpr int N = LRM_TT_N, kSub = LRM_TT_SUB;
    int threads = (int)std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : (threads > 8 ? 8 : threads);
    if (const char* e = std::getenv("LRM_TOLTAB_THREADS")) threads = std::max(1, std::atoi(e));
    const double Hs[2] = {LRM_TT_H_INNER, LRM_TT_H_OUTER};
    GridCells grids[2];
    for (int g = 0; g < 2; g++) classify_grid(L, Hs[g], bands[g], bands[g] * (double)LRM_TOL_TIE, threads, &grids[g]);
    Rows R;
    R.row(kNoneRow);   // row 0
    R.vrow(kFalseRow); // validity rows 0, 1
    R.vrow(kTrueRow);
    bool rows_ok = true;
    auto code_of = [&](const CellCode& c) -> uint16_t {
        if (!c.ok) return (uint16_t)LRM_TT_UNANSWERED;
        const int a = R.row(c.t[0]), b = R.row(c.t[1]), vr = R.vrow(c.v);
        if (a > LRM_TT_MAX_ROWS - 1 || b > LRM_TT_MAX_ROWS - 1 || vr > LRM_TT_MAX_ROWS - 1) rows_ok = false;
        return (uint16_t)(((unsigned)a & 31u) | (((unsigned)b & 31u) << 5) | (((unsigned)vr & 31u) << 10));
    };
    LrmTolTabHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::vector<uint16_t> cells;
    size_t n_un[2] = {0, 0};
    for (int g = 0; g < 2; g++) {
        const GridCells& G = grids[g];
        if (G.fine.size() > 0x7fff) return false;
        hd.coarse_off[g] = (uint32_t)cells.size();
        for (size_t i = 0; i < (size_t)N * N; i++) {
            if (G.fine_of[i] >= 0) cells.push_back((uint16_t)(0x8000u | (unsigned)G.fine_of[i]));
            else {
                cells.push_back(code_of(G.coarse[i]));
                n_un[g] += !G.coarse[i].ok;
            }
        }
        hd.fine_off[g] = (uint32_t)cells.size();
        hd.n_fine[g] = (uint32_t)G.fine.size();
        for (const auto& blk : G.fine)
            for (const CellCode& c : blk) cells.push_back(code_of(c));
        if (G.fine.empty()) cells.insert(cells.end(), (size_t)kSub * kSub, (uint16_t)LRM_TT_UNANSWERED); // the lookup reads block 0 for unrefined cells
        hd.inv_h[g] = (float)(1.0 / Hs[g]);
        if (g == 0) hd.lb_unit = (float)(2.0 * Hs[g] / 16.0 / 64.0);
    }
    if (!rows_ok) return false;
    { // 32-bit bounds of the inner grid, little endian: d0, then the gradient bytes (the outer grid's lanes use the outer circle)
        std::vector<BoundEntry> bounds;
        build_bounds(grids[0], Hs[0], threads, &bounds);
        if (cells.size() & 1u) cells.push_back(0);
        hd.bound_off[0] = hd.bound_off[1] = (uint32_t)cells.size();
        for (const BoundEntry& e : bounds) {
            cells.push_back(half_floor(e.d0));
            cells.push_back((uint16_t)(((unsigned)e.gx & 0xffu) | (((unsigned)e.gz & 0xffu) << 8)));
        }
    }
    hd.band_max = (float)bands[0];
    hd.band_max_outer = (float)bands[1];
    // both plane points of a point lie within max(r + coxa_length, |z|) of the femur joint (|u| <= r)
    hd.far_limit = (float)(0.5 * N * Hs[0] - 1.0);
    hd.n_rows = (uint32_t)R.rows.size();
    hd.n_vrows = (uint32_t)R.vrows.size();
    for (size_t i = 0; i < 32; i++) hd.rows[i] = i < R.rows.size() ? R.rows[i] : kNoneRow;
    for (size_t i = 0; i < 32; i++) hd.vrows[i] = i < R.vrows.size() ? R.vrows[i] : kFalseRow;
    // LRM_TT_UNANSWERED names row 31 three times: its validity is nan, which no test passes (lrm_tol_plane_tab: doubt)
    hd.vrows[31] = LrmTabVRow{0.f, 0.f, 0.f, std::nanf("")};
    out->resize(sizeof hd + cells.size() * 2);
    std::memcpy(out->data(), &hd, sizeof hd);
    std::memcpy(out->data() + sizeof hd, cells.data(), cells.size() * 2);
    if (std::getenv("LRM_TOL_DEBUG"))
        std::fprintf(stderr, "tol tab: %zu rows, %zu validity rows; inner grid %u refined, %zu coarse cells unanswered; outer grid %u refined, %zu unanswered; %zu bytes\n",
                     R.rows.size(), R.vrows.size(), hd.n_fine[0], n_un[0], hd.n_fine[1], n_un[1], out->size());
    return true;
}

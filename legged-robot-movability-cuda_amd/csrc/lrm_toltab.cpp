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
#include <vector>
#include "lrm_compile.h"
#include "lrm_point_tol.h"

namespace {

int g_reason = 0; // why the last cell went unanswered (statistics): 1 regions differ, 2 two open validities, 3 a centre nearby, 4 more than two targets, 5 none
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
    out.v = vstate == 0 ? kFalseRow : (vstate == 1 ? kTrueRow : LrmTabVRow{ct[open_j].x, ct[open_j].y, ct[open_j].gs, ct[open_j].c});
    out.ok = true;
    return out;
}

bool same_row(const LrmTabRow& a, const LrmTabRow& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// code of the cell, or LRM_TT_UNANSWERED
unsigned classify_cell(const LrmTolLeg& L, Rows& R, double cx, double cz, double rho, double band, double tau) {
    double v[4];
    for (int i = 0; i < 3; i++) v[i] = (double)L.dir_cos[i] * cz - (double)L.dir_sin[i] * cx;
    v[3] = cz; // the atan2f wrap ray (x < 0, z = +-0) is the sign of z itself
    unsigned open_bits = 0, base = 0;
    for (int i = 0; i < 4; i++) {
        if (!(std::fabs(v[i]) > band + rho)) open_bits |= 1u << i;
        if (v[i] < 0) base |= 1u << i;
    }
    CellCode code;
    unsigned seen = 0;
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // every assignment of the open signs
        const unsigned pat = (base & ~open_bits) | sub;
        const unsigned reg = (L.region_lut >> (pat << 1)) & 3u;
        if (!(seen & (1u << reg))) {
            const CellCode c = classify_reg(L, reg, cx, cz, rho, band, tau);
            if (!c.ok) return LRM_TT_UNANSWERED;
            if (!seen) code = c;
            else {
                bool same = c.n == code.n && std::memcmp(&c.v, &code.v, sizeof c.v) == 0;
                if (same && c.n == 1) same = same_row(c.t[0], code.t[0]);
                if (same && c.n == 2)
                    same = (same_row(c.t[0], code.t[0]) && same_row(c.t[1], code.t[1])) ||
                           (same_row(c.t[0], code.t[1]) && same_row(c.t[1], code.t[0]));
                if (!same) { g_reason = 1; return LRM_TT_UNANSWERED; }
            }
            seen |= 1u << reg;
        }
        if (sub == 0) break;
    }
    const int a = R.row(code.t[0]), b = R.row(code.t[1]), vr = R.vrow(code.v);
    if (a > LRM_TT_MAX_ROWS - 1 || b > LRM_TT_MAX_ROWS - 1 || vr > LRM_TT_MAX_ROWS - 1) return 0x10000u; // out of rows: no table for this leg
    return (unsigned)a | ((unsigned)b << 5) | ((unsigned)vr << 10);
}

} // namespace

// -> false when the leg needs more distinct rows than a code can name (the caller then uses the kernels without a table)
bool lrm_build_tol_tab(const LrmTolLeg& L, std::vector<uint8_t>* out) {
    // the largest decision band the table is built for: points up to |p|_1 = 4096 mm
    const double band = (double)L.band_base + (double)L.band_slope * 4096.0;
    const double tau = band * (double)LRM_TOL_TIE;
    const double H = LRM_TT_H, h = H / (double)LRM_TT_SUB;
    constexpr int kSub = LRM_TT_SUB;
    Rows R;
    R.row(kNoneRow);   // row 0
    R.vrow(kFalseRow); // vrow 0
    R.vrow(kTrueRow);  // vrow 1
    std::vector<uint16_t> coarse((size_t)LRM_TT_N * LRM_TT_N);
    std::vector<uint16_t> fine;
    size_t n_fine = 0, n_un = 0;
    for (int iz = 0; iz < LRM_TT_N; iz++)
        for (int ix = 0; ix < LRM_TT_N; ix++) {
            const double x0 = -LRM_TT_HALF + ix * H, z0 = -LRM_TT_HALF + iz * H;
            unsigned code = classify_cell(L, R, x0 + 0.5 * H, z0 + 0.5 * H, 0.5 * H * 1.41421357, band, tau);
            if (code == 0x10000u) return false;
            if (code == LRM_TT_UNANSWERED && n_fine < 0x7ffe) {
                uint16_t sub[kSub * kSub];
                bool any = false;
                for (int sz = 0; sz < kSub; sz++)
                    for (int sx = 0; sx < kSub; sx++) {
                        const unsigned c = classify_cell(L, R, x0 + (sx + 0.5) * h, z0 + (sz + 0.5) * h, 0.5 * h * 1.41421357, band, tau);
                        if (c == 0x10000u) return false;
                        sub[sz * kSub + sx] = (uint16_t)c;
                        any = any || c != LRM_TT_UNANSWERED;
                    }
                if (any) {
                    fine.insert(fine.end(), sub, sub + kSub * kSub);
                    coarse[(size_t)iz * LRM_TT_N + ix] = (uint16_t)(0x8000u | (unsigned)n_fine);
                    n_fine++;
                    continue;
                }
            }
            if (code == LRM_TT_UNANSWERED) n_un++;
            coarse[(size_t)iz * LRM_TT_N + ix] = (uint16_t)code;
        }
    LrmTolTabHeader hd;
    std::memset(&hd, 0, sizeof hd);
    hd.n_fine = (uint32_t)n_fine;
    hd.band_max = (float)band;
    hd.n_rows = (uint32_t)R.rows.size();
    hd.n_vrows = (uint32_t)R.vrows.size();
    for (size_t i = 0; i < 32; i++) hd.rows[i] = i < R.rows.size() ? R.rows[i] : kNoneRow;
    for (size_t i = 0; i < 32; i++) hd.vrows[i] = i < R.vrows.size() ? R.vrows[i] : kFalseRow;
    fine.insert(fine.end(), (size_t)kSub * kSub, (uint16_t)LRM_TT_UNANSWERED); // one spare block: the lookup reads fine[0] for unrefined cells
    out->resize(sizeof hd + coarse.size() * 2 + fine.size() * 2);
    std::memcpy(out->data(), &hd, sizeof hd);
    std::memcpy(out->data() + sizeof hd, coarse.data(), coarse.size() * 2);
    std::memcpy(out->data() + sizeof hd + coarse.size() * 2, fine.data(), fine.size() * 2);
    if (std::getenv("LRM_TOL_DEBUG"))
        std::fprintf(stderr, "tol tab: %zu rows, %zu validity rows, %zu refined cells, %zu coarse cells unanswered, %zu bytes\n",
                     R.rows.size(), R.vrows.size(), n_fine, n_un, out->size());
    return true;
}

// lrm_toltab_build.h -- the per-cell arithmetic of the plane table with deferred decisions (LrmTolTabHeader, lrm_types.h),
// ONE source for the host builder (lrm_toltab.cpp) and the device builder (lrm_toltab_dev.hip).
//
// The two builders must produce the same table BYTE FOR BYTE (tests/test_gpu_toltab.py), so everything here is double
// arithmetic made of +, -, *, /, sqrt, fabs, floor / ceil and comparisons only -- each of them correctly rounded on both
// sides, no fused contraction (-ffp-contract=off), no libm function whose last bit may differ (hypot, lround, frexp are
// written out below) -- and every sum that enters a rounding decision is taken in one fixed order.
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
// Where find_region's rays cross the cell every region in reach must give the same rows.
//
// Rows are named by CANONICAL ids while cells are classified (a cell's answer then is six bytes, whoever computes it):
//   target rows    0 none | 1 + 2 (4 reg + i): circle i of list reg with its arc test | 2 + 2 (4 reg + i): the same circle, valid
//                  all over the cell (no arc test) | 33 + i: corner point i
//   validity rows  0 false | 1 true | 2 + 4 reg + j: circle j of list reg decides
// ids with identical contents (the same circle in several lists) are merged (canon_*), and the table's row numbers are the
// ranks of the canonical ids in use -- a numbering that does not depend on the order in which cells are visited.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "lrm_types.h"

#define LRM_TB_ROWS 43
#define LRM_TB_VROWS 18
#define LRM_TB_MAX_CAND (LRM_N_CIRCLES + LRM_N_CORNERS)
#define LRM_TB_RHO_FACTOR 1.41421357 // (a little above sqrt(2) / ... as the first builder wrote it: part of the table's definition)

struct LrmTbCell { // what the classification says about one cell
    double lb;           // lower bound of sqrt(du^2 + dz^2) over the cell (set whether or not the cell has an answer)
    uint8_t ok;          // the cell has an answer: n targets t[], validity row v
    uint8_t n;
    uint8_t all_invalid; // every point of the cell is invalid, whatever region list it is evaluated with
    uint8_t v;
    uint8_t t[2];
    uint8_t pad_[2];
};

struct LrmTbInput { // everything the classification reads: the leg's tolerance block, the rows by canonical id, their merging
    LrmTolLeg L;
    LrmTabRow rows[LRM_TB_ROWS];
    LrmTabVRow vrows[LRM_TB_VROWS];
    uint8_t canon_row[LRM_TB_ROWS + 1]; // (+1: padding to an even size)
    uint8_t canon_vrow[LRM_TB_VROWS];
    double band[2], tau[2]; // the decision band / tie band the inner / outer grid is built for
};

LRM_HD double lrm_tb_min(double a, double b) { return a < b ? a : b; }
LRM_HD double lrm_tb_max(double a, double b) { return a > b ? a : b; }
LRM_HD double lrm_tb_hyp(double x, double y) { return sqrt(x * x + y * y); }

// host: rows by canonical id and their merging
inline void lrm_tb_make_input(const LrmTolLeg& L, LrmTbInput* in) {
    memset(in, 0, sizeof *in);
    in->L = L;
    const LrmTabRow none{0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 2.f, 0.f};
    for (int i = 0; i < LRM_TB_ROWS; i++) in->rows[i] = none;
    for (int reg = 0; reg < 4; reg++)
        for (int i = 0; i < LRM_N_CIRCLES; i++) {
            const LrmTolLeg::Circle& c = L.circ[reg][i];
            in->rows[1 + 2 * (4 * reg + i)] = LrmTabRow{c.x, c.y, c.r, 0.f, c.mx, c.my, c.chw, c.bw};
            in->rows[2 + 2 * (4 * reg + i)] = LrmTabRow{c.x, c.y, c.r, 0.f, 1.f, 0.f, -2.f, 0.f}; // valid all over the cell: no arc test
            in->vrows[2 + 4 * reg + i] = LrmTabVRow{c.x, c.y, c.gs, c.c};
        }
    for (int i = 0; i < LRM_N_CORNERS; i++) {
        const LrmCircle& f = L.feat[4 * LRM_N_CIRCLES + i];
        in->rows[33 + i] = LrmTabRow{f.x, f.y, 0.f, 3.0e38f, 1.f, 0.f, -2.f, 0.f};
    }
    in->vrows[0] = LrmTabVRow{0.f, 0.f, 0.f, 1.0e30f};  // constant false
    in->vrows[1] = LrmTabVRow{0.f, 0.f, 0.f, -1.0e30f}; // constant true
    for (int i = 0; i < LRM_TB_ROWS; i++) {
        in->canon_row[i] = (uint8_t)i;
        for (int j = 0; j < i; j++)
            if (memcmp(&in->rows[j], &in->rows[i], sizeof(LrmTabRow)) == 0) { in->canon_row[i] = (uint8_t)j; break; }
    }
    for (int i = 0; i < LRM_TB_VROWS; i++) {
        in->canon_vrow[i] = (uint8_t)i;
        for (int j = 0; j < i; j++)
            if (memcmp(&in->vrows[j], &in->vrows[i], sizeof(LrmTabVRow)) == 0) { in->canon_vrow[i] = (uint8_t)j; break; }
    }
    // the largest decision bands the grids are built for: points up to |p|_1 = 4096 mm on the inner grid (whatever lies further
    // out is beyond it), 16384 mm on the outer one (its cells are 8 times as large: the band stays the same fraction of a sub-cell)
    in->band[0] = (double)L.band_base + (double)L.band_slope * 4096.0;
    in->band[1] = (double)L.band_base + (double)L.band_slope * 16384.0;
    in->tau[0] = in->band[0] * 0.25; // LRM_TOL_TIE
    in->tau[1] = in->band[1] * 0.25;
}

// one region list at a cell
LRM_HD LrmTbCell lrm_tb_classify_reg(const LrmTbInput& in, unsigned reg, double cx, double cz, double rho, double band, double tau) {
    LrmTbCell out;
    out.lb = 0.0;
    out.ok = 0; out.n = 0; out.all_invalid = 0; out.v = 0; out.t[0] = out.t[1] = 0; out.pad_[0] = out.pad_[1] = 0;
    const LrmTolLeg::Circle* ct = &in.L.circ[reg][0];
    double mag[LRM_N_CIRCLES];
    int n_open = 0, open_j = -1;
    bool one_out = false;
    for (int j = 0; j < LRM_N_CIRCLES; j++) {
        const double vx = cx - (double)ct[j].x, vy = cz - (double)ct[j].y;
        mag[j] = lrm_tb_hyp(vx, vy);
        const double v = (vx * vx + vy * vy) * (double)ct[j].gs + (double)ct[j].c;
        const double lip = 2.0 * fabs((double)ct[j].gs) * (mag[j] + rho) * rho;
        if (v - lip - band > 0) one_out = true;
        else if (!(v + lip + band < 0)) { n_open++; open_j = j; }
    }
    int vstate; // 0: invalid all over the cell, 1: valid all over it, 2: one circle decides
    if (one_out) vstate = 0;
    else if (n_open == 0) vstate = 1;
    else if (n_open == 1) vstate = 2;
    else return out; // two open validities
    // candidates: distance at the centre, gradient, turning rate, row id, the sign of the distance is open, available all over the cell
    double cd[LRM_TB_MAX_CAND], cgx[LRM_TB_MAX_CAND], cgy[LRM_TB_MAX_CAND], cturn[LRM_TB_MAX_CAND];
    uint8_t cid[LRM_TB_MAX_CAND];
    bool cflips[LRM_TB_MAX_CAND], calways[LRM_TB_MAX_CAND];
    int nc = 0;
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        if (!(mag[i] > 2.0 * rho)) return out; // the centre of a circle in or next to the cell: directions turn freely
        const double vx = cx - (double)ct[i].x, vy = cz - (double)ct[i].y;
        const double ux = vx / mag[i], uy = vy / mag[i];
        const double w = vx * (double)ct[i].mx + vy * (double)ct[i].my - (double)ct[i].chw * mag[i];
        bool ok, maybe = false;
        if (fabs((double)ct[i].chw) > 1.0) ok = ct[i].chw < 0; // always / never
        else {
            const double gx = (double)ct[i].mx - (double)ct[i].chw * ux, gy = (double)ct[i].my - (double)ct[i].chw * uy;
            const double lip = lrm_tb_hyp(gx, gy) + fabs((double)ct[i].chw) * rho / (mag[i] - rho);
            maybe = !(fabs(w) - (double)ct[i].bw * (mag[i] + rho) > tau + lip * rho);
            ok = w >= 0;
        }
        if (!ok && !maybe) continue;
        const double s = (mag[i] >= (double)ct[i].r) ? 1.0 : -1.0;
        cd[nc] = fabs((double)ct[i].r - mag[i]);
        cgx[nc] = s * ux;
        cgy[nc] = s * uy;
        cturn[nc] = rho / (mag[i] - rho);
        cid[nc] = (uint8_t)((maybe ? 1 : 2) + 2 * (4 * (int)reg + i));
        cflips[nc] = cd[nc] <= rho;
        calways[nc] = !maybe;
        nc++;
    }
    if (vstate != 1)
        for (int i = 0; i < in.L.n_corners; i++) {
            const LrmCircle& f = in.L.feat[4 * LRM_N_CIRCLES + i];
            const double vx = cx - (double)f.x, vy = cz - (double)f.y, d = lrm_tb_hyp(vx, vy);
            if (!(d > 2.0 * rho)) return out;
            cd[nc] = d;
            cgx[nc] = vx / d;
            cgy[nc] = vy / d;
            cturn[nc] = rho / (d - rho);
            cid[nc] = (uint8_t)(33 + i);
            cflips[nc] = false;
            calways[nc] = vstate == 0;
            nc++;
        }
    if (nc == 0) return out;
    bool excl[LRM_TB_MAX_CAND];
    for (int k = 0; k < nc; k++) excl[k] = false;
    for (int k = 0; k < nc; k++)
        for (int a = 0; a < nc && !excl[k]; a++) {
            if (a == k || !calways[a]) continue;
            double lip = lrm_tb_hyp(cgx[k] - cgx[a], cgy[k] - cgy[a]) + cturn[k] + cturn[a];
            if (cflips[k] || cflips[a] || lip > 2.0) lip = 2.0; // a circle crossing the cell: the sign of its gradient is open
            // + the 4 mantissa bits the full evaluation's ranking keys drop and its relative tie allowance
            if (cd[k] - cd[a] > tau + lip * rho + 8.0e-6 * (cd[k] + 1.0)) excl[k] = true;
        }
    int n = 0;
    uint8_t t0 = 0, t1 = 0;
    for (int k = 0; k < nc; k++)
        if (!excl[k]) {
            if (n >= 2) return out; // more than two targets
            if (n == 0) t0 = cid[k];
            else t1 = cid[k];
            n++;
        }
    if (n == 0) return out;
    // Lower bound of the distance to the chosen target: the choice is one of the survivors, each distance 1-Lipschitz.  A
    // point that may be valid gets 0 (a candidate on a yaw-limit plane then collapses to its offset).  When no survivor is
    // available all over the cell the evaluation may find no target at all and return the raw point (one_leg.cu:141-142).
    {
        double m = 1.0e30;
        bool any_always = false;
        for (int k = 0; k < nc; k++)
            if (!excl[k]) {
                m = lrm_tb_min(m, cd[k]);
                any_always = any_always || calways[k];
            }
        if (!any_always) m = lrm_tb_min(m, lrm_tb_hyp(cx, cz));
        out.lb = vstate == 0 ? lrm_tb_max(0.0, m - rho) : 0.0;
        out.all_invalid = vstate == 0;
    }
    out.n = (uint8_t)n;
    out.t[0] = in.canon_row[t0];
    out.t[1] = in.canon_row[t1]; // (0 = none for a one-target cell)
    out.v = vstate == 0 ? 0 : (vstate == 1 ? 1 : in.canon_vrow[2 + 4 * (int)reg + open_j]);
    out.ok = 1;
    return out;
}

// The bound of a cell without an answer, for the region lists `regs` (bit set): every circle whose clamp point is not
// proven invalid all over the cell, every corner point, the raw point; 0 where a point may be valid.
LRM_HD double lrm_tb_generic_lb(const LrmTbInput& in, unsigned regs, double cx, double cz, double rho, double band, double tau) {
    double lb = lrm_tb_hyp(cx, cz);
    bool maybe_valid = false;
    for (unsigned reg = 0; reg < 4; reg++) {
        if (!(regs & (1u << reg))) continue;
        const LrmTolLeg::Circle* ct = &in.L.circ[reg][0];
        bool one_out = false;
        for (int j = 0; j < LRM_N_CIRCLES; j++) {
            const double vx = cx - (double)ct[j].x, vy = cz - (double)ct[j].y, mag = lrm_tb_hyp(vx, vy);
            bool never = false;
            if (mag > 2.0 * rho) {
                if (fabs((double)ct[j].chw) > 1.0) never = !(ct[j].chw < 0);
                else {
                    const double ux = vx / mag, uy = vy / mag;
                    const double w = vx * (double)ct[j].mx + vy * (double)ct[j].my - (double)ct[j].chw * mag;
                    const double gx = (double)ct[j].mx - (double)ct[j].chw * ux, gy = (double)ct[j].my - (double)ct[j].chw * uy;
                    const double lip = lrm_tb_hyp(gx, gy) + fabs((double)ct[j].chw) * rho / (mag - rho);
                    never = !(w >= 0) && (fabs(w) - (double)ct[j].bw * (mag + rho) > tau + lip * rho);
                }
            }
            if (!never) lb = lrm_tb_min(lb, fabs((double)ct[j].r - mag));
            const double v = (vx * vx + vy * vy) * (double)ct[j].gs + (double)ct[j].c;
            const double lip = 2.0 * fabs((double)ct[j].gs) * (mag + rho) * rho;
            if (v - lip - band > 0) one_out = true;
        }
        if (!one_out) maybe_valid = true;
    }
    for (int i = 0; i < in.L.n_corners; i++) {
        const LrmCircle& f = in.L.feat[4 * LRM_N_CIRCLES + i];
        lb = lrm_tb_min(lb, lrm_tb_hyp(cx - (double)f.x, cz - (double)f.y));
    }
    return maybe_valid ? 0.0 : lrm_tb_max(0.0, lb - rho);
}

// what the cell's points may be evaluated with, or !ok.  Where find_region's rays cross the cell, every region in
// reach must give the same rows.
LRM_HD LrmTbCell lrm_tb_classify_cell(const LrmTbInput& in, double cx, double cz, double rho, double band, double tau) {
    double v[4];
    for (int i = 0; i < 3; i++) v[i] = (double)in.L.dir_cos[i] * cz - (double)in.L.dir_sin[i] * cx;
    v[3] = cz; // the atan2f wrap ray (x < 0, z = +-0) is the sign of z itself
    unsigned open_bits = 0, base = 0;
    for (int i = 0; i < 4; i++) {
        if (!(fabs(v[i]) > band + rho)) open_bits |= 1u << i;
        if (v[i] < 0) base |= 1u << i;
    }
    LrmTbCell code;
    code.lb = 0.0;
    code.ok = 0; code.n = 0; code.all_invalid = 0; code.v = 0; code.t[0] = code.t[1] = 0; code.pad_[0] = code.pad_[1] = 0;
    unsigned seen = 0, reach = 0;
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // the region lists in reach
        reach |= 1u << ((in.L.region_lut >> (((base & ~open_bits) | sub) << 1)) & 3u);
        if (sub == 0) break;
    }
    const double lb_generic = lrm_tb_generic_lb(in, reach, cx, cz, rho, band, tau);
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // every assignment of the open signs
        const unsigned pat = (base & ~open_bits) | sub;
        const unsigned reg = (in.L.region_lut >> (pat << 1)) & 3u;
        if (!(seen & (1u << reg))) {
            LrmTbCell c = lrm_tb_classify_reg(in, reg, cx, cz, rho, band, tau);
            if (!c.ok) {
                c.lb = lb_generic;
                return c;
            }
            if (!seen) code = c;
            else {
                code.lb = lrm_tb_min(code.lb, c.lb);
                code.all_invalid = code.all_invalid && c.all_invalid;
                bool same = c.n == code.n && c.v == code.v;
                if (same && c.n == 1) same = c.t[0] == code.t[0];
                if (same && c.n == 2) same = (c.t[0] == code.t[0] && c.t[1] == code.t[1]) || (c.t[0] == code.t[1] && c.t[1] == code.t[0]);
                if (!same) {
                    code.ok = 0;
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

// The per-point code finds its cell from float arithmetic (one FMA + floor: off by at most 2^-17 of a cell) on a plane
// point that itself carries a few 1e-5 mm of rounding: the classification holds a margin around the cell.
LRM_HD double lrm_tb_slack(double H) { return H * 1.6e-5 + 1.0e-3; }
// coarse cell (ix, iz) of a grid of N x N cells of H mm around the femur joint
LRM_HD LrmTbCell lrm_tb_coarse_cell(const LrmTbInput& in, int g, double H, int ix, int iz) {
    const double half = 0.5 * LRM_TT_N * H;
    const double x0 = -half + ix * H, z0 = -half + iz * H;
    return lrm_tb_classify_cell(in, x0 + 0.5 * H, z0 + 0.5 * H, 0.5 * H * LRM_TB_RHO_FACTOR + lrm_tb_slack(H), in.band[g], in.tau[g]);
}
// sub-cell (sx, sz) of that coarse cell
LRM_HD LrmTbCell lrm_tb_fine_cell(const LrmTbInput& in, int g, double H, int ix, int iz, int sx, int sz) {
    const double half = 0.5 * LRM_TT_N * H, h = H / LRM_TT_SUB;
    const double x0 = -half + ix * H, z0 = -half + iz * H;
    return lrm_tb_classify_cell(in, x0 + (sx + 0.5) * h, z0 + (sz + 0.5) * h, 0.5 * h * LRM_TB_RHO_FACTOR + lrm_tb_slack(H), in.band[g], in.tau[g]);
}

// Lower bound of the distance to the target the evaluation picks at a point within rho of (cx, cz), for a cell with an answer
// whose points are all invalid: the choice is one of the cell's targets (or, when none of them is available all over the cell,
// possibly none: the raw point, one_leg.cu:141-142), each distance 1-Lipschitz.
LRM_HD double lrm_tb_survivor_lb(const LrmTbInput& in, const LrmTbCell& c, double cx, double cz, double rho) {
    double m = 1.0e30;
    bool any_always = false;
    for (int k = 0; k < (int)c.n; k++) {
        const LrmTabRow& t = in.rows[c.t[k]];
        m = lrm_tb_min(m, fabs((double)t.r - lrm_tb_hyp(cx - (double)t.x, cz - (double)t.y)));
        any_always = any_always || t.chw == -2.f;
    }
    if (!any_always) m = lrm_tb_min(m, lrm_tb_hyp(cx, cz));
    return lrm_tb_max(0.0, m - rho);
}

// ---- bounds: LRM_TT_NB^2 bound cells of 2 H, each a plane over its 16 x 16 sub-cells (size hs = H / 8),
//     lb(sx, sz) = d0 + unit (gx sx + gz sz),  unit = hs / 64 mm, gx, gz integers of 8 bits.
// A sub-cell's own bound comes from the coarse cell it lies in: the distance to that cell's targets at the sub-cell's centre
// minus its half-diagonal (cells with an answer and no valid point), the minimum over the fine sub-cells it covers (refined
// cells), else the coarse cell's constant.  The plane: least-squares gradient, quantised, then lowered until no sub-cell's
// bound lies below it -- nothing to prove about curvature.
#define LRM_TB_S 16
struct LrmTbBound {
    double d0;
    int gx, gz;
};
// the own bound of sub-cell (sx, sz) of bound cell (bx, bz); coarse = the grid's N x N cells, fine = the SUB x SUB cells of the
// coarse cell this sub-cell lies in when that one was refined (else null)
LRM_HD double lrm_tb_subcell_lb(const LrmTbInput& in, double H, int bx, int bz, int sx, int sz, const LrmTbCell& c, const LrmTbCell* fine) {
    constexpr int S = LRM_TB_S, kSub = LRM_TT_SUB;
    const double half = 0.5 * LRM_TT_N * H, hs = 2.0 * H / S, rho = 0.5 * hs * LRM_TB_RHO_FACTOR + lrm_tb_slack(H);
    const double cx = -half + bx * 2.0 * H + (sx + 0.5) * hs, cz = -half + bz * 2.0 * H + (sz + 0.5) * hs;
    if (c.ok) return c.all_invalid ? lrm_tb_survivor_lb(in, c, cx, cz, rho) : 0.0;
    if (fine) { // the 2 x 2 fine sub-cells this sub-cell covers
        const int fx = (sx % (S / 2)) * (kSub / (S / 2)), fz = (sz % (S / 2)) * (kSub / (S / 2));
        double v = 1.0e30;
        for (int a = 0; a < kSub / (S / 2); a++)
            for (int b = 0; b < kSub / (S / 2); b++) v = lrm_tb_min(v, fine[(fz + a) * kSub + fx + b].lb);
        return v;
    }
    return c.lb;
}
LRM_HD double lrm_tb_round_half_away(double v) { return v >= 0 ? floor(v + 0.5) : ceil(v - 0.5); } // lround for |v| <= 127
// the plane of one bound cell from its 16 x 16 own bounds lb[sz * 16 + sx]; every sum in this one order
LRM_HD LrmTbBound lrm_tb_fit_bound(const double* lb, double H) {
    constexpr int S = LRM_TB_S;
    const double hs = 2.0 * H / S, unit = hs / 64.0;
    double mean = 0, gx = 0, gz = 0;
    for (int i = 0; i < S * S; i++) mean += lb[i];
    mean /= S * S;
    const double m = 0.5 * (S - 1), var = S * (S * S - 1.0) / 12.0 * S; // sum over the grid of (s - m)^2
    for (int sz = 0; sz < S; sz++)
        for (int sx = 0; sx < S; sx++) {
            gx += (sx - m) * (lb[sz * S + sx] - mean);
            gz += (sz - m) * (lb[sz * S + sx] - mean);
        }
    LrmTbBound e;
    e.gx = (int)lrm_tb_round_half_away(lrm_tb_max(-127.0, lrm_tb_min(127.0, gx / var / unit)));
    e.gz = (int)lrm_tb_round_half_away(lrm_tb_max(-127.0, lrm_tb_min(127.0, gz / var / unit)));
    e.d0 = 1.0e30;
    for (int sz = 0; sz < S; sz++)
        for (int sx = 0; sx < S; sx++) e.d0 = lrm_tb_min(e.d0, lb[sz * S + sx] - unit * (e.gx * sx + e.gz * sz));
    return e;
}
// IEEE half (bits) of a bound, rounded DOWN (towards -inf); normal halves and zero only: a positive value below the smallest
// normal half becomes 0, a negative one above its negative becomes that
LRM_HD uint16_t lrm_tb_half_floor(double v) {
    const bool neg = v < 0;
    double a = fabs(v);
    if (!(a >= 6.103515625e-5)) return neg ? 0x8400 : 0;
    if (a >= 65504.0) a = 65504.0; // (a bound that large does not occur: the grids end at 8192 mm)
    uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = (uint64_t)__double_as_longlong(a);
#else
    memcpy(&u, &a, 8);
#endif
    const int e = (int)((u >> 52) & 0x7ffu) - 1023;                       // a = 1.f * 2^e
    const uint64_t mant = u & 0xfffffffffffffull;                         // 52 fraction bits
    unsigned frac = (unsigned)(mant >> 42);                               // floor of the 10-bit fraction
    const bool inexact = (mant & ((1ull << 42) - 1ull)) != 0ull;
    if (neg && inexact) frac += 1;                                        // towards -inf: the magnitude grows
    unsigned ex = (unsigned)(e + 15);
    if (frac > 1023u) { frac = 0; ex++; }
    if (ex > 30u) { ex = 30u; frac = 1023u; }
    return (uint16_t)((neg ? 0x8000u : 0u) | (ex << 10) | frac);
}
LRM_HD uint32_t lrm_tb_bound_word(const LrmTbBound& e) { // little endian: d0 (half), then the gradient bytes
    return (uint32_t)lrm_tb_half_floor(e.d0) | (((uint32_t)e.gx & 0xffu) << 16) | (((uint32_t)e.gz & 0xffu) << 24);
}

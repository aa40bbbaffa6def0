// lrm_tolgrid.cpp -- builds the plane table of the tolerance mode (LrmTolGridHeader, lrm_types.h) on the host.
//
// For a cell (centre c, half-diagonal rho) the code of lrm_tol_plane at c is the code of every point q of the cell
// when each decision value f keeps |f(c)| > band + L rho, L a Lipschitz constant of f over the cell:
//   region rays (cross products with unit directions)                  L = 1
//   point validity v_j = |q - c_j|^2 gs_j + c_j                        L = 2 |gs_j| (|c - c_j| + rho)
//   clamp validity  w_i = (q - c_i) . m_i - chw_i |q - c_i|            L = 1 + |chw_i|   (|chw_i| > 1: the sign is fixed)
//   ranking: the distances to the clamp targets are 1-Lipschitz, so the winner at c stays the winner over the cell
//            when it leads the runner-up by more than 2 rho (+ the tie band)
//   conditioning of the winner's clamp (LRM_TOL_AMP2) with the smallest |q - c_i| of the cell.
// Everything in double on the host, from the same float tables the device reads.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lrm_compile.h"
#include "lrm_point_tol.h"

namespace {
int g_reason[8];
#define AMB(k) do { g_reason[k]++; return LRM_TG_AMBIG8; } while (0)

// code of the cell if its points are evaluated with circle list `reg`, or 0xff
unsigned classify_cell_reg(const LrmTolLeg& L, unsigned reg, double cx, double cz, double rho, double band, double tau) {
    const LrmTolLeg::Circle* ct = &L.circ[reg][0];
    // validity of the point
    bool all_in = true, one_out = false;
    double mag[LRM_N_CIRCLES];
    for (int j = 0; j < LRM_N_CIRCLES; j++) {
        const double vx = cx - ct[j].x, vy = cz - ct[j].y;
        mag[j] = std::hypot(vx, vy);
        const double v = (vx * vx + vy * vy) * (double)ct[j].gs + (double)ct[j].c;
        const double lip = 2.0 * std::fabs((double)ct[j].gs) * (mag[j] + rho) * rho;
        all_in = all_in && (v + lip + band < 0);
        one_out = one_out || (v - lip - band > 0);
    }
    if (!all_in && !one_out) AMB(1);
    const bool valid = all_in;
    // clamp targets: distance, its gradient at the centre (a unit vector) and the bound rho / |q - c| on how far the
    // gradient turns over the cell.  Distances are 1-Lipschitz, but their DIFFERENCES vary much more slowly where
    // the two targets are seen under a small angle (the far field): d_k - d_w changes by at most
    // rho (|g_k - g_w| + turn_k + turn_w) over the cell.  A circle whose clamp validity is open over the cell
    // ("maybe") only has to lose against the winner.
    struct Cand { double d, gx, gy, turn; int idx; bool flips, maybe; } cand[LRM_N_CIRCLES + LRM_N_CORNERS];
    int nc = 0;
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        if (!(mag[i] > rho)) AMB(2); // the centre of the circle is in (or at) the cell
        const double vx = cx - ct[i].x, vy = cz - ct[i].y;
        const double ux = vx / mag[i], uy = vy / mag[i];
        const double w = vx * (double)ct[i].mx + vy * (double)ct[i].my - (double)ct[i].chw * mag[i];
        bool ok, maybe = false;
        if (std::fabs((double)ct[i].chw) > 1.0) ok = ct[i].chw < 0; // always / never
        else {
            // gradient of w: m - chw u; u turns by at most rho / |q - c| over the cell
            const double gx = (double)ct[i].mx - (double)ct[i].chw * ux, gy = (double)ct[i].my - (double)ct[i].chw * uy;
            const double lip = std::hypot(gx, gy) + std::fabs((double)ct[i].chw) * rho / (mag[i] - rho);
            maybe = !(std::fabs(w) - (double)ct[i].bw * (mag[i] + rho) > tau + lip * rho);
            ok = w >= 0;
        }
        if (!ok && !maybe) continue;
        const double s = (mag[i] >= ct[i].r) ? 1.0 : -1.0;
        cand[nc++] = Cand{std::fabs((double)ct[i].r - mag[i]), s * ux, s * uy, rho / (mag[i] - rho),
                          (int)reg * LRM_N_CIRCLES + i, std::fabs((double)ct[i].r - mag[i]) <= rho, maybe};
    }
    if (!valid)
        for (int i = 0; i < L.n_corners; i++) {
            const LrmCircle& f = L.feat[4 * LRM_N_CIRCLES + i];
            const double vx = cx - f.x, vy = cz - f.y, d = std::hypot(vx, vy);
            if (!(d > rho)) AMB(2);
            cand[nc++] = Cand{d, vx / d, vy / d, rho / (d - rho), 4 * LRM_N_CIRCLES + i, false, false};
        }
    if (nc == 0) AMB(4);
    int wi = 0;
    for (int k = 1; k < nc; k++)
        if (cand[k].d < cand[wi].d) wi = k;
    if (cand[wi].maybe) AMB(3);
    for (int k = 0; k < nc; k++) {
        if (k == wi) continue;
        double lip = std::hypot(cand[k].gx - cand[wi].gx, cand[k].gy - cand[wi].gy) + cand[k].turn + cand[wi].turn;
        if (cand[k].flips || cand[wi].flips || lip > 2.0) lip = 2.0; // a circle crossing the cell: the sign of its gradient is open
        if (!(cand[k].d - cand[wi].d > tau + lip * rho)) AMB(5);
    }
    const int win = cand[wi].idx;
    // conditioning of the winner's clamp
    const LrmCircle& f = L.feat[win];
    const double mw = std::hypot(cx - f.x, cz - f.y) - rho;
    if (!(mw > 0) || !(mw * mw * (double)LRM_TOL_AMP2 > (double)f.r * (double)f.r)) AMB(6);
    return (unsigned)win | (valid ? 32u : 0u);
}

// code of the cell, or 0xff.  Where find_region's rays cross the cell, every region a point of the cell can fall
// in must give the same validity and the same clamp target (the same circle appears in several lists).
unsigned classify_cell(const LrmTolLeg& L, double cx, double cz, double rho, double band, double tau) {
    double v[4];
    for (int i = 0; i < 3; i++) v[i] = (double)L.dir_cos[i] * cz - (double)L.dir_sin[i] * cx;
    v[3] = cz;
    // lrm_region_from_signs reads four sign bits; a value within band + rho of 0 can have either sign in the cell.
    // (The atan2f wrap ray -- x < 0, z = +-0 -- is the sign of z itself.)
    unsigned open_bits = 0, base = 0;
    for (int i = 0; i < 4; i++) {
        if (!(std::fabs(v[i]) > band + rho)) open_bits |= 1u << i;
        if (v[i] < 0) base |= 1u << i;
    }
    unsigned code = LRM_TG_AMBIG8;
    bool first = true;
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) { // every assignment of the open signs
        const unsigned pat = (base & ~open_bits) | sub;
        const unsigned reg = (L.region_lut >> (pat << 1)) & 3u;
        const unsigned c = classify_cell_reg(L, reg, cx, cz, rho, band, tau);
        if (c == LRM_TG_AMBIG8) return c;
        if (first) { code = c; first = false; }
        else {
            const LrmCircle &a = L.feat[code & 31u], &b = L.feat[c & 31u];
            if ((code & 32u) != (c & 32u) || a.x != b.x || a.y != b.y || a.r != b.r) AMB(0);
        }
        if (sub == 0) break;
    }
    return code;
}

} // namespace

size_t lrm_build_tol_grid(const LrmTolLeg& L, size_t max_fine, std::vector<uint8_t>* out) {
    // the largest decision band the table is built for: points up to |p|_1 = 4096 mm
    const double band = (double)L.band_base + (double)L.band_slope * 4096.0;
    const double tau = band * (double)LRM_TOL_TIE;
    const double H = LRM_TG_H, h = H / 4.0;
    std::vector<uint16_t> coarse((size_t)LRM_TG_N * LRM_TG_N);
    std::vector<uint8_t> fine;
    fine.reserve(max_fine * 16);
    size_t n_fine = 0;
    for (int iz = 0; iz < LRM_TG_N; iz++)
        for (int ix = 0; ix < LRM_TG_N; ix++) {
            const double x0 = -LRM_TG_HALF + ix * H, z0 = -LRM_TG_HALF + iz * H;
            unsigned code = classify_cell(L, x0 + 0.5 * H, z0 + 0.5 * H, 0.5 * H * 1.41421357, band, tau);
            if (code == LRM_TG_AMBIG8) {
                if (n_fine < max_fine) {
                    uint8_t sub[16];
                    bool any = false;
                    for (int sz = 0; sz < 4; sz++)
                        for (int sx = 0; sx < 4; sx++) {
                            const unsigned c = classify_cell(L, x0 + (sx + 0.5) * h, z0 + (sz + 0.5) * h, 0.5 * h * 1.41421357, band, tau);
                            sub[sz * 4 + sx] = (uint8_t)c;
                            any = any || c != LRM_TG_AMBIG8;
                        }
                    if (any) {
                        fine.insert(fine.end(), sub, sub + 16);
                        coarse[(size_t)iz * LRM_TG_N + ix] = (uint16_t)(0x8000u | (unsigned)n_fine);
                        n_fine++;
                        continue;
                    }
                }
                coarse[(size_t)iz * LRM_TG_N + ix] = (uint16_t)LRM_TG_AMBIG16;
            } else {
                coarse[(size_t)iz * LRM_TG_N + ix] = (uint16_t)code;
            }
        }
    if (std::getenv("LRM_TOL_DEBUG")) {
        size_t amb_fine = 0, amb_l3 = 0, l3_cells = 0;
        for (int iz = 0; iz < LRM_TG_N; iz++)
            for (int ix = 0; ix < LRM_TG_N; ix++) {
                const uint16_t c = coarse[(size_t)iz * LRM_TG_N + ix];
                if (!(c & 0x8000u) || c == LRM_TG_AMBIG16) continue;
                const double x0 = -LRM_TG_HALF + ix * H, z0 = -LRM_TG_HALF + iz * H;
                for (int k = 0; k < 16; k++)
                    if (fine[(size_t)(c & 0x7fffu) * 16 + k] == LRM_TG_AMBIG8) {
                        amb_fine++;
                        const double fx0 = x0 + (k & 3) * h, fz0 = z0 + (k >> 2) * h;
                        for (int q = 0; q < 16; q++) {
                            l3_cells++;
                            if (classify_cell(L, fx0 + ((q & 3) + 0.5) * h / 4, fz0 + ((q >> 2) + 0.5) * h / 4, 0.5 * h / 4 * 1.41421357, band, tau) == LRM_TG_AMBIG8) amb_l3++;
                        }
                    }
            }
        std::fprintf(stderr, "tol grid: %zu ambiguous fine cells (4 mm); at 1 mm %zu of their %zu sub-cells stay ambiguous\n", amb_fine, amb_l3, l3_cells);
    }
    if (std::getenv("LRM_TOL_DEBUG"))
        std::fprintf(stderr, "tol grid: %zu refined; ambiguous cell tests by reason: rays %d validity %d centre %d clamp %d none %d tie %d cond %d\n",
                     n_fine, g_reason[0], g_reason[1], g_reason[2], g_reason[3], g_reason[4], g_reason[5], g_reason[6]);
    LrmTolGridHeader hd;
    std::memset(&hd, 0, sizeof hd);
    hd.n_fine = (uint32_t)n_fine;
    hd.band_max = (float)band;
    fine.insert(fine.end(), 16, (uint8_t)LRM_TG_AMBIG8); // one spare block: the lookup reads fine[0] for unrefined cells
    out->resize(sizeof hd + coarse.size() * 2 + fine.size());
    std::memcpy(out->data(), &hd, sizeof hd);
    std::memcpy(out->data() + sizeof hd, coarse.data(), coarse.size() * 2);
    if (!fine.empty()) std::memcpy(out->data() + sizeof hd + coarse.size() * 2, fine.data(), fine.size());
    return n_fine;
}

// cand_experiment.cpp -- host experiment (not part of the product): how often the second yaw candidate of the config-2
// cloud is needed, how often it wins, and what a per-cell lower bound of the in-plane distance would prune.
//   g++ -O2 -std=c++17 -I../csrc cand_experiment.cpp ../csrc/build/lrm_compile.o -o build/cand_experiment
#include <random>
#include <algorithm>
#include <unordered_map>
#include "../csrc/lrm_toltab.cpp"

// lower bound of |plane point - clamp target| over a cell, whatever region list and target the evaluation uses
static int g_mode = 0;
static double cell_lb(const LrmTolLeg& L, double cx, double cz, double rho, double band) {
    double lb = std::hypot(cx, cz); // no target at all: the raw point
    bool maybe_valid = false;
    double v[4];
    for (int i = 0; i < 3; i++) v[i] = (double)L.dir_cos[i] * cz - (double)L.dir_sin[i] * cx;
    v[3] = cz;
    unsigned open_bits = 0, base = 0, regs = 0;
    for (int i = 0; i < 4; i++) {
        if (!(std::fabs(v[i]) > band + rho)) open_bits |= 1u << i;
        if (v[i] < 0) base |= 1u << i;
    }
    for (unsigned sub = open_bits;; sub = (sub - 1) & open_bits) {
        regs |= 1u << ((L.region_lut >> ((((base & ~open_bits) | sub)) << 1)) & 3u);
        if (sub == 0) break;
    }
    if (g_mode == 0) regs = 15u;
    for (unsigned reg = 0; reg < 4; reg++) {
        if (!(regs & (1u << reg))) continue;
        const LrmTolLeg::Circle* ct = &L.circ[reg][0];
        bool one_out = false;
        for (int j = 0; j < LRM_N_CIRCLES; j++) {
            const double vx = cx - ct[j].x, vy = cz - ct[j].y, mag = std::hypot(vx, vy);
            bool never = false;
            if (g_mode >= 2 && mag > 2.0 * rho) {
                const double ux = vx / mag, uy = vy / mag;
                const double w = vx * (double)ct[j].mx + vy * (double)ct[j].my - (double)ct[j].chw * mag;
                if (std::fabs((double)ct[j].chw) > 1.0) never = !(ct[j].chw < 0);
                else {
                    const double gx = (double)ct[j].mx - (double)ct[j].chw * ux, gy = (double)ct[j].my - (double)ct[j].chw * uy;
                    const double lip = std::hypot(gx, gy) + std::fabs((double)ct[j].chw) * rho / (mag - rho);
                    const bool maybe = !(std::fabs(w) - (double)ct[j].bw * (mag + rho) > band + lip * rho);
                    never = !(w >= 0) && !maybe;
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
    if (g_mode >= 4 && !maybe_valid) {
        g_reason = 0;
        const double tau = band * LRM_TOL_TIE;
        const CellCode c = classify_cell(L, cx, cz, rho, band, tau);
        if (c.ok) {
            double m = 1e30;
            for (int k = 0; k < c.n; k++) m = std::min(m, std::fabs((double)c.t[k].r - std::hypot(cx - c.t[k].x, cz - c.t[k].y)));
            if (m - rho < lb - rho - 1e-9) printf("survivor bound below the generic one?\n");
            return std::max(0.0, m - rho);
        }
    }
    return maybe_valid ? 0.0 : std::max(0.0, lb - rho);
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1000000;
    const double H = argc > 2 ? atof(argv[2]) : 16.0;
    g_mode = argc > 3 ? atoi(argv[3]) : 0;
    LrmLegDimensions leg;
    lrm_host_leg_factory(0.f, 181.f, -45.f, 65.5f, 129.f, 135.f, 60.f, 90.f, 120.f, -5.f, -5.f, &leg); // get_M2_leg
    const float quat[4] = {1, 0, 0, 0};
    LrmCompiledLeg CL;
    lrm_compile_leg(leg, quat, 1, &CL);
    LrmTolLeg L;
    lrm_compile_tol(CL, &L);
    const LrmTolTables T{&L.circ[0][0], &L.feat[0]};
    const double band = (double)L.band_base + (double)L.band_slope * 4096.0;
    struct Lin { float d0, gx, gz; };
    std::unordered_map<uint64_t, Lin> memo_lin;
    auto lin_at = [&](float u, float z) {
        const double x = (double)u - (double)L.coxa_length;
        const long ix = (long)std::floor(x / H), iz = (long)std::floor(z / H);
        const uint64_t key = ((uint64_t)(uint32_t)ix << 32) | (uint32_t)iz;
        auto it = memo_lin.find(key);
        if (it == memo_lin.end()) {
            const int S = 16; const double h = H / S;
            double sub[16][16];
            const int keep = g_mode; g_mode = 4;
            for (int a = 0; a < S; a++) for (int b = 0; b < S; b++)
                sub[a][b] = cell_lb(L, ix * H + (b + 0.5) * h, iz * H + (a + 0.5) * h, 0.5 * h * 1.41421357 + 1e-3, band);
            g_mode = keep;
            // gradient: least squares plane through the sub-cell bounds, quantised to 1/64 per sub-cell step
            double sx = 0, sz = 0, n = 0, mx = 7.5, mz = 7.5, cxx = 0, czz = 0, m = 0;
            for (int a = 0; a < S; a++) for (int b = 0; b < S; b++) m += sub[a][b];
            m /= 256.0;
            for (int a = 0; a < S; a++) for (int b = 0; b < S; b++) { sx += (b - mx) * (sub[a][b] - m); sz += (a - mz) * (sub[a][b] - m); cxx += (b - mx) * (b - mx); czz += (a - mz) * (a - mz); }
            (void)n;
            const double gx = std::round(sx / cxx * 64.0) / 64.0, gz = std::round(sz / czz * 64.0) / 64.0;
            double d0 = 1e30;
            for (int a = 0; a < S; a++) for (int b = 0; b < S; b++) d0 = std::min(d0, sub[a][b] - gx * b - gz * a);
            it = memo_lin.emplace(key, Lin{(float)d0, (float)gx, (float)gz}).first;
        }
        const double fx = std::floor((x - ix * H) / H * 16.0), fz = std::floor(((double)z - iz * H) / H * 16.0);
        return std::max(0.f, (float)(it->second.d0 + it->second.gx * fx + it->second.gz * fz));
    };
    std::unordered_map<uint64_t, float> memo;
    auto lb_at = [&](float u, float z) {
        if (g_mode == 5) return lin_at(u, z);
        const double x = (double)u - (double)L.coxa_length;
        const long ix = (long)std::floor(x / H), iz = (long)std::floor(z / H);
        const uint64_t key = ((uint64_t)(uint32_t)ix << 32) | (uint32_t)iz;
        auto it = memo.find(key);
        if (it == memo.end()) it = memo.emplace(key, (float)cell_lb(L, (ix + 0.5) * H, (iz + 0.5) * H, 0.5 * H * 1.41421357 + 1e-3, band)).first;
        return it->second;
    };
    std::mt19937_64 rng(42);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    size_t two = 0, need = 0, flag = 0, cls[2][2] = {{0}}, wins1[2][2] = {{0}}, need_new[2][2] = {{0}}, wrong = 0;
    size_t close_gap[4] = {0, 0, 0, 0}; // | |d_1| - |d_0| | below 4 / 16 / 32 / 64 mm
    for (size_t i = 0; i < n; i++) {
        LrmVec3 p{U(rng) * 900.f - 200.f, U(rng) * 1000.f - 500.f, U(rng) * 800.f - 500.f};
        const LrmTolPoint S = lrm_tol_prologue(L, p);
        uint32_t lu = 0;
        float du, dz; bool valid;
        lrm_tol_plane(L, T, S.u0, S.z, S.band, S.tau, du, dz, valid, lu);
        const LrmTolCand A = lrm_tol_candidate(S, false, du, dz, valid, lu);
        const bool nd = lrm_tol_need_second(L, S, A);
        if (A.flag) flag++;
        if (!S.two || A.flag) continue;
        two++;
        need += nd;
        lrm_tol_plane(L, T, S.u1, S.z, S.band, S.tau, du, dz, valid, lu);
        const LrmTolCand B = lrm_tol_candidate(S, true, du, dz, valid, lu);
        cls[S.lim0][S.lim1]++;
        if (B.n < A.n) wins1[S.lim0][S.lim1]++;
        {
            const double gap = std::fabs(std::sqrt((double)B.n) - std::sqrt((double)A.n));
            close_gap[0] += gap < 4.0; close_gap[1] += gap < 16.0; close_gap[2] += gap < 32.0; close_gap[3] += gap < 64.0;
        }
        // the new scheme: bounds b_i = w_i^2 + lb_i^2; the candidate in range goes first, else the smaller bound; the other one is
        // needed unless n_first < b_other - thr
        float lb0 = lb_at(S.u0, S.z), lb1 = lb_at(S.u1, S.z);
        if (g_mode == 9) { lb0 = std::sqrt(std::max(0.f, A.n - S.w0 * S.w0)) ; lb1 = std::sqrt(std::max(0.f, B.n - S.w1 * S.w1)); }
        const float b0 = S.w0 * S.w0 + lb0 * lb0, b1 = S.w1 * S.w1 + lb1 * lb1;
        if (A.n < b0 * 0.9999f - 1e-3f || B.n < b1 * 0.9999f - 1e-3f) wrong++;
        const bool first0 = g_mode >= 3 ? b0 <= b1 : (S.in0 || b0 <= b1);
        const float nf = first0 ? A.n : B.n, bo = first0 ? b1 : b0;
        const float thr = S.tau * (2.f * std::sqrt(nf) + S.tau);
        if (!(nf < bo - thr)) need_new[S.lim0][S.lim1]++;
    }
    printf("H %.0f mm, points %zu: flag %.4f, two&&!flag %.4f, need (present bound) %.4f; bound violated %zu\n", H, n, (double)flag / n, (double)two / n, (double)need / n, wrong);
    printf("  the two candidates differ by less than 4 / 16 / 32 / 64 mm for %.4f / %.4f / %.4f / %.4f of the points\n", (double)close_gap[0] / n,
           (double)close_gap[1] / n, (double)close_gap[2] / n, (double)close_gap[3] / n);
    size_t tot = 0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (cls[a][b]) {
        printf("  first clamped %d second clamped %d: %.4f of points; second wins %.4f; need with cell bounds %.5f\n", a, b, (double)cls[a][b] / n, (double)wins1[a][b] / n, (double)need_new[a][b] / n);
        tot += need_new[a][b];
    }
    const double pn = (double)tot / n;
    printf("  need with cell bounds %.5f of points -> %.3f of waves\n", pn, 1.0 - std::pow(1.0 - pn, 64));
    return 0;
}

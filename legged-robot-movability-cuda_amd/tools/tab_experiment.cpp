// tab_experiment.cpp -- host experiment (not part of the product): which plane evaluations of the config-2 cloud the
// plane table with deferred decisions (csrc/lrm_toltab.cpp) leaves unanswered, by reason and cell size.
//   g++ -O2 -std=c++17 -I../csrc tab_experiment.cpp ../csrc/build/lrm_compile.o -o build/tab_experiment
#include <random>
#include <unordered_map>
#include "../csrc/lrm_toltab.cpp"

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1000000;
    LrmLegDimensions leg;
    lrm_host_leg_factory(0.f, 181.f, -45.f, 65.5f, 129.f, 135.f, 60.f, 90.f, 120.f, -5.f, -5.f, &leg); // get_M2_leg
    const float quat[4] = {1, 0, 0, 0};
    LrmCompiledLeg CL;
    lrm_compile_leg(leg, quat, 1, &CL);
    LrmTolLeg L;
    lrm_compile_tol(CL, &L);
    const LrmTolTables T{&L.circ[0][0], &L.feat[0]};
    const double band = (double)L.band_base + (double)L.band_slope * 4096.0, tau = band * LRM_TOL_TIE;
    for (double h : {16.0, 4.0, 1.0}) {
        std::unordered_map<uint64_t, int> memo; // cell -> reason (0 = answered)
        size_t cnt[8] = {0}, lookups = 0, defer = 0;
        std::mt19937_64 rng(42);
        std::uniform_real_distribution<float> U(0.f, 1.f);
        for (size_t i = 0; i < n; i++) {
            LrmVec3 p{U(rng) * 900.f - 200.f, U(rng) * 1000.f - 500.f, U(rng) * 800.f - 500.f};
            const LrmTolPoint S = lrm_tol_prologue(L, p);
            uint32_t lu = 0;
            float du, dz; bool valid;
            lrm_tol_plane(L, T, S.u0, S.z, S.band, S.tau, du, dz, valid, lu);
            const LrmTolCand A = lrm_tol_candidate(S, false, du, dz, valid, lu);
            const bool need = lrm_tol_need_second(L, S, A);
            for (int k = 0; k < (need ? 2 : 1); k++) {
                const double x = (double)(k ? S.u1 : S.u0) - (double)L.coxa_length, z = S.z;
                const long ix = (long)std::floor(x / h), iz = (long)std::floor(z / h);
                const uint64_t key = ((uint64_t)(uint32_t)ix << 32) | (uint32_t)iz;
                auto it = memo.find(key);
                if (it == memo.end()) {
                    g_reason = 0;
                    const CellCode c = classify_cell(L, (ix + 0.5) * h, (iz + 0.5) * h, 0.5 * h * 1.41421357, band, tau);
                    int r = c.ok ? 0 : g_reason;
                    const bool open_validity = std::memcmp(&c.v, &kFalseRow, sizeof c.v) != 0 && std::memcmp(&c.v, &kTrueRow, sizeof c.v) != 0;
                    if (c.ok && (c.n == 2 || open_validity)) r = -1; // a cell that defers a decision to the lane
                    it = memo.emplace(key, r).first;
                }
                lookups++;
                if (it->second > 0) cnt[it->second]++;
                if (it->second < 0) defer++;
            }
        }
        printf("h %5.2f: lookups/point %.3f; deferring cells %.4f; unanswered per lookup: regions differ %.4f, two validities %.4f, centre %.4f, >2 targets %.4f, none %.4f\n",
               h, (double)lookups / n, (double)defer / lookups, (double)cnt[1] / lookups, (double)cnt[2] / lookups, (double)cnt[3] / lookups, (double)cnt[4] / lookups, (double)cnt[5] / lookups);
    }
    return 0;
}

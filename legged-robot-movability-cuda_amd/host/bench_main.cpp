// bench_main.cpp -- the reference's benchmark harness (bench.cpp) re-hosted on liblrm.so.
//
// Same sweep as bench.cpp:61-180: for each of {GPU reach, CPU reach, GPU distance, CPU
// distance}, planar grids x in [XMin,XMax], y = 0, z in [XMin (sic, bench.cpp:114), ZMax] at
// pitch MinPix * 2^k <= MaxPix, `subsample` repeats each, one CSV row "N;ns_per_point" per
// repeat (bench.cpp:164-171).  Written against include/lrm_compat.hpp, i.e. against the
// reference's own API names.  Compute index 4 (bench.cpp:87-91, :153-159: apply_RBDL, MinPixRBDL = 0.4,
// 3 repeats, rbdl.csv) runs the RBDL-equivalent Levenberg-Marquardt IK of lrm_rbdl_equiv_cpu: RBDL itself is
// an external, unpinned dependency that is absent here -- a timing baseline, parity unpinned.
//
//   lrm_bench [outdir] [min_pix] [gpu_repeats] [cpu_repeats] [rbdl_repeats] [min_pix_rbdl] [fast|tol|tol_rel|strict]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "../../include/lrm_compat.hpp"

// setting_bench.h:3-18
constexpr float Spacing = 2;
constexpr float MaxPix = 50;
constexpr float XMin = -100, XMax = 601, YMin = 0, YMax = 0, ZMax = 51;

static std::vector<float> arange(float start, float end, float step) {
    std::vector<float> r;
    for (float v = start; v <= end; v += step) r.push_back(v);
    return r;
}

static Array<float3> generate3DGrid(const std::vector<float>& xs, const std::vector<float>& ys,
                                    const std::vector<float>& zs) {
    Array<float3> out;
    out.length = xs.size() * ys.size() * zs.size();
    out.elements = new float3[out.length];
    size_t i = 0;
    for (float x : xs)
        for (float y : ys)
            for (float z : zs) out.elements[i++] = {x, y, z};
    return out;
}

int main(int argc, char** argv) {
    const std::string outdir = argc > 1 ? argv[1] : ".";
    const float min_pix = argc > 2 ? (float)atof(argv[2]) : 0.04f; // setting_bench.h:9
    const int gpu_rep = argc > 3 ? atoi(argv[3]) : 100;            // SubSamples_GPU
    const int cpu_rep = argc > 4 ? atoi(argv[4]) : 10;             // SubSamples_CPU
    const int rbdl_rep = argc > 5 ? atoi(argv[5]) : 3;             // SubSamples_RBDL
    const float min_pix_rbdl = argc > 6 ? (float)atof(argv[6]) : 0.4f; // MinPixRBDL
    if (argc > 7) { // arithmetic mode of the GPU kernels: fast (default, bit-exact) | tol (contract tolerance) | tol_rel (1e-5 relative on every vector) | strict
        const std::string m = argv[7];
        const int mode = m == "tol" ? LRM_MODE_TOL : (m == "tol_rel" ? LRM_MODE_TOL_REL : (m == "strict" ? LRM_MODE_STRICT : LRM_MODE_FAST));
        if (lrm_set_mode(mode) != 0) {
            std::cerr << "lrm_set_mode failed" << std::endl;
            return 1;
        }
    }
    const LegDimensions dim = get_M2_leg(0);                       // RobotNumb == 1, settings.h:58
    const char* files[5] = {"rgpu.csv", "rcpu.csv", "dgpu.csv", "dcpu.csv", "rbdl.csv"};
    for (int ci = 0; ci < 5; ci++) {
        const bool rbdl = ci == 4;
        const bool gpu = !rbdl && (ci % 2) == 0, reach = rbdl || ci < 2;
        const int subsample = rbdl ? rbdl_rep : (gpu ? gpu_rep : cpu_rep);
        if (subsample <= 0) continue;
        std::ofstream csv(outdir + "/" + files[ci]);
        if (!csv.is_open()) {
            std::cerr << "Failed to open file." << std::endl;
            return 1;
        }
        for (float pix = rbdl ? min_pix_rbdl : min_pix; pix <= MaxPix; pix *= Spacing) {
            Array<float3> target_map = generate3DGrid(arange(XMin, XMax, pix), arange(YMin, YMax, pix),
                                                      arange(XMin, ZMax, pix));
            double last = 0;
            for (int sub = 0; sub < subsample; sub++) {
                double duration;
                if (reach) {
                    Array<bool> out{target_map.length, new bool[target_map.length]};
                    duration = rbdl ? apply_RBDL(target_map, dim, out)
                               : gpu ? apply_kernel(target_map, dim, reachability_global_kernel, out)
                                     : apply_reach_cpu(target_map, dim, out);
                    delete[] out.elements;
                } else {
                    Array<float3> out{target_map.length, new float3[target_map.length]};
                    duration = gpu ? apply_kernel(target_map, dim, distance_global_kernel, out)
                                   : apply_dist_cpu(target_map, dim, out);
                    delete[] out.elements;
                }
                last = duration / (double)target_map.length * 1'000'000.0;
                csv << (long)target_map.length << ";" << last << std::endl;
            }
            std::cout << files[ci] << ": " << last << " ns per point (total: " << target_map.length << ")"
                      << std::endl;
            delete[] target_map.elements;
        }
    }
    return 0;
}

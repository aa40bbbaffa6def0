// cuda_main.cpp -- the reference's file-to-file driver (target `cuda`, several_leg.cpp:124-224)
// on liblrm.so: read dist_input_t{x,y,z}.bin (headerless little-endian float32, one file per
// component: readArrayFromFile, math_util.cpp:63-89), compute reachability and the distance field
// for the leg selected by RobotNumb (settings.h:58: 1 = M2, 0 = moonbot) at azimuth 0, write
// out_reachability.bin (one byte per point, several_leg.cpp:159-160) and out_dist_x{x,y,z}.bin
// (:201-219).  The on-disk SoA layout goes straight to the device SoA kernels; the reference's
// threeArrays2float3Arr AoS conversion (math_util.cpp:92) is not needed.
//
//   lrm_cuda [dir] [robot: 1 = M2 (default), 0 = moonbot]
// so that LAUNCH.bash's `before.py -> ./cuda -> after.py` chain runs unchanged around it.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "../../include/lrm.h"

static bool read_f32(const std::string& path, std::vector<float>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) {
        std::cerr << "Error opening file: " << path << std::endl; // math_util.cpp:84
        return false;
    }
    f.seekg(0, std::ios::end);
    const long len = (long)f.tellg() / (long)sizeof(float);
    f.seekg(0, std::ios::beg);
    out.resize((size_t)len);
    f.read(reinterpret_cast<char*>(out.data()), len * (long)sizeof(float));
    return true;
}

template <class T> static void save(const std::string& path, const T* data, size_t n) {
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) {
        std::cout << "error saving file"; // math_util.cpp:52
        return;
    }
    f.write(reinterpret_cast<const char*>(data), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? std::string(argv[1]) + "/" : "";
    const int robot = argc > 2 ? atoi(argv[2]) : 1;
    LrmLegDimensions dim;
    if (robot == 0) lrm_get_moonbot_leg(0.f, &dim);
    else lrm_get_M2_leg(0.f, &dim);
    std::vector<float> x, y, z;
    if (!read_f32(dir + "dist_input_tx.bin", x) || !read_f32(dir + "dist_input_ty.bin", y) ||
        !read_f32(dir + "dist_input_tz.bin", z) || x.size() != y.size() || x.size() != z.size()) {
        std::cerr << "dist_input_t{x,y,z}.bin missing or of different lengths" << std::endl;
        return 1;
    }
    const size_t n = x.size();
    std::vector<unsigned char> reach(n);
    float ms = 0.f;
    if (lrm_reach_soa(x.data(), y.data(), z.data(), n, &dim, nullptr, reach.data(), &ms) != LRM_OK) {
        std::fprintf(stderr, "HIP error in Kernel launch: %s\n", lrm_last_error());
        return EXIT_FAILURE;
    }
    std::cout << "Cuda reachability took " << ms << " milliseconds to finish." << std::endl; // several_leg.cpp:151
    std::cout << "That's " << (double)ms / (double)n * 1'000'000.0 << " ns per point (total: " << n << ")" << std::endl;
    save(dir + "out_reachability.bin", reach.data(), n);
    std::vector<float> dx(n), dy(n), dz(n);
    if (lrm_dist_soa(x.data(), y.data(), z.data(), n, &dim, nullptr, dx.data(), dy.data(), dz.data(), nullptr, &ms) != LRM_OK) {
        std::fprintf(stderr, "HIP error in Kernel launch: %s\n", lrm_last_error());
        return EXIT_FAILURE;
    }
    std::cout << "Cuda distance took " << ms << " milliseconds to finish." << std::endl; // several_leg.cpp:188
    std::cout << "That's " << (double)ms / (double)n * 1'000'000.0 << " ns per point (total: " << n << ")" << std::endl;
    save(dir + "out_dist_xx.bin", dx.data(), n);
    save(dir + "out_dist_xy.bin", dy.data(), n);
    save(dir + "out_dist_xz.bin", dz.data(), n);
    return 0;
}

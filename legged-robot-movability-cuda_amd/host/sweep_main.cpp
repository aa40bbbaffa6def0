// sweep_main.cpp -- the reference-shaped C++ entry points of the positionability paths, end to end:
//   robot_full_struct(body_map, target_map, legs)   several_leg.cu.h:12-14, several_leg.cu:796-877
//   apply_oct(input, dim, output)                   several_leg_octree.cu.h:4, several_leg_octree.cu:391-488
// called exactly as the reference's (commented-out / dead) call sites do (several_leg.cpp:25-123, :245), on
// headerless float32 files in the reference's on-disk layout (one file per component, math_util.cpp:46-89).
// Written against include/lrm_compat.hpp only: plain g++, no HIP headers.
//
//   lrm_sweep <dir> <robot: 0 moonbot | 1 M2> <nlegs>
// reads  <dir>/body_{x,y,z}.bin, <dir>/target_{x,y,z}.bin
// writes <dir>/accepted_{x,y,z}.bin (robot_full_struct: accepted bodies), <dir>/accepted_count.bin (int32),
//        <dir>/oct_{x,y,z}.bin (apply_oct on the targets as footholds, leg 0: centres of the valid leaves)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/lrm_compat.hpp"

static std::vector<float> read_f32(const std::string& path) {
    std::vector<float> v;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "Failed to open file %s\n", path.c_str());
        std::exit(1);
    }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)bytes / sizeof(float));
    if (!v.empty() && std::fread(v.data(), sizeof(float), v.size(), f) != v.size()) std::exit(1);
    std::fclose(f);
    return v;
}
template <class T> static void write_raw(const std::string& path, const std::vector<T>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) {
        std::fprintf(stderr, "Failed to open file %s\n", path.c_str());
        std::exit(1);
    }
    if (!v.empty()) std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}
static Array<float3> read_cloud(const std::string& dir, const char* stem) { // threeArrays2float3Arr, math_util.cpp:92
    const std::vector<float> x = read_f32(dir + "/" + stem + "_x.bin"), y = read_f32(dir + "/" + stem + "_y.bin"),
                             z = read_f32(dir + "/" + stem + "_z.bin");
    Array<float3> a{x.size(), new float3[x.size() ? x.size() : 1]};
    for (size_t i = 0; i < x.size(); i++) a.elements[i] = {x[i], y[i], z[i]};
    return a;
}
static void write_cloud(const std::string& dir, const char* stem, const Array<float3>& a) {
    std::vector<float> x(a.length), y(a.length), z(a.length);
    for (size_t i = 0; i < a.length; i++) {
        x[i] = a.elements[i].x;
        y[i] = a.elements[i].y;
        z[i] = a.elements[i].z;
    }
    write_raw(dir + "/" + stem + "_x.bin", x);
    write_raw(dir + "/" + stem + "_y.bin", y);
    write_raw(dir + "/" + stem + "_z.bin", z);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: lrm_sweep <dir> <robot 0|1> <nlegs>\n");
        return 2;
    }
    const std::string dir = argv[1];
    const int robot = std::atoi(argv[2]), nlegs = std::atoi(argv[3]);
    const float pI = 3.14159265358979323846264338327950288419716939937510582097f;
    Array<float3> body_map = read_cloud(dir, "body"), target_map = read_cloud(dir, "target");
    // legs mounted every 2 pi / nlegs, as the commented-out call site builds them (several_leg.cpp:60-70)
    Array<LegDimensions> legs{(size_t)nlegs, new LegDimensions[nlegs]};
    for (int l = 0; l < nlegs; l++) {
        const float az = 2 * pI * (float)l / (float)nlegs;
        legs.elements[l] = robot ? get_M2_leg(az) : get_moonbot_leg(az);
    }
    Array<float3> accepted;
    Array<int> counts;
    std::tie(accepted, counts) = robot_full_struct(body_map, target_map, legs);
    write_cloud(dir, "accepted", accepted);
    write_raw(dir + "/accepted_count.bin", std::vector<int>(counts.elements, counts.elements + counts.length));
    std::printf("robot_full_struct: %zu of %zu bodies accepted\n", accepted.length, body_map.length);

    Array<float3> oct{0, new float3[1]};
    const float ms = apply_oct(target_map, legs.elements[0], oct);
    write_cloud(dir, "oct", oct);
    std::printf("apply_oct: %zu valid leaves, %.3f ms of kernels\n", oct.length, ms);
    delete[] accepted.elements;
    delete[] counts.elements;
    delete[] oct.elements;
    delete[] body_map.elements;
    delete[] target_map.elements;
    delete[] legs.elements;
    return 0;
}

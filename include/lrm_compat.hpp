// lrm_compat.hpp -- header-only C++ mirror of the reference's host interface for the hot
// path, on top of the C ABI (lrm.h).  A harness written against the reference's
// cross_compiled.cuh / HeaderCUDA.h / static_variables.h keeps its logic unchanged:
//
//   Array<float3> in; Array<bool> out; LegDimensions dim = get_M2_leg(0);
//   float ms = apply_kernel(in, dim, reachability_global_kernel, out);     // GPU (HIP)
//   double ms2 = apply_reach_cpu(in, dim, out);                            // CPU
//
// Reference declarations mirrored here:
//   Array<T>                     HeaderCUDA.h:38-66     {size_t length; T* elements;}
//   LegDimensions                HeaderCPP.h:19-52      14 floats, 56 bytes
//   apply_kernel<Tin,param,Tout> cross_compiled.cuh:4-7 (kernel chosen by function-pointer identity)
//   apply_reach_cpu/apply_dist_cpu cross_compiled.cuh:12-15
//   reachability_global_kernel / distance_global_kernel  one_leg.cu.h:34-37 (tags here)
//   get_M2_leg / get_moonbot_leg static_variables.h
//   robot_full_struct            several_leg.cu.h:12-14
//   apply_oct                    several_leg_octree.cu.h:4
//   apply_RBDL                   RBDL_benchmark.h:5 (RBDL-equivalent solver, see below)
// Error behaviour is the reference's: print to stderr and exit(EXIT_FAILURE)
// (CUDA_CHECK_ERROR, cross_compiled.cu:12-20).
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>
#include "lrm.h"

#ifndef LRM_COMPAT_NO_FLOAT3
// 12-byte {x,y,z}; if the HIP headers are included first, use theirs by defining LRM_COMPAT_NO_FLOAT3
struct float3 {
    float x, y, z;
};
struct float4 {
    float x, y, z, w;
};
#endif
static_assert(sizeof(float3) == 12, "float3 must be 12 bytes (AoS stride of the reference)");

typedef LrmLegDimensions LegDimensions;
static_assert(sizeof(LegDimensions) == 56, "LegDimensions must be 14 floats");
typedef float4 Quaternion;
constexpr Quaternion quatTest = {1, 0, 0, 0}; // settings.h:51

template <typename T> struct Array {
    size_t length;
    T* elements;
};

// kernel tags: the reference selects the computation by the address of a __global__ function
inline void reachability_global_kernel(const Array<float3>, const LegDimensions, Array<bool>) {}
inline void distance_global_kernel(const Array<float3>, const LegDimensions, Array<float3>) {}

namespace lrm_compat_detail {
[[noreturn]] inline void die(const char* where) {
    std::fprintf(stderr, "HIP error in %s: %s\n", where, lrm_last_error());
    std::exit(EXIT_FAILURE);
}
inline void check(int rc, const char* where) {
    if (rc != LRM_OK) die(where);
}
} // namespace lrm_compat_detail

// apply_kernel<float3, LegDimensions, bool>, cross_compiled.cu:33-79 (returns kernel ms)
inline float apply_kernel(const Array<float3> input, const LegDimensions dim,
                          void (*kernel)(const Array<float3>, const LegDimensions, Array<bool>),
                          Array<bool> const output) {
    (void)kernel; // only reachability_global_kernel has this signature
    static_assert(sizeof(bool) == 1, "bool outputs are one byte");
    float ms = 0.f;
    lrm_compat_detail::check(lrm_reach(&input.elements->x, input.length, &dim, &quatTest.x,
                                       reinterpret_cast<uint8_t*>(output.elements), &ms),
                             "Kernel launch");
    return ms;
}

// apply_kernel<float3, LegDimensions, float3>
inline float apply_kernel(const Array<float3> input, const LegDimensions dim,
                          void (*kernel)(const Array<float3>, const LegDimensions, Array<float3>),
                          Array<float3> const output) {
    (void)kernel; // only distance_global_kernel has this signature
    float ms = 0.f;
    lrm_compat_detail::check(lrm_dist(&input.elements->x, input.length, &dim, &quatTest.x,
                                      &output.elements->x, nullptr, &ms),
                             "Kernel launch");
    return ms;
}

// apply_kernel_multi: NEW (the reference is single-device, several_leg.cu:800): the fused reach + distance of one cloud
// over `ndev` GPUs of this process -- contiguous 64-point-aligned shards, one stream per device, RCCL all-gather of the
// bit-packed reach mask (lrm_reach_dist_multi).  Same ownership rules as apply_kernel (cross_compiled.cu:33-79): the
// caller owns all four arrays; gathered_bits (may be {0, nullptr}) receives ceil(n / 64) words.  Returns the largest
// per-device kernel time in ms.
inline float apply_kernel_multi(const Array<float3> input, const LegDimensions dim, int ndev, Array<bool> const reach_out,
                                Array<float3> const dist_out, Array<unsigned long long> const gathered_bits = {0, nullptr}) {
    static_assert(sizeof(bool) == 1 && sizeof(unsigned long long) == 8, "byte masks, 64-bit ballot words");
    float ms[64] = {0.f};
    if (ndev < 1 || ndev > 64) lrm_compat_detail::die("apply_kernel_multi");
    lrm_compat_detail::check(lrm_reach_dist_multi(&input.elements->x, input.length, &dim, &quatTest.x, ndev, nullptr,
                                                  reinterpret_cast<uint8_t*>(reach_out.elements), &dist_out.elements->x,
                                                  reinterpret_cast<uint64_t*>(gathered_bits.elements), ms),
                             "Kernel launch");
    float worst = 0.f;
    for (int d = 0; d < ndev; d++) worst = ms[d] > worst ? ms[d] : worst;
    return worst;
}

// apply_reach_cpu / apply_dist_cpu, cross_compiled.cu:163-181 (return ms as double)
inline double apply_reach_cpu(const Array<float3> input, const LegDimensions dim, Array<bool> const output) {
    double ms = 0;
    lrm_compat_detail::check(lrm_reach_cpu(&input.elements->x, input.length, &dim, &quatTest.x,
                                           reinterpret_cast<uint8_t*>(output.elements), &ms),
                             "apply_reach_cpu");
    return ms;
}
inline double apply_dist_cpu(const Array<float3> input, const LegDimensions dim, Array<float3> const output) {
    double ms = 0;
    lrm_compat_detail::check(lrm_dist_cpu(&input.elements->x, input.length, &dim, &quatTest.x,
                                          &output.elements->x, nullptr, &ms),
                             "apply_dist_cpu");
    return ms;
}

// apply_RBDL, RBDL_benchmark.h:5 / rbdl_benchmark.cpp:18-111 (returns the loop's ms).  RBDL itself is an external,
// unpinned dependency that is absent here: this runs the same Levenberg-Marquardt position-IK iteration on the
// same chain ("RBDL-equivalent", lrm_rbdl_equiv_cpu): a timing baseline, its mask's parity is unpinned.
inline float apply_RBDL(Array<float3> input, LegDimensions leg, Array<bool> output) {
    double ms = 0;
    lrm_compat_detail::check(lrm_rbdl_equiv_cpu(&input.elements->x, input.length, &leg,
                                                reinterpret_cast<uint8_t*>(output.elements), &ms),
                             "apply_RBDL");
    return (float)ms;
}

// apply_oct, several_leg_octree.cu.h:4 / several_leg_octree.cu:391-488: replaces output.elements by a
// new[]-ed array of the valid leaves' centres (the reference delete[]s the old one, :469-472).
inline float apply_oct(Array<float3> input, LegDimensions dim, Array<float3>& output) {
    size_t n = 0, cap = 4096;
    float ms = 0.f;
    float3* buf = new float3[cap];
    int rc = lrm_apply_oct(&input.elements->x, input.length, &dim, nullptr, &buf->x, cap, &n, &ms);
    if (rc == LRM_EINVAL && n > cap) {
        delete[] buf;
        cap = n;
        buf = new float3[cap];
        rc = lrm_apply_oct(&input.elements->x, input.length, &dim, nullptr, &buf->x, cap, &n, &ms);
    }
    if (rc != LRM_OK) {
        std::fprintf(stderr, "HIP error in Kernel launch: %s\n", lrm_octree_last_error());
        std::exit(EXIT_FAILURE);
    }
    delete[] output.elements;
    output.elements = buf;
    output.length = n;
    return ms;
}

inline LegDimensions get_M2_leg(float body_angle) {
    LegDimensions l;
    lrm_get_M2_leg(body_angle, &l);
    return l;
}
inline LegDimensions get_moonbot_leg(float body_angle) {
    LegDimensions l;
    lrm_get_moonbot_leg(body_angle, &l);
    return l;
}

// robot_full_struct, several_leg.cu:796-877: orientation sweep roll(3) x pitch(3) x yaw(5)
// (quat = yaw * pitch * roll, several_leg.cu:831-857); returns the accepted bodies and the
// reference's dummy count array (filled with 3, several_leg.cu:868); both new[]-ed, owned by
// the caller.  The estimator's one-time sphere culls (several_leg.cu:413-502) and per-orientation
// cylinder culls (:504-559) are applied; accepted bodies come back in input order (the reference's
// thrust::partition leaves them in an unspecified order).
inline std::tuple<Array<float3>, Array<int>> robot_full_struct(Array<float3> body_map, Array<float3> target_map,
                                                               Array<LegDimensions> legs) {
    // orientation list with the reference's own quaternion helpers, restated
    // (quatFromVectAngle unified_math_cuda.cu.h:48-57, qtMultiply :40-46)
    auto from_axis = [](float ax, float ay, float az, float angle) {
        const float s = sinf(angle / 2), c = cosf(angle / 2);
        const float mag = sqrtf(ax * ax + ay * ay + az * az);
        return float4{s, c * ax / mag, c * ay / mag, c * az / mag};
    };
    auto mul = [](float4 a, float4 b) {
        float4 r;
        r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
        r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
        r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
        r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
        return r;
    };
    const float pI = 3.14159265358979323846264338327950288419716939937510582097f;
    std::vector<float> quats;
    const float4 q_init = from_axis(0, 0, 1, 0);
    for (int r = 0; r <= 2; r++) {
        const float roll = -pI / 8 + (pI / 8 - -pI / 8) * ((float)r / 2.f);
        const float4 q_roll = mul(from_axis(1, 0, 0, roll), q_init);
        for (int p = 0; p <= 2; p++) {
            const float pitch = -pI / 8 + (pI / 8 - -pI / 8) * ((float)p / 2.f);
            const float4 q_pitch = mul(from_axis(0, 1, 0, pitch), q_roll);
            for (int y = 0; y <= 4; y++) {
                const float yaw = 0 + (pI / 2 - 0) * ((float)y / 4.f);
                const float4 q = mul(from_axis(0, 0, 1, yaw), q_pitch);
                quats.insert(quats.end(), {q.x, q.y, q.z, q.w});
            }
        }
    }
    std::vector<uint8_t> accepted(body_map.length);
    float ms = 0.f;
    lrm_compat_detail::check(lrm_positionability(&body_map.elements->x, body_map.length, &target_map.elements->x,
                                                 target_map.length, legs.elements, legs.length, quats.data(),
                                                 quats.size() / 4, /*reference_culls=*/1, accepted.data(), &ms),
                             "robot_full_struct");
    size_t n = 0;
    for (uint8_t a : accepted) n += a;
    Array<float3> out_body{n, new float3[n ? n : 1]};
    Array<int> out_count{n, new int[n ? n : 1]};
    size_t k = 0;
    for (size_t i = 0; i < body_map.length; i++)
        if (accepted[i]) {
            out_body.elements[k] = body_map.elements[i];
            out_count.elements[k] = 3;
            k++;
        }
    return std::make_tuple(out_body, out_count);
}

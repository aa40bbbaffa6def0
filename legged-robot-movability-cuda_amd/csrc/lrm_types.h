// lrm_types.h -- host/device shared plain-old-data of the MI355X reach/distance path.
#pragma once
#include <stdint.h>
#include "../../include/lrm.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LRM_HD __host__ __device__ __forceinline__
#else
#define LRM_HD inline
#endif

// lrm_fresh(v): the same object through a pointer the optimiser cannot see through.  Used on the
// by-value kernel argument (kernarg segment, constant address space): loads through the result
// are re-issued as s_load at the point of use (the scalar cache holds the segment) instead of
// being hoisted to the top of the kernel, where ~100 live constants overflow the SGPR file and
// come back as v_readlane spills in the hot loop.  `v` MUST live in constant/global memory.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ const T& lrm_fresh(const T& v) {
    const __attribute__((address_space(4))) T* p = (const __attribute__((address_space(4))) T*)&v;
    asm volatile("" : "+s"(p));
    return *(const T*)p;
}
template <class T>
__device__ __forceinline__ const T& lrm_kernarg(unsigned offset) {
    const __attribute__((address_space(4))) char* p =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const T*)(p + offset);
}
#else
template <class T>
LRM_HD const T& lrm_fresh(const T& v) { return v; }
template <class T>
LRM_HD const T& lrm_kernarg(unsigned) { // device only: this is what the host pass of a kernel body sees, never run
    return *reinterpret_cast<const T*>(sizeof(T));
}
#endif

#define LRM_N_CIRCLES 4  // circles.cu.h:8-14 MAX_CIRCLES
#define LRM_N_CORNERS 10 // circles.cu.h:15 MAX_INTERSECT

// Circle, HeaderCPP.h:9-15, widened to 16 B so that one ds_read_b128 fetches it.
// attract: 1.0f = the point must lie inside, 0.0f = outside.
struct alignas(16) LrmCircle {
    float x, y, r, attract;
};

// Everything the per-point code needs that depends only on (leg, body orientation).
// The reference rebuilds all of it for every point (insert_circles circles.cu.h:337-383,
// insert_intersecv2 :417-476, rotate_leg_data one_leg_global.cu:48-60, the sincosf of
// coxa_pitch / body_angle); here the host computes it once per (leg, quaternion) with the
// reference's own expressions and libm calls, so the values are the same floats.
struct LrmCompiledLeg {
    // circle list for region (upper, fully_extended): lists[upper*2 + fe][0..3]
    LrmCircle lists[4][LRM_N_CIRCLES];
    // corner points that survive the joint-limit filter, in reference order
    float corner_x[LRM_N_CORNERS];
    float corner_y[LRM_N_CORNERS];
    int32_t n_corners;
    // qtRotate(qtInvert(q), .) and qtRotate(q, .) coefficient sums, row-major 3x3:
    //   out.x = 2*(m[0]*x + m[1]*y + m[2]*z) + x   (unified_math_cuda.cu.h:13-27)
    float inv_rot[9];
    float fwd_rot[9];
    float cos_body, sin_body;           // sincosf(-body_angle)   one_leg_global.cu:62-67
    float body;                         // one_leg.cu:13
    float cos_pitch, sin_pitch;         // sincosf(-coxa_pitch)   one_leg.cu:19
    float cos_pitch_rev, sin_pitch_rev; // sincosf(+coxa_pitch)   one_leg.cu:17
    float coxa_length;                  // one_leg.cu:172
    float max_coxa, min_coxa;           // one_leg.cu:305-306
    float mega_hi, mega_lo;             // max_coxa + PI/2, min_coxa - PI/2  one_leg.cu:219-220
    float coxa_mid;                     // (max_coxa + min_coxa)/2           one_leg.cu:229
    float region_mid;                   // circles.cu.h:52-54
    float full_sat[2];                  // circles.cu.h:68, indexed by UpperRegion
    float reach_r2_max;                 // (body-frame) squared radius beyond which nothing is reachable
    // ---- constants of the filtered (LRM_MODE_FAST) evaluation: lrm_point_fast.h ----
    // squared-domain validity test of circle lists[k][i]:  valid <=> sg*(m - T) < 0 with
    // m = |p - c|^2;  T = (r + margin)^2, sg = +1 (attractive) or T = (r - margin)^2, sg = -1;
    // g = 2*(r + margin) converts a distance band into a band on m.
    struct alignas(16) FastCircle {
        float T, sg, g, pad;
    } flists[4][LRM_N_CIRCLES];
    // direction tests "angle > C" as cross products: (cos C, sin C) for
    // C = region_mid, full_sat[0], full_sat[1], max_coxa, min_coxa
    float dir_cos[5], dir_sin[5];
    float fast_scale;                   // max over circles of |cx| + |cy| + r (mm)
    int32_t fast_ok;                    // 0: this leg must take the strict path everywhere
    int32_t n_ucorners;                 // corner points with exact duplicates removed
    float ucorner_x[LRM_N_CORNERS];
    float ucorner_y[LRM_N_CORNERS];
    // find_region as a table: 2 bits (upper * 2 + fully_extended) per pattern of the sign bits of
    // (t_mid, t_s0, t_s1, y) -- pattern bit k set = value k negative; see lrm_region_from_signs
    uint32_t region_lut;
    // ---- lean reach filter (lrm_point_fast.h): everything below is decision-only data ----
    // body-frame point -> coxa frame in one affine map (row-major 3x4, FMA-evaluated):
    //   global: Rp * (Rz * (Rq * p) - (body,0,0));   pair: Rp * (Rz * t - (body,0,0)), t = target - body
    float aff_global[12];
    float aff_pair[12];
    float grav_row[3];                  // x row of Rz * Rq: the "gravity side" test of reachable_rotate_leg
    float pad2_[1];
    // bounding sphere (centre relative to the body position, squared radius with slack) of everything
    // reachable_rotate_leg can accept for THIS leg: see lrm_compile.cpp.  The pair kernels skip a batch
    // of footholds for a leg when none of them is inside.
    float pair_center[3];
    float pair_r2;
    // one 16-byte record per circle of a list: v = m * gs + c with m = |p - centre|^2,
    // gs = sg / (2 (r + margin)), c = -T * gs: the point is valid <=> v < 0, and |v| is its
    // distance (mm) to that decision boundary
    struct alignas(16) LeanCircle {
        float x, y, gs, c;
    } lean[4][LRM_N_CIRCLES];
    float band_base, band_slope;        // band = band_base + band_slope * (|px| + |py| + |pz|), see lrm_point_fast.h
    // ---- lean distance filter ----
    // Per circle i of a list: the point-validity record, r, and for each other circle j the
    // "arc" record that answers "is the clamp point of i valid for j" without building the clamp
    // point:  val_ij = P * ((p - c_i) . e_ij / |p - c_i|) + Q  (mm, < 0 = valid), with
    // e_ij = c_i - c_j, P = 2 r_i gs_j, Q = (|e_ij|^2 + r_i^2 - T_j) gs_j.  (ex, ey) hold P * e_ij.
    // 64 bytes per circle, four ds_read_b128: {x, y, gs, c} {r, arc0} {arc1, arc2.ex} {arc2.ey, arc2.Q, -, -}
    struct alignas(16) DistCircle {
        float x, y, gs, c;
        float r;
        struct Arc {
            float ex, ey, Q;
        } arc[3];
        float pad[2];
    } dist_tab[4][LRM_N_CIRCLES];
    LrmCircle corner_tab[LRM_N_CORNERS]; // de-duplicated corner points as zero-radius circles
    float band_q;                        // LRM_BAND_DIST * 2 * fast_scale: clamp points live on the circles
    float pad3_[3];
};

// ---- tolerance mode (LRM_MODE_TOL, lrm_point_tol.h) -------------------------------------------
// Everything the contract-tolerance evaluation reads, compiled from an LrmCompiledLeg (lrm_compile_tol).
// A separate, smaller block: it travels by value in the kernarg segment next to nothing else.
#define LRM_TOL_FEATS (4 * LRM_N_CIRCLES + LRM_N_CORNERS)
struct LrmTolLeg {
    // One record per circle of a region list (48 B, three ds_read_b128):
    //   {x, y, gs, c}  point validity as in LrmCompiledLeg::LeanCircle: v = |p - centre|^2 * gs + c, valid <=> v < 0
    //   {r, mx, my, chw}  the clamp point of this circle (radial projection of p) is valid for the other
    //       three circles <=> its direction lies on ONE arc of the circle: u . (mx, my) >= chw
    //       (chw = 2: never, chw = -2: always); evaluated as w = (p - centre) . (mx, my) - chw * |p - centre|
    //   {bw}  doubt band of that test: |w| < bw * |p - centre| + tie band
    struct alignas(16) Circle {
        float x, y, gs, c;
        float r, mx, my, chw;
        float bw, pad[3];
    } circ[4][LRM_N_CIRCLES];
    // clamp targets by candidate number: the circles of list k at [k*4 + i], the corner points (radius 0) from
    // [16] on: {x, y, r, -}
    LrmCircle feat[LRM_TOL_FEATS];
    float aff[12];        // body-frame point -> coxa frame (= LrmCompiledLeg::aff_global)
    float back[9];        // coxa-frame vector -> output frame: Rq * Rz(+body_angle) * Rp(+coxa_pitch), row-major
    float yaw_cs[4];      // cos/sin of max_coxa, cos/sin of min_coxa
    float dir_cos[3], dir_sin[3]; // region rays (region_mid, full_sat[0], full_sat[1])
    uint32_t region_lut;
    float coxa_length;
    float band_base, band_slope; // decision band (mm) = band_base + band_slope * |p|_1, as the lean reach filter
    float r_outer;        // every clamp target lies within r_outer of the femur joint (femur + tibia)
    int32_t n_corners;    // corner points in feat[16..], near-duplicates removed
    int32_t tol_ok;       // 0: this leg must use LRM_MODE_FAST
    float pad_[2];
};

// ---- plane table with deferred decisions (lrm_toltab.cpp; lrm_point_tol.h: lrm_tol_plane_tab) -------------------
// Which clamp target wins (and whether the point is valid) is piecewise constant over the leg's meridian plane, so a
// grid over plane coordinates (x = abscissa - coxa_length, z) can answer most plane evaluations.  (A first generation,
// round 2, stored ONE answer per cell and left 7 % of the evaluations unanswered.)  Here a cell names up to two clamp targets
// that can win anywhere in the cell and the one circle (if any) whose point validity is open over the cell; the
// per-point code evaluates just those (a "reduced" lrm_tol_plane: same arithmetic on fewer operands, so the winner's
// vector is the same float for float).  Everything else -- the region, the other circles' validity, the other clamp
// targets, the arc tests of targets that are valid all over the cell -- is decided per cell on the host with
// Lipschitz bounds.  With 16 mm cells refined once to 4 mm, 99.3 % of the plane evaluations of the config-2 cloud
// are answered (the first-generation table: 92.8 %); the rest go to the bit-exact fix-up like any doubtful point.
//   layout: LrmTolTabHeader | uint16 cells[]: per grid (inner, outer) coarse[LRM_TT_N^2] and fine[LRM_TT_SUB^2 * max(n_fine, 1)],
//           then the inner grid's bound[LRM_TT_NB^2] (32 bits each, at an even index)
//   coarse: bit 15 set: refined, bits 0-14 = fine block;  else a cell code
//   bound:  one entry per BOUND CELL = 2 x 2 coarse cells (32 mm) of the inner grid, {d0 (IEEE half), gx (int8), gz (int8)}:
//           max(0, d0 + lb_unit (gx sx + gz sz)) is a LOWER BOUND (mm) of the in-plane part sqrt(du^2 + dz^2) of a yaw candidate's
//           distance for every plane point of sub-cell (sx, sz), 16 x 16 per bound cell -- 0 wherever a point may be valid.  The
//           per-point code orders the two yaw candidates by w^2 + bound^2, takes the reach flag from the first and skips the second
//           when the first one's distance is below the other's bound (the round-2 bound, "beyond the outer circle", left the second
//           evaluation to 26 % of the points of a random cloud, that is to every wave; this one to 0.2 %).  The kernel keeps the
//           inner grid's bounds (16 KB) in LDS: a wave's scattered look-up costs 8 cycles of its CU there, 42 in a table the L1
//           holds and 110-140 in one of 128-512 KB (tools/gather_rates.hip).
//   code (15 bits): target A (row, 5 bits) | target B (row, 5 bits) << 5 | validity row (5 bits) << 10;
//                   LRM_TT_UNANSWERED = 0x7fff: no answer (rows 31 never exist)
//   rows[]:  a clamp target {x, y, r, corner | mx, my, chw, bw}: `corner` = 3e38 for a corner point (it only competes when
//            the point is invalid: one_leg.cu:109-116), 0 for a circle; the arc record as LrmTolLeg::Circle, with
//            (1, 0, -2, 0) = "valid all over the cell".  Row 0 is NONE (never valid): the B slot of a one-target cell.
//   vrows[]: point validity v = |p - (x, y)|^2 gs + c (valid <=> v < 0); rows 0 / 1 are the constants false / true.
#define LRM_TT_N 128           // coarse cells per axis, both grids
#define LRM_TT_OFF 64.0f       // LRM_TT_N / 2: cell index = floor(coordinate / cell size + LRM_TT_OFF)
#ifndef LRM_TT_SUB
#define LRM_TT_SUB 16          // sub-cells per axis of a refined cell: a fine block is LRM_TT_SUB^2 uint16
#endif
#define LRM_TT_H_INNER 16.0f   // inner grid: 16 mm cells (1 mm refined) over +-1024 mm around the femur joint
#define LRM_TT_H_OUTER 128.0f  // outer grid: 128 mm cells (8 mm refined) over +-8192 mm: points the inner grid does not cover
#define LRM_TT_NB 64          // bound cells per axis (two coarse cells each)
#define LRM_TT_UNANSWERED 0x7fffu
#define LRM_TT_MAX_ROWS 31
struct alignas(16) LrmTabRow {
    float x, y, r, corner;
    float mx, my, chw, bw;
};
struct alignas(16) LrmTabVRow {
    float x, y, gs, c;
};
struct LrmTolTabHeader {
    uint32_t n_fine[2];     // refined cells of the inner / outer grid (2 LRM_TT_SUB^2 bytes each)
    uint32_t coarse_off[2]; // uint16 index, from the end of this header, of each grid's coarse array
    uint32_t fine_off[2];   // ... of each grid's fine blocks (block 0 of a grid without refined cells is a spare)
    float inv_h[2];         // 1 / cell size (mm)
    uint32_t bound_off[2];  // ... (even) of the inner grid's array of 32-bit bounds (both entries: the outer grid has none)
    float lb_unit;          // the unit of a bound's gradient bytes: (bound cell / 16) / 64 mm
    float band_max_outer;   // the outer grid holds for points whose decision band (mm) is at most this (|p|_1 <= 16384 mm)
    float band_max;         // the inner grid: ... at most this (|p|_1 <= 4096 mm)
    float far_limit;        // a point with max(r + coxa_length, |z|) below this has both plane points on the inner grid
    uint32_t n_rows, n_vrows;
    LrmTabRow rows[32];
    LrmTabVRow vrows[32];
};

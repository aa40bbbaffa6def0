// lrm_point_tol.h -- contract-tolerance evaluation (LRM_MODE_TOL) of reach + distance.
//
// BASELINE.json's contract for this path is "reach mask bit-exact, distance field within 1e-5 relative".
// LRM_MODE_STRICT / LRM_MODE_FAST deliver tolerance 0 on the distance field and pay for it: two FP64
// sincosf, one FDLIBM atan2f, IEEE divisions and square roots, the strict clamp of the winner and strict norms
// for BOTH yaw candidates -- about 1000 VALU instructions per point.  This mode keeps
// the DECISIONS of the reference (so the mask stays bit-exact and the vector lands on the same boundary
// feature) and computes the VALUES with plain FP32 FMA arithmetic and v_rsq_f32:
//
//  * no trigonometry at all per point: a yaw candidate rotates the point either onto its own meridian
//    plane ((cos, sin) = +-(x, y)/r from one v_rsq_f32) or onto a yaw-limit plane ((cos, sin) are leg constants);
//    every comparison of an angle with a leg constant is the sign of a cross / dot product with that
//    constant's direction (finish_finding_closest one_leg.cu:215-230 compares atan2f values);
//  * the two candidates of distance_circles (one_leg.cu:321-341) are the SAME configuration when one of them is
//    "mega-saturated" onto the other (|yaw| within pi/2 - limit of the x axis): one plane evaluation instead of
//    two (the strict modes must run both: the reference's pick between them is rounding noise, but both picks
//    are the same vector to 1e-7);
//  * the yaw-limit alternative (one_leg.cu:258-274) is w * (-sin, cos) of the nearer limit plane, w being the
//    cross product already taken for the sector test;
//  * in the plane (multi_circle_clamp one_leg.cu:91-145): squared distances rank the candidates (no sqrt
//    for the corner points), "the clamp point of circle i passes the other three circles" is ONE dot
//    product against the arc of circle i that is valid (LrmTolLeg::Circle, built and verified on the host),
//    and only the winner's clamp is formed: delta = (p - c) * (1 - r / |p - c|).
//
// Every decision is taken with a conservative band (mm) exactly as in lrm_point_fast.h; a point with ANY
// decision inside its band is reported in `doubt` and its outputs are not used: the kernel queues it and a
// second, small launch re-evaluates the queue with the bit-exact filtered code (lrm_kernels.hip,
// tol_fixup_kernel).  So for every point the result is either the bit-exact one or within the tolerance
// of it (tests/test_tol_cpu.py, tests/test_gpu_tol.py state and check the metric).
//
// Error model (u = 2^-24, S = |p|_1 + body + fast_scale): the coxa-frame point comes from one FMA affine map,
// <= 22u S away from the strict chain (as the lean reach filter); the plane abscissa sgn * r from v_rsq_f32
// (1 ulp) is <= 8u r from the strict x cos(a) - y sin(a); squared distances by FMA carry 2u relative.  The
// decision band is the lean reach filter's (LRM_BAND * S-like, >= 4.5x the worst case); ties between two
// distances taken from the SAME plane point move by at most twice the point's own error, so the tie band is a
// quarter of the decision band (LRM_TOL_TIE; still >= 2x the worst case).  Empirical margin
// (profiles/r02_tol_band_margin.txt): no mismatch on 3e6 evaluations with the tie band 80x or the decision band 65x
// smaller than the values in use.
#pragma once
#include "lrm_point_fast.h"

#ifndef LRM_TOL_AMP2
#define LRM_TOL_AMP2 64.0f // (largest accepted r / |p - c| of the winning clamp)^2
#endif
// Points closer than this to the coxa axis go to the bit-exact code: the yaw direction (x, y) / r turns every rounding of
// (x, y) into a displacement of the clamp target that is (target radius) / r times larger -- for the reference just as
// much (its own result moves by 2e-3 mm when such an input moves by one ulp).  With 8 mm the randomised campaign
// (tools/stress_tol.py, cloud "coxa_axis") reached 7.4e-6 of the 1e-5 bound at r = 9.1 mm; with 16 mm 3.2e-6.
#ifndef LRM_TOL_RMIN
#define LRM_TOL_RMIN 16.0f
#endif
#ifndef LRM_TOL_CAND_UNROLL
#define LRM_TOL_CAND_UNROLL 2 // 2: the candidate evaluation is instantiated twice; 1: one copy executed twice
#endif
#ifndef LRM_TOL_CIRCLE_UNROLL
#define LRM_TOL_CIRCLE_UNROLL 2
#endif
// all lanes of the wave agree (device) / this point (host)
#if defined(__HIP_DEVICE_COMPILE__)
#define LRM_TOL_ALL(c) (__all(c))
#define LRM_TOL_ANY(c) (__any(c))
#else
#define LRM_TOL_ALL(c) (c)
#define LRM_TOL_ANY(c) (c)
#endif
#ifndef LRM_TOL_TIE
#define LRM_TOL_TIE 0.25f
#endif

// doubt bits (host: statistics; any bit sends the point to the bit-exact re-evaluation).  On the device they
// all collapse to one bit, so that the compiler ORs lane masks on the scalar unit instead of materialising and
// OR-ing one VGPR constant per test.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LRM_TOL_STATS)
#define LRM_TD_YAW 1u
#define LRM_TD_REGION 1u
#define LRM_TD_CLAMP 1u
#define LRM_TD_TIE 1u
#define LRM_TD_NONE 1u
#define LRM_TD_LIMIT 1u
#define LRM_TD_PICK 1u
#define LRM_TD_AMBIG 1u
#else
#define LRM_TD_YAW 1u      // yaw within the band of a sector boundary (limit, limit +- pi/2), or the point near the coxa axis
#define LRM_TD_REGION 2u   // find_region or point-in-circle decision inside the band
#define LRM_TD_CLAMP 4u    // clamp-point validity inside its band
#define LRM_TD_TIE 8u      // the two nearest clamp targets tie
#define LRM_TD_NONE 16u    // no clamp target at all (the reference then returns the raw point), or an ill-conditioned clamp
#define LRM_TD_LIMIT 32u   // yaw-limit alternative ties with the in-plane distance, or the two limits tie
#define LRM_TD_PICK 64u    // the two yaw candidates tie
#define LRM_TD_AMBIG 0x100u // table kernel: the cell of a candidate's plane point carries no answer
#endif
#define LRM_TD_SECOND 0x10000u // statistic only: the second candidate could not be pruned by its lower bound

#ifndef LRM_TOL_DIET
#define LRM_TOL_DIET 1
#endif
// min(|a|, |b|, c) and max(a, |b|) as ONE instruction each: fminf(fabsf(a), fabsf(b)) compiles to a canonicalising
// v_max_f32 per operand in front of the v_min_f32 (IEEE mode), six instructions for a four-way minimum instead of two.
// NaN operands are ignored, exactly as fminf / fmaxf ignore them.
#if defined(__HIP_DEVICE_COMPILE__) && LRM_TOL_DIET
__device__ __forceinline__ float lrm_min3_aa(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float lrm_max_a(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
LRM_HD float lrm_min3_aa(float a, float b, float c) { return fminf(fminf(fabsf(a), fabsf(b)), c); }
LRM_HD float lrm_max_a(float a, float b) { return fmaxf(a, fabsf(b)); }
#endif

struct LrmTolTables {
    const LrmTolLeg::Circle* circ; // [16]
    const LrmCircle* feat;         // [LRM_TOL_FEATS]
};

// kind of a yaw candidate from the sign pattern of (w_max, u_max, w_min, u_min) (bit set = negative):
// code 0: the point's own meridian plane (+r), 1: the opposite one (-r), 2: the max-limit plane, 3: the min-limit
// plane.  finish_finding_closest (one_leg.cu:215-230) with `angle` = yaw of the point (direct) or yaw -+ pi (flip):
//   inside [min, max]                 -> unsaturated: own plane (direct) / opposite plane (flip)
//   (max, max + pi/2] / [min - pi/2, min) -> clamped to that limit
//   beyond                            -> "mega": sat = angle -+ pi: opposite plane (direct) / own plane (flip)
LRM_HD constexpr uint32_t lrm_tol_kind(uint32_t pat) { // 0 inside, 2 / 3 clamped to max / min, 1 beyond
    return ((pat & 5u) == 1u) ? 0u : (((pat & 3u) == 0u) ? 2u : (((pat & 12u) == 4u) ? 3u : 1u));
}
LRM_HD constexpr uint32_t lrm_tol_lut(bool flip) {
    uint32_t lut = 0;
    for (uint32_t pat = 0; pat < 16; pat++) {
        uint32_t k = lrm_tol_kind(flip ? (pat ^ 15u) : pat);
        if (flip && k < 2u) k ^= 1u; // the flipped candidate sees the mirrored point: own <-> opposite plane
        lut |= k << (2 * pat);
    }
    return lut;
}

// eval_plane_circles<DIST> + multi_circle_clamp (one_leg.cu:91-145, :167-208) at plane point (u, z):
// (du, dz) = point - nearest valid clamp target; `valid` = the point passes its region's four circles.
LRM_HD void lrm_tol_plane(const LrmTolLeg& L, const LrmTolTables T, float u, float z, float band, float tau,
                          float& du, float& dz, bool& valid, uint32_t& doubt) {
    const float x = u - L.coxa_length;
    // region (circles.cu.h:48-78), as lrm_plane_dist_fast
    const float t_mid = __builtin_fmaf(L.dir_cos[0], z, -(L.dir_sin[0] * x));
    const float t_s0 = __builtin_fmaf(L.dir_cos[1], z, -(L.dir_sin[1] * x));
    const float t_s1 = __builtin_fmaf(L.dir_cos[2], z, -(L.dir_sin[2] * x));
    const uint32_t reg = lrm_region_from_signs(L.region_lut, t_mid, t_s0, t_s1, z) * LRM_N_CIRCLES;
#if LRM_TOL_DIET
    const float macc = lrm_min3_aa(t_s1, t_s1, lrm_min3_aa(t_mid, t_s0, lrm_max_a(x, z)));
#else
    const float macc = fminf(fminf(fabsf(t_mid), fabsf(t_s0)), fminf(fabsf(t_s1), fmaxf(x, fabsf(z))));
#endif
    float vacc = -3.0e38f, cacc = 3.0e38f;
    uint32_t lo = 0x7f80000fu, hi = 0x7f80000fu; // the two smallest keys: squared distance bits | candidate number
    const LrmTolLeg::Circle* ct = T.circ + reg;
#pragma unroll LRM_TOL_CIRCLE_UNROLL
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        const LrmTolLeg::Circle c = ct[i];
        const float vx = x - c.x, vy = z - c.y;
        const float m = __builtin_fmaf(vy, vy, vx * vx);
        // validity of the point itself (mm, < 0 = valid).  Only the LARGEST value decides: below -band every
        // circle is passed with margin, above +band one circle fails with margin whatever the others say.
        vacc = fmaxf(vacc, __builtin_fmaf(m, c.gs, c.c));
        const float rs = LRM_FAST_RSQ(m);
        const float mag = m * rs;
        const float d = c.r - mag;
        // is the clamp point valid for the other three circles: one arc test, scaled by |p - c|.  (At the centre
        // of the circle, where the reference switches to a fixed direction (one_leg.cu:54-58), |w| <= 3 |p - c|
        // is inside the band; should that clamp win all the same, LRM_TOL_AMP2 below catches it.)
        const float w = __builtin_fmaf(-c.chw, mag, __builtin_fmaf(vx, c.mx, vy * c.my));
        cacc = fminf(cacc, __builtin_fmaf(-c.bw, mag, fabsf(w)));
        // an invalid clamp ranks as +huge: max(d^2, -1e30 w)
        const float kf = fmaxf(d * d, w * -1.0e30f);
        const uint32_t k = (lrm_f2u(kf) & ~15u) | (uint32_t)i;
        hi = lrm_umed3(lo, hi, k);
        lo = lo < k ? lo : k;
    }
    valid = vacc < 0.f;
    // corner points only matter when the point itself is invalid (one_leg.cu:109-116)
#if LRM_TOL_DIET
    const uint32_t lo_circ = lo, hi_circ = hi; // ranking without the corner points: selected again after the loop
#else
    const uint32_t keep = valid ? 0u : 0xffffffffu;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const int n_corners = __builtin_amdgcn_readfirstlane(L.n_corners); // a scalar also when the leg block lives in LDS
#else
    const int n_corners = L.n_corners;
#endif
#pragma unroll
    for (int i = 0; i < LRM_N_CORNERS; i++) {
        if (i < n_corners) { // wave-uniform
            const LrmCircle c = T.feat[4 * LRM_N_CIRCLES + i];
            const float vx = x - c.x, vy = z - c.y;
            const float m = __builtin_fmaf(vy, vy, vx * vx);
            uint32_t k = (lrm_f2u(m) & ~15u) | (uint32_t)(LRM_N_CIRCLES + i);
#if !LRM_TOL_DIET
            k = (k & keep) | (0x7f80000fu & ~keep);
#endif
            hi = lrm_umed3(lo, hi, k);
            lo = lo < k ? lo : k;
        }
    }
#if LRM_TOL_DIET
    lo = valid ? lo_circ : lo;
    hi = valid ? hi_circ : hi;
#endif
    const float lo2 = lrm_u2f(lo & ~15u), hi2 = lrm_u2f(hi & ~15u);
    // |b - a| < tau  <=>  b^2 < a^2 + tau (2a + tau); the keys dropped 4 mantissa bits (< 2e-6 relative)
    const float a = LRM_FAST_SQRT(lo2);
    const float tie_thr = __builtin_fmaf(lo2, 4.0e-6f, __builtin_fmaf(tau, __builtin_fmaf(2.0f, a, tau), lo2));
    // the winner's clamp: delta = (p - c)(1 - r / |p - c|); a corner point is a circle of radius 0
    const uint32_t win = lo & 15u;
    const LrmCircle f = T.feat[win < LRM_N_CIRCLES ? reg + win : 3 * LRM_N_CIRCLES + win];
    const float vx = x - f.x, vy = z - f.y;
    const float m = __builtin_fmaf(vy, vy, vx * vx);
    const float s = __builtin_fmaf(-f.r, LRM_FAST_RSQ(m), 1.0f);
    du = vx * s;
    dz = vy * s;
    uint32_t lu = 0;
    lu |= (!(macc > band) || !(fabsf(vacc) > band)) ? LRM_TD_REGION : 0u;
    lu |= !(cacc > tau) ? LRM_TD_CLAMP : 0u;
    lu |= !(hi2 > tie_thr) ? LRM_TD_TIE : 0u;
    // Near the centre of the winning circle the clamp point amplifies the rounding of the plane point by
    // r / |p - c| (the reference's own result is just as sensitive: the two would differ by amplified noise):
    // beyond 4x the point goes to the bit-exact code.  A corner point has r = 0: only m = 0 (0 * inf) is caught.
    lu |= (!(lo2 < 1.0e30f) || !(m * LRM_TOL_AMP2 > f.r * f.r)) ? LRM_TD_NONE : 0u;
    doubt |= lu;
#if defined(LRM_TOL_TRACE)
    printf("  plane u %.5f z %.5f x %.5f reg %u valid %d lo %08x hi %08x (lo2 %.6f hi2 %.6f) win %u feat(%.4f %.4f r %.4f) m %.5f d(%.5f %.5f) lu %x\n",
           u, z, x, reg / 4, (int)valid, lo, hi, lo2, hi2, win, f.x, f.y, f.r, m, du, dz, lu);
#endif
}

// ---- plane evaluator of lrm_dist_tol_t: the full evaluation above ----
struct LrmTolPlaneFull {
    static constexpr bool kSkipSame = true; // a wave whose lanes all have one configuration skips the second evaluation
    const LrmTolLeg& L;
    LrmTolTables T;
    LRM_HD void operator()(float u, float z, float band, float tau, float& du, float& dz, bool& valid, uint32_t& doubt) const {
        lrm_tol_plane(L, T, u, z, band, tau, du, dz, valid, doubt);
    }
};

// ---- plane table with deferred decisions (lrm_types.h LrmTolTabHeader, lrm_toltab.cpp) ----
struct LrmTolTabView {
    const LrmTabRow* rows;   // [32]
    const LrmTabVRow* vrows; // [32]
    const uint16_t* cells;   // the grids' coarse, fine and bound arrays (behind the header)
    const uint32_t* bound_inner; // the inner grid's bounds where the caller keeps them (the kernel: its LDS copy; the host: the table's own)
    float band_max, band_max_outer, far_limit;
    float inv_h[2], lb_unit;
    uint32_t coarse_off[2], fine_off[2];
    float r_outer; // every clamp target and every valid point lies within this of the femur joint (LrmTolLeg::r_outer)
};
// rows / vrows / bound_inner: where the caller keeps them (the kernel: its LDS copies; the host: the table's own)
LRM_HD LrmTolTabView lrm_toltab_view(const uint8_t* tab, const LrmTabRow* rows, const LrmTabVRow* vrows, const uint32_t* bound_inner, float r_outer) {
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab);
    return LrmTolTabView{rows, vrows, reinterpret_cast<const uint16_t*>(tab + sizeof(LrmTolTabHeader)), bound_inner,
                         hd->band_max, hd->band_max_outer, hd->far_limit, {hd->inv_h[0], hd->inv_h[1]}, hd->lb_unit, {hd->coarse_off[0], hd->coarse_off[1]},
                         {hd->fine_off[0], hd->fine_off[1]}, r_outer};
}
// the table's own copy of the inner grid's bounds (host callers)
inline const uint32_t* lrm_toltab_bound_inner(const uint8_t* tab) {
    const LrmTolTabHeader* hd = reinterpret_cast<const LrmTolTabHeader*>(tab);
    return reinterpret_cast<const uint32_t*>(tab + sizeof(LrmTolTabHeader) + 2 * (size_t)hd->bound_off[0]);
}
// cells[i] of the table (global memory).  Device: "uniform base + 32-bit offset" addressing (global_load_ushort / _dword with
// an SGPR base) instead of a 64-bit address per lane.
#if defined(__HIP_DEVICE_COMPILE__)
LRM_HD uint32_t lrm_tt_cell(const uint16_t* cells, uint32_t i) {
    typedef const __attribute__((address_space(1))) char* GP;
#if defined(LRM_TAB_EXP_ONE_LINE) // timing experiment (wrong results): every look-up reads the same cache line
    asm volatile("v_and_b32 %0, 31, %0" : "+v"(i));
#endif
    return *(const __attribute__((address_space(1))) uint16_t*)((GP)cells + (i << 1));
}
LRM_HD float lrm_half_bits_to_float(uint32_t h) { // the low 16 bits (v_cvt_f32_f16 reads nothing else)
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}
LRM_HD int lrm_dot_bytes(uint32_t a, uint32_t b) { return __builtin_amdgcn_sdot4((int)a, (int)b, 0, false); } // sum of the four products of signed bytes
#else
LRM_HD uint32_t lrm_tt_cell(const uint16_t* cells, uint32_t i) { return cells[i]; }
LRM_HD float lrm_half_bits_to_float(uint32_t h) { // normal halves and zero only (lrm_toltab.cpp writes nothing else)
    const uint32_t e = (h >> 10) & 31u, f = h & 1023u;
    return e == 0u ? 0.f : lrm_u2f(((h & 0x8000u) << 16) | ((e + 112u) << 23) | (f << 13));
}
LRM_HD int lrm_dot_bytes(uint32_t a, uint32_t b) {
    int t = 0;
    for (int k = 0; k < 4; k++) t += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
    return t;
}
#endif
// Look-up of two plane points (x0, z), (x1, z) of one point: the codes of their cells, or LRM_TT_UNANSWERED, and lower bounds of
// their in-plane distances (0 outside the grids).  The point uses the outer grid when `far` (its plane points may lie beyond the
// inner one); `anyfar`: some lane of the wave does (device) / this point does (host).  Straight-line code: every lane reads one
// coarse and one fine entry per plane point, the two coarse loads are issued together, then the two fine loads; an inner-grid
// lane takes its bounds from G.bound_inner (LDS in the kernel), an outer-grid lane from the outer circle.
// kMixed: the variant for a wave with outer-grid lanes (two instances: written as ONE body with `if (anyfar)` blocks the compiler turns
// the blocks into selects that every wave executes -- 6 instructions per point).  band_doubt: the point's decision band exceeds what its
// grid was built for.
template <bool kMixed>
LRM_HD void lrm_toltab_lookup2(const LrmTolTabView& G, bool far, float band, float x0, float x1, float z,
                               uint32_t& c0, uint32_t& c1, uint32_t& s0, uint32_t& s1, uint32_t& fbase_out, float& lb0, float& lb1, bool& band_doubt) {
    constexpr bool anyfar = kMixed;
    if (!kMixed) far = false;
    const uint16_t* cells = G.cells;
    // position in units of SUB-cells: q = floor(coordinate / cell size * SUB + OFF * SUB); the cell is q >> 4, the sub-cell q & 15
    // (one FMA, one conversion, one shift and one mask per coordinate).  On the inner grid every plane point lies inside (`far`
    // is false when max(r + coxa_length, |z|) < far_limit): positions are positive, truncation is floor, no range test.  In a wave
    // with outer-grid lanes (wave-uniform branch) the grid is chosen per lane, a position is clamped into the grid and the point
    // marked (-> unanswered, bound 0); nan fails the test too (the conversion maps it to 0).  (Choosing the grid per WAVE sent the
    // near points of mixed waves to the outer grid's 8 mm sub-cells: three times as many unanswered, i.e. queued, points.)
    constexpr float kOffS = LRM_TT_OFF * (float)LRM_TT_SUB, kMaxS = (float)(LRM_TT_N * LRM_TT_SUB) - 0.5f;
    float invs = G.inv_h[0] * (float)LRM_TT_SUB;
    uint32_t cbase = G.coarse_off[0], fbase = G.fine_off[0];
    if (anyfar) {
        invs = far ? G.inv_h[1] * (float)LRM_TT_SUB : invs;
        cbase = far ? G.coarse_off[1] : cbase;
        fbase = far ? G.fine_off[1] : fbase;
    }
    float pz = __builtin_fmaf(z, invs, kOffS), p0 = __builtin_fmaf(x0, invs, kOffS), p1 = __builtin_fmaf(x1, invs, kOffS);
    bool out0 = false, out1 = false;
    band_doubt = !(band <= G.band_max);
    if (anyfar) {
        band_doubt = !(band <= (far ? G.band_max_outer : G.band_max));
        const bool oz = !(pz >= 0.f && pz <= kMaxS);
        out0 = oz || !(p0 >= 0.f && p0 <= kMaxS);
        out1 = oz || !(p1 >= 0.f && p1 <= kMaxS);
        pz = fminf(fmaxf(pz, 0.f), kMaxS);
        p0 = fminf(fmaxf(p0, 0.f), kMaxS);
        p1 = fminf(fmaxf(p1, 0.f), kMaxS);
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: a nan position (the coxa origin itself: r = 0 * inf) converts to INT_MIN on x86 where v_cvt_i32_f32 gives 0; the
    // point is in doubt either way, the look-up must only stay inside the table
    if (!(pz == pz)) pz = 0.f;
    if (!(p0 == p0)) p0 = 0.f;
    if (!(p1 == p1)) p1 = 0.f;
    if (!kMixed) { pz = fminf(fmaxf(pz, 0.f), kMaxS); p0 = fminf(fmaxf(p0, 0.f), kMaxS); p1 = fminf(fmaxf(p1, 0.f), kMaxS); }
#endif
    const uint32_t qz = (uint32_t)(int)pz, q0 = (uint32_t)(int)p0, q1 = (uint32_t)(int)p1; // >= 0: truncation is floor
    static_assert(LRM_TT_SUB == 16, "shifts below");
    const uint32_t row = (qz >> 4) * (uint32_t)LRM_TT_N;
    const uint32_t a0 = row + (q0 >> 4), a1 = row + (q1 >> 4); // cell numbers
    c0 = lrm_tt_cell(cells, cbase + a0);
    c1 = lrm_tt_cell(cells, cbase + a1);
    // where the sub-cell sits inside a refined cell's block: the caller resolves (lrm_toltab_resolve) only the candidate it evaluates
    const uint32_t szn = qz & 15u, sx0 = q0 & 15u, sx1 = q1 & 15u;
    const uint32_t sz = szn * (uint32_t)LRM_TT_SUB + fbase;
    s0 = sz + sx0;
    s1 = sz + sx1;
    fbase_out = fbase;
    // The bounds.  Inner grid: one entry per 2 x 2 coarse cells, d0 + unit (gx sx + gz sz) over its 16 x 16 sub-cells, the two
    // products of signed bytes in one v_dot4 (bytes 2 and 3 of the entry), from LDS.  Outer grid: the distance beyond the circle
    // that holds every target and every valid point -- far from the workspace that is nearly the distance itself, and it costs no
    // look-up (bounds of the outer grid's cells, read from global memory, made a far cloud 6 % slower: profiles/r03_ab_far_bounds.txt).
    static_assert(LRM_TT_N == 2 * LRM_TT_NB, "bound cell = 2 x 2 coarse cells = 32 sub-cell units");
    const uint32_t g0 = (qz >> 5) * (uint32_t)LRM_TT_NB + (q0 >> 5), g1 = (qz >> 5) * (uint32_t)LRM_TT_NB + (q1 >> 5); // (in range for outer-grid lanes too)
    const uint32_t zb = ((qz >> 1) & 15u) << 24;
    const uint32_t e0 = G.bound_inner[g0], e1 = G.bound_inner[g1];
    const float t0 = (float)lrm_dot_bytes(e0, zb | (((q0 >> 1) & 15u) << 16)), t1 = (float)lrm_dot_bytes(e1, zb | (((q1 >> 1) & 15u) << 16));
    lb0 = fmaxf(__builtin_fmaf(t0, G.lb_unit, lrm_half_bits_to_float(e0)), 0.f);
    lb1 = fmaxf(__builtin_fmaf(t1, G.lb_unit, lrm_half_bits_to_float(e1)), 0.f);
    if (anyfar) {
        const float zz = z * z;
        const float o0 = fmaxf(LRM_FAST_SQRT(__builtin_fmaf(x0, x0, zz)) - G.r_outer, 0.f);
        const float o1 = fmaxf(LRM_FAST_SQRT(__builtin_fmaf(x1, x1, zz)) - G.r_outer, 0.f);
        lb0 = far ? o0 : lb0;
        lb1 = far ? o1 : lb1;
        c0 = out0 ? (uint32_t)LRM_TT_UNANSWERED : c0; // (not a refined cell: resolves to itself)
        c1 = out1 ? (uint32_t)LRM_TT_UNANSWERED : c1;
        lb0 = out0 ? 0.f : lb0;
        lb1 = out1 ? 0.f : lb1;
    }
}
// The code of a plane point from its coarse entry c and sub-cell place s (lrm_toltab_lookup2): the entry itself, or, of a refined cell,
// the sub-cell's entry in the cell's block -- one more 2-byte load, issued by every lane (the others read the grid's first fine entry).
LRM_HD uint32_t lrm_toltab_resolve(const LrmTolTabView& G, uint32_t c, uint32_t s, uint32_t fbase) {
    const bool r = (c & 0x8000u) != 0u;
    const uint32_t b = ((c & 0x7fffu) << 8) + s; // LRM_TT_SUB^2 = 256 entries per block
    static_assert(LRM_TT_SUB * LRM_TT_SUB == 256, "fine block size");
    const uint32_t f = lrm_tt_cell(G.cells, r ? b : fbase);
    return r ? f : c;
}
// lrm_tol_plane restricted to what the cell's code names: at most two clamp targets, one circle's point validity.
// Same arithmetic as lrm_tol_plane on those operands.  x = abscissa - coxa_length.
// kInfo: also report which of the cell's two rows won (`row`), for the queue records of LRM_MODE_TOL_REL.
template <bool kInfo = false>
LRM_HD void lrm_tol_plane_tab(const LrmTolTabView& G, uint32_t code, float x, float z, float band, float tau,
                              float& du, float& dz, bool& valid, uint32_t& doubt, uint32_t* row = nullptr) {
#if defined(LRM_TAB_EXP_ONE_ROW) && defined(__HIP_DEVICE_COMPILE__) // timing experiment (wrong results): every lane reads the same rows -- no LDS bank conflicts
    uint32_t code_rows = code;
    asm volatile("v_and_b32 %0, 0x421, %0" : "+v"(code_rows));
#else
    const uint32_t code_rows = code;
#endif
    const LrmTabVRow vr = G.vrows[(code_rows >> 10) & 31u];
    const LrmTabRow ra = G.rows[code_rows & 31u], rb = G.rows[(code_rows >> 5) & 31u];
    float vacc;
    {
        const float vx = x - vr.x, vy = z - vr.y;
        vacc = __builtin_fmaf(__builtin_fmaf(vy, vy, vx * vx), vr.gs, vr.c);
    }
    valid = vacc < 0.f;
    const float cpen = valid ? 1.0f : 0.0f; // a corner point only competes when the point is invalid (one_leg.cu:109-116)
    const float ax = x - ra.x, ay = z - ra.y, bx = x - rb.x, by = z - rb.y;
    const float ma = __builtin_fmaf(ay, ay, ax * ax), mb = __builtin_fmaf(by, by, bx * bx);
    const float rsa = LRM_FAST_RSQ(ma), rsb = LRM_FAST_RSQ(mb);
    const float maga = ma * rsa, magb = mb * rsb;
    const float da = ra.r - maga, db = rb.r - magb;
    const float wa = __builtin_fmaf(-ra.chw, maga, __builtin_fmaf(ax, ra.mx, ay * ra.my));
    const float wb = __builtin_fmaf(-rb.chw, magb, __builtin_fmaf(bx, rb.mx, by * rb.my));
    const float cacc = fminf(__builtin_fmaf(-ra.bw, maga, fabsf(wa)), __builtin_fmaf(-rb.bw, magb, fabsf(wb)));
    // an invalid clamp ranks as +huge, and so does a corner point next to a valid point
    const float ka = fmaxf(fmaxf(da * da, wa * -1.0e30f), ra.corner * cpen);
    const float kb = fmaxf(fmaxf(db * db, wb * -1.0e30f), rb.corner * cpen);
    const bool wina = ka <= kb;
    const float lo2 = wina ? ka : kb, hi2 = wina ? kb : ka;
    // |b - a| < tau  <=>  b^2 < a^2 + tau (2a + tau), with the relative allowance of lrm_tol_plane's ranking keys
    // (the winner's distance itself is |r - mag|; when even the winner is an invalid clamp, lo2 is huge and the point in doubt)
    const float a = fabsf(wina ? da : db);
    const float tie_thr = __builtin_fmaf(lo2, 4.0e-6f, __builtin_fmaf(tau, __builtin_fmaf(2.0f, a, tau), lo2));
    const float vx = wina ? ax : bx, vy = wina ? ay : by, rs = wina ? rsa : rsb, r = wina ? ra.r : rb.r;
    const float s = __builtin_fmaf(-r, rs, 1.0f); // 1 - r / |p - c|
    du = vx * s;
    dz = vy * s;
    if (kInfo) *row = wina ? 0u : 0x8000u; // which of the cell's two rows won
    uint32_t lu = 0;
    // An unanswered cell names validity row 31, which lrm_toltab.cpp fills with nan: vacc is nan and fails this test.
    lu |= !(fabsf(vacc) > band) ? LRM_TD_REGION : 0u;
    lu |= !(cacc > tau) ? LRM_TD_CLAMP : 0u;
    lu |= !(hi2 > tie_thr) ? LRM_TD_TIE : 0u;
    // the amplification guard of lrm_tol_plane, |p - c|^2 AMP2 > r^2, as r / |p - c| < sqrt(AMP2): 1 - s < 8
    static_assert(LRM_TOL_AMP2 == 64.0f, "s > 1 - sqrt(AMP2)");
    lu |= (!(lo2 < 1.0e30f) || !(s > -7.0f)) ? LRM_TD_NONE : 0u;
#if !defined(__HIP_DEVICE_COMPILE__)
    lu |= (code == (uint32_t)LRM_TT_UNANSWERED) ? LRM_TD_AMBIG : 0u; // host statistic (the device needs no test of its own: see above)
#endif
    doubt |= lu;
}

inline thread_local unsigned long long lrm_tab_host_seconds = 0; // host statistic: points whose second candidate was evaluated
// The whole evaluation of one point with the table: distance_global + reachability_global (one_leg_global.cu:74-130).
// The flow of lrm_tol_prologue / lrm_tol_candidate / lrm_tol_need_second / lrm_tol_finish, written as one function so
// that a candidate is carried as (code, du, w, dz) -- the rotation back to the coxa frame is formed once, for the
// candidate that wins -- and its plane abscissa is selected from (+-r, uM, um) instead of rotated.
// p: in = the point, out = the distance vector.  Returns the reach / validity flag.  doubt != 0: do not use the outputs.
// kInfo (LRM_MODE_TOL_REL): *info = what the evaluation DECIDED, for the strict replay of the winner's value chain by the fix-up
// (lrm_xtab_replay, lrm_point_xtab.h):  bits 0-14 the cell code of the winning candidate's plane point | 15 its second row won |
// 16-17 the winning candidate's kind (0 own meridian plane, 1 the opposite one, 2 / 3 clamped to the max / min yaw limit) | 18 it is
// the flipped candidate | 19 its vector is the offset from a yaw-limit plane (the alternative of one_leg.cu:258-274 was taken, or a
// clamped candidate collapsed onto it) | 20 the flag.
template <bool kInfo = false>
LRM_HD bool lrm_tab_point(const LrmTolLeg& L, const LrmTolTabView& G, LrmVec3& p, uint32_t& doubt, uint32_t* info = nullptr) {
    const float* a = L.aff;
    const float x = __builtin_fmaf(a[0], p.x, __builtin_fmaf(a[1], p.y, __builtin_fmaf(a[2], p.z, a[3])));
    const float y = __builtin_fmaf(a[4], p.x, __builtin_fmaf(a[5], p.y, __builtin_fmaf(a[6], p.z, a[7])));
    const float z = __builtin_fmaf(a[8], p.x, __builtin_fmaf(a[9], p.y, __builtin_fmaf(a[10], p.z, a[11])));
    // non-finite input: the band is nan/inf and every "> band" test fails closed (doubt)
    const float band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), L.band_slope, L.band_base);
    const float tau = band * LRM_TOL_TIE;
    const float m2 = __builtin_fmaf(y, y, x * x);
    const float rs = LRM_FAST_RSQ(m2);
    const float r = m2 * rs;
    const float cu = x * rs, su = y * rs; // cos / sin of the point's yaw
    const float cM = L.yaw_cs[0], sM = L.yaw_cs[1], cm = L.yaw_cs[2], sm = L.yaw_cs[3];
    const float uM = __builtin_fmaf(x, cM, y * sM), wM = __builtin_fmaf(y, cM, -(x * sM));
    const float um = __builtin_fmaf(x, cm, y * sm), wm = __builtin_fmaf(y, cm, -(x * sm));
    const uint32_t pat = (lrm_f2u(wM) >> 31) | ((lrm_f2u(uM) >> 30) & 2u) | ((lrm_f2u(wm) >> 29) & 4u) |
                         ((lrm_f2u(um) >> 28) & 8u);
    constexpr uint32_t kLutD = lrm_tol_lut(false), kLutF = lrm_tol_lut(true);
    const uint32_t codeD = (kLutD >> (pat << 1)) & 3u, codeF = (kLutF >> (pat << 1)) & 3u;
    const bool inD = (pat & 5u) == 1u, inF = (pat & 5u) == 4u;
    const float ymin = lrm_min3_aa(wm, um, lrm_min3_aa(wM, uM, 3.0e38f));
    uint32_t lu = (!(ymin > band) || !(r > LRM_TOL_RMIN)) ? LRM_TD_YAW : 0u;
    const bool two = codeD != codeF;
    // plane point of a candidate kind: abscissa {r, -r, uM, um}[code], offset {0, 0, wM, wm}[code].  Every `?:` below
    // selects between values that exist already: straight-line v_cndmask code, no branches.
    const bool limD = codeD >= 2u, limF = codeF >= 2u;
    const float wlD = (codeD == 3u) ? wm : wM, wlF = (codeF == 3u) ? wm : wM;
    const float wD = limD ? wlD : 0.f, wF = limF ? wlF : 0.f;
    const float ulD = (codeD == 3u) ? um : uM, ulF = (codeF == 3u) ? um : uM;
    const float urD = lrm_u2f(lrm_f2u(r) ^ (codeD << 31)), urF = lrm_u2f(lrm_f2u(r) ^ (codeF << 31));
    const float uD = limD ? ulD : urD, uF = limF ? ulF : urF;
    // Both plane points lie within max(r + coxa_length, |z|) of the femur joint: one grid for both look-ups.  A wave without
    // outer-grid lanes (wave-uniform) skips everything the outer grid needs.
    const bool far = !(fmaxf(r + L.coxa_length, fabsf(z)) < G.far_limit);
    const bool anyfar = LRM_TOL_ANY(far);
    const float xD = uD - L.coxa_length, xF = uF - L.coxa_length;
    uint32_t cD, cF, sD, sF, fbase;
    float lbD, lbF;
    bool band_doubt;
    if (anyfar) lrm_toltab_lookup2<true>(G, far, band, xD, xF, z, cD, cF, sD, sF, fbase, lbD, lbF, band_doubt);
    else lrm_toltab_lookup2<false>(G, false, band, xD, xF, z, cD, cF, sD, sF, fbase, lbD, lbF, band_doubt);
    lu |= band_doubt ? LRM_TD_YAW : 0u;
    // Which candidate first: the one with the smaller lower bound w^2 + lb^2 of its squared distance (lb: the cell's bound of
    // the in-plane part).  A candidate that may be valid has lb = 0; inside the yaw range (w = 0) its bound is 0 and it goes
    // first -- the reach flag is always taken from the first candidate.  Equal bounds (the two candidates are one
    // configuration): the one inside the range.
    const float bD = __builtin_fmaf(lbD, lbD, wD * wD), bF = __builtin_fmaf(lbF, lbF, wF * wF);
    const bool firstD = inF ? (bD < bF) : (bD <= bF);
    const uint32_t code0 = firstD ? codeD : codeF;
    const float w0 = firstD ? wD : wF, x0 = firstD ? xD : xF;
    const uint32_t cell0 = lrm_toltab_resolve(G, firstD ? cD : cF, firstD ? sD : sF, fbase); // only the candidate that is evaluated
    const float b1 = firstD ? bF : bD;
    const bool lim0 = code0 >= 2u;
    const bool in0 = firstD ? inD : inF;
    // ---- first candidate ----
    float du0, dz0;
    bool valid0;
    uint32_t row0 = 0; // kInfo: cell code | winner bit of the winning candidate
    bool offW = false;
    lrm_tol_plane_tab<kInfo>(G, cell0, x0, z, band, tau, du0, dz0, valid0, lu, &row0);
    if (kInfo) row0 |= cell0;
    // A candidate clamped to a yaw limit whose plane point is valid collapses to the offset from that plane, unless
    // sqrt(du^2 + w^2 + dz^2) rounds to |w| (see lrm_dist_tol_t): q / w^2 above 2^-20 collapses, below 2^-25 stays, between: doubt
    {
        const float q = __builtin_fmaf(du0, du0, dz0 * dz0), w2 = w0 * w0;
        const bool lv = lim0 && valid0, big = q > w2 * 9.6e-7f;
        const bool collapse = lv && big;
        du0 = collapse ? 0.f : du0;
        dz0 = collapse ? 0.f : dz0;
        if (kInfo) offW = collapse;
        lu |= (lv && !big) ? LRM_TD_LIMIT : 0u; // (below 2^-25 the reference keeps the vector: one plane point in 1e6, left to the bit-exact code too)
    }
    const float n0 = __builtin_fmaf(du0, du0, __builtin_fmaf(w0, w0, dz0 * dz0));
    const bool flag = valid0 && in0;
    // ---- yaw-limit alternative (one_leg.cu:258-274) of a valid first candidate: the nearer limit plane wins when it is closer ----
    uint32_t codeW;
    float duW, wW, dzW;
    {
        const float aM = fabsf(wM), am = fabsf(wm);
        const float dl = fminf(aM, am), dl2 = dl * dl;
        const float thr = tau * __builtin_fmaf(2.0f, dl, tau);
        // in doubt: the two distances tie, or the alternative is (or may be) taken and the two limit planes tie
        const bool tie = !(fabsf(n0 - dl2) > thr) || (!(n0 < dl2 - thr) && !(fabsf(aM - am) > tau));
        lu |= (flag && tie) ? LRM_TD_LIMIT : 0u;
        const bool alt = flag && (n0 > dl2);
        const bool useM = aM < am;
        const uint32_t codeL = useM ? 2u : 3u;
        const float wL = useM ? wM : wm;
        codeW = alt ? codeL : code0;
        if (kInfo) offW = offW || alt;
        duW = alt ? 0.f : du0;
        dzW = alt ? 0.f : dz0;
        wW = alt ? wL : w0;
    }
    // ---- the second one only when it can still win: not below its lower bound by more than the tie band ----
    const bool need = two && !flag && !(n0 < b1 - tau * __builtin_fmaf(2.0f, LRM_FAST_SQRT(n0), tau));
    bool infoB = false;
    uint32_t codeB = 0;
#if defined(LRM_TAB_EXP_NO_SECOND) // timing experiment (wrong results): the second candidate is never evaluated
    if (false) {
#else
    if (LRM_TOL_ANY(need)) { // device: when a lane of the wave needs it (3 % of the waves of a random cloud); host: when this point does
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
        lrm_tab_host_seconds++;
#endif
        const uint32_t code1 = firstD ? codeF : codeD; // the other candidate's operands: selected here, where they are needed
        const uint32_t cell1 = lrm_toltab_resolve(G, firstD ? cF : cD, firstD ? sF : sD, fbase);
        const float w1 = firstD ? wF : wD, x1 = firstD ? xF : xD;
        const bool lim1 = code1 >= 2u;
        float du1, dz1;
        bool valid1;
        uint32_t bd = 0;
        uint32_t row1 = 0;
        lrm_tol_plane_tab<kInfo>(G, cell1, x1, z, band, tau, du1, dz1, valid1, bd, &row1);
        const float q = __builtin_fmaf(du1, du1, dz1 * dz1), w2 = w1 * w1;
        const bool lv = lim1 && valid1, big = q > w2 * 9.6e-7f;
        const bool collapse = lv && big;
        du1 = collapse ? 0.f : du1;
        dz1 = collapse ? 0.f : dz1;
        bd |= (lv && !big) ? LRM_TD_LIMIT : 0u;
        const float n1 = __builtin_fmaf(du1, du1, __builtin_fmaf(w1, w1, dz1 * dz1));
        // distance_circles' pick (one_leg.cu:334): both invalid here (a valid first candidate never asks for the second): the shorter one
        const float nmin = LRM_FAST_SQRT(fminf(n0, n1));
        const float thr = tau * __builtin_fmaf(2.0f, nmin, tau);
        bd |= !(fabsf(n0 - n1) > thr) ? LRM_TD_PICK : 0u;
        lu |= need ? bd : 0u;
        const bool useB = need && !(n0 < n1);
        codeW = useB ? code1 : codeW;
        if (kInfo) { // the second candidate won: its row, its kind, the other flip; its vector is an offset when it collapsed
            row0 = useB ? (row1 | cell1) : row0;
            offW = useB ? collapse : offW;
            infoB = useB;
            codeB = code1;
        }
        duW = useB ? du1 : duW;
        wW = useB ? w1 : wW;
        dzW = useB ? dz1 : dzW;
    }
    // ---- back: rotate by the winner's yaw, then to the caller's frame ----
    const float cl = (codeW == 3u) ? cm : cM, sl = (codeW == 3u) ? sm : sM;
    const float cr = lrm_u2f(lrm_f2u(cu) ^ (codeW << 31)), sr = lrm_u2f(lrm_f2u(su) ^ (codeW << 31));
    const bool limW = codeW >= 2u;
    const float c = limW ? cl : cr, s = limW ? sl : sr;
    const float vx = __builtin_fmaf(duW, c, -(wW * s)), vy = __builtin_fmaf(duW, s, wW * c), vz = dzW;
    const float* b = L.back;
    p.x = __builtin_fmaf(b[0], vx, __builtin_fmaf(b[1], vy, b[2] * vz));
    p.y = __builtin_fmaf(b[3], vx, __builtin_fmaf(b[4], vy, b[5] * vz));
    p.z = __builtin_fmaf(b[6], vx, __builtin_fmaf(b[7], vy, b[8] * vz));
    doubt |= lu;
    if (kInfo) {
        const uint32_t codeC = infoB ? codeB : code0;   // the candidate's own kind (codeW names the limit plane when the alternative won)
        const bool flipC = infoB ? firstD : !firstD;
        *info = row0 | (codeC << 16) | (flipC ? 0x40000u : 0u) | (offW ? 0x80000u : 0u) | (flag ? 0x100000u : 0u);
    }
    return flag;
}

// distance_global + reachability_global (one_leg_global.cu:74-130) of one body-frame point.
// p: in = the point, out = the distance vector.  Returns the reach / validity flag (the two coincide whenever
// no decision is in doubt, see lrm_reach_from_dist).  doubt != 0: the outputs must not be used.
template <class Plane>
LRM_HD bool lrm_dist_tol_t(const LrmTolLeg& L, const Plane& plane, LrmVec3& p, uint32_t& doubt) {
    const float* a = L.aff;
    const float x = __builtin_fmaf(a[0], p.x, __builtin_fmaf(a[1], p.y, __builtin_fmaf(a[2], p.z, a[3])));
    const float y = __builtin_fmaf(a[4], p.x, __builtin_fmaf(a[5], p.y, __builtin_fmaf(a[6], p.z, a[7])));
    const float z = __builtin_fmaf(a[8], p.x, __builtin_fmaf(a[9], p.y, __builtin_fmaf(a[10], p.z, a[11])));
    // non-finite input: the band is nan/inf and every "> band" test fails closed (doubt)
    const float band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), L.band_slope, L.band_base);
    const float tau = band * LRM_TOL_TIE;
    const float m2 = __builtin_fmaf(y, y, x * x);
    const float rs = LRM_FAST_RSQ(m2);
    const float r = m2 * rs;
    const float cu = x * rs, su = y * rs; // cos / sin of the point's yaw
    const float cM = L.yaw_cs[0], sM = L.yaw_cs[1], cm = L.yaw_cs[2], sm = L.yaw_cs[3];
    // abscissa / offset of the point in the two yaw-limit planes; their signs are the sector tests
    const float uM = __builtin_fmaf(x, cM, y * sM), wM = __builtin_fmaf(y, cM, -(x * sM));
    const float um = __builtin_fmaf(x, cm, y * sm), wm = __builtin_fmaf(y, cm, -(x * sm));
    const uint32_t pat = (lrm_f2u(wM) >> 31) | ((lrm_f2u(uM) >> 30) & 2u) | ((lrm_f2u(wm) >> 29) & 4u) |
                         ((lrm_f2u(um) >> 28) & 8u);
    constexpr uint32_t kLutD = lrm_tol_lut(false), kLutF = lrm_tol_lut(true);
    const uint32_t codeD = (kLutD >> (pat << 1)) & 3u, codeF = (kLutF >> (pat << 1)) & 3u;
    // inside [min, max]: the only kind of candidate that can be reachable and that sees the yaw-limit alternative
    const bool inD = (pat & 5u) == 1u, inF = (pat & 5u) == 4u;
    const bool same = codeD == codeF; // one of them is mega-saturated onto the other
    const float ymin = fminf(fminf(fabsf(wM), fabsf(uM)), fminf(fabsf(wm), fabsf(um)));
    uint32_t lu = !(ymin > band) ? LRM_TD_YAW : 0u;
    // Close to the coxa axis the yaw direction (x, y) / r amplifies the rounding of (x, y) by |du| / r (for the
    // reference just as much): inside LRM_TOL_RMIN the point goes to the bit-exact code.
    lu |= !(r > LRM_TOL_RMIN) ? LRM_TD_YAW : 0u;

    // one copy of the candidate evaluation, executed twice (k = 0 direct, 1 flipped)
    float Xa = 0.f, Ya = 0.f, Za = 0.f, na = 0.f, Xb = 0.f, Yb = 0.f, Zb = 0.f, nb = 0.f;
    bool fa = false, fb = false;
#pragma unroll LRM_TOL_CAND_UNROLL
    for (int k = 0; k < 2; k++) {
        if (Plane::kSkipSame && k && LRM_TOL_ALL(same)) break; // the second candidate is the first one again
        const uint32_t code = k ? codeF : codeD;
        const bool lim = code >= 2u, mn = code == 3u, neg = code == 1u;
        // rotate the point by -sat: (cos, sin)(sat) = +-(x, y) / r for its own / the opposite meridian plane,
        // the limit's constants for a limit plane
        const float c = lim ? (mn ? cm : cM) : lrm_u2f(lrm_f2u(cu) ^ (neg ? 0x80000000u : 0u));
        const float s = lim ? (mn ? sm : sM) : lrm_u2f(lrm_f2u(su) ^ (neg ? 0x80000000u : 0u));
        const float u = __builtin_fmaf(x, c, y * s);
        const float w = lim ? __builtin_fmaf(y, c, -(x * s)) : 0.f;
        float du = 0.f, dz = 0.f;
        bool valid = false;
        uint32_t d = 0;
        plane(u, z, band, tau, du, dz, valid, d);
        // A candidate clamped to a yaw limit whose plane point is valid: the yaw-limit alternative of
        // one_leg.cu:258-274 runs with th = -(limit - sat) = 0, d_limit = |w| < |(du, w, dz)|: the result is
        // the offset from the limit plane alone
        // -- unless |(du, dz)| is so small that sqrt(du^2 + w^2 + dz^2) ROUNDS to |w|: the reference's
        // `d_clamped > d_limit` is then false and it keeps the in-plane part (a vector 3e-4 |w| long at most,
        // still 30x the tolerance).  q / w^2 below 2^-25 cannot move the float sum, above 2^-20 always does;
        // in between the outcome depends on the rounding of the sum and of the square root: doubt.
        if (lim && valid) {
            const float q = __builtin_fmaf(du, du, dz * dz), w2 = w * w;
            if (q > w2 * 9.6e-7f) du = dz = 0.f;
            else if (!(q < w2 * 2.9e-8f)) d |= LRM_TD_LIMIT;
        }
        if (!(k && same)) lu |= d;
        const float X = __builtin_fmaf(du, c, -(w * s)), Y = __builtin_fmaf(du, s, w * c);
        const float nn = __builtin_fmaf(du, du, __builtin_fmaf(w, w, dz * dz));
        const bool fl = valid && (k ? inF : (inD || (same && inF)));
#if defined(LRM_TOL_TRACE)
        printf(" cand %d code %u c %.6f s %.6f u %.5f w %.5f -> X %.5f Y %.5f Z %.5f n %.6f flag %d\n", k, code, c, s, u, w, X, Y, dz, sqrtf(nn), (int)fl);
#endif
        if (k) { Xb = X; Yb = Y; Zb = dz; nb = nn; fb = fl; }
        else { Xa = X; Ya = Y; Za = dz; na = nn; fa = fl; }
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    const float na0 = na, nb0 = nb;
#endif
    // the yaw-limit alternative (one_leg.cu:258-274) of the candidate inside the yaw range, when it is valid:
    // the nearer limit plane wins over the in-plane boundary when it is closer
    {
        const bool onB = inF && !same;
        const float nu = onB ? nb : na;
        if (onB ? fb : fa) {
            const float aM = fabsf(wM), am = fabsf(wm);
            const float dl = fminf(aM, am), dl2 = dl * dl;
            const float thr = tau * __builtin_fmaf(2.0f, dl, tau);
            // in doubt: the two distances tie, or the alternative is (or may be) taken and the two limit planes tie
            // (the reference picks by `angle > coxa_mid`; on the symmetry plane of a symmetric leg -- the bench
            // grid, y = 0 -- both are the same distance, which only matters when the alternative wins)
            if (!(fabsf(nu - dl2) > thr) || (!(nu < dl2 - thr) && !(fabsf(aM - am) > tau))) lu |= LRM_TD_LIMIT;
            if (nu > dl2) {
                const bool useM = aM < am;
                const float wl = useM ? wM : wm, sl = useM ? sM : sm, cl = useM ? cM : cm;
                const float X = -(wl * sl), Y = wl * cl;
                if (onB) { Xb = X; Yb = Y; Zb = 0.f; nb = dl2; }
                else { Xa = X; Ya = Y; Za = 0.f; na = dl2; }
            }
        }
    }
    // distance_circles' pick (one_leg.cu:334): equal validities -> the shorter vector, else the valid one
    bool useD = true;
    if (!same) {
        const float nmin = LRM_FAST_SQRT(fminf(na, nb));
        const float thr = tau * __builtin_fmaf(2.0f, nmin, tau);
        const bool eq = fa == fb;
        if (eq && !(fabsf(na - nb) > thr)) lu |= LRM_TD_PICK;
        useD = eq ? (na < nb) : fa;
#if !defined(__HIP_DEVICE_COMPILE__)
        // statistic (host only): could one candidate have been skipped?  The squared norm of a candidate is at
        // least lb = w^2 + (|(u - coxa, z)| - r_outer)+^2.  Evaluate the candidate with the smaller bound first;
        // the other one cannot win when the first is valid or already shorter than the other's bound.
        {
            float lb[2];
            for (int k = 0; k < 2; k++) {
                const uint32_t code = k ? codeF : codeD;
                const bool lim = code >= 2u, mn = code == 3u, neg = code == 1u;
                const float u = lim ? (mn ? um : uM) : (neg ? -r : r);
                const float w = lim ? (mn ? wm : wM) : 0.f;
                const float ux = u - L.coxa_length;
                const float out = fmaxf(LRM_FAST_SQRT(__builtin_fmaf(ux, ux, z * z)) - L.r_outer, 0.f);
                lb[k] = __builtin_fmaf(w, w, out * out);
            }
            const bool firstD = inD || (!inF && lb[0] <= lb[1]); // a candidate inside the yaw range goes first
            const float n1 = firstD ? na0 : nb0, lb2 = firstD ? lb[1] : lb[0];
            const bool f1 = firstD ? fa : fb;
            if (!f1 && !(n1 < lb2 - thr)) lu |= LRM_TD_SECOND;
        }
#endif
    }
    const float vx = useD ? Xa : Xb, vy = useD ? Ya : Yb, vz = useD ? Za : Zb;
    const float* b = L.back;
    p.x = __builtin_fmaf(b[0], vx, __builtin_fmaf(b[1], vy, b[2] * vz));
    p.y = __builtin_fmaf(b[3], vx, __builtin_fmaf(b[4], vy, b[5] * vz));
    p.z = __builtin_fmaf(b[6], vx, __builtin_fmaf(b[7], vy, b[8] * vz));
    doubt |= lu;
    return fa || fb;
}

// -------------------------------------------------------------------------------------------------------
// Staged form of the same evaluation: the more promising yaw candidate first, the second one only when it can
// still win.  distance_circles (one_leg.cu:321-341) keeps the valid candidate, or the shorter of two invalid ones;
// only a candidate inside the yaw range can be valid, so that one goes first (else the one whose plane is
// nearer), and the other is skipped when the first is valid or already shorter than a lower bound of the
// other's norm: |d|^2 >= w^2 + (|(u - coxa, z)| - r_outer)+^2 (every clamp target lies within r_outer of the femur
// joint).  About 75 % of the second evaluations of a random cloud go away; what remains is mostly inherent: behind the
// robot the mirrored yaw is in range, but the limit plane in front is as near as the opposite meridian plane (338 against
// 336 mm is typical), so whichever goes first the other's bound cannot exclude it.  (Ordering both by their bounds
// instead: 20 % instead of 26 % second evaluations, but 16 more instructions for every point -- 113 us against 111.)
// The kernel runs the remaining ones
// compacted over the workgroup (lrm_tol_kernels.hip): only the PLANE evaluation travels to another lane.
// -------------------------------------------------------------------------------------------------------
struct LrmTolPoint {
    float z, band, tau;
    float wM, wm;            // offsets from the two yaw-limit planes (the yaw-limit alternative)
    float c0, s0, u0, w0;    // first candidate: rotation (cos, sin) and plane point (u, z) + offset w
    float c1, s1, u1, w1;    // second candidate
    uint32_t lu;             // doubt bits so far
    bool lim0, lim1;         // the candidate is clamped to a yaw limit
    bool in0;                // the first candidate lies inside the yaw range (the only kind that can be valid)
    bool two;                // a second, different candidate exists
};
struct LrmTolCand {
    float X, Y, Z, n; // vector in the coxa frame, squared norm
    bool flag;        // valid and inside the yaw range
};

LRM_HD LrmTolPoint lrm_tol_prologue(const LrmTolLeg& L, LrmVec3 p) {
    LrmTolPoint S;
    const float* a = L.aff;
    const float x = __builtin_fmaf(a[0], p.x, __builtin_fmaf(a[1], p.y, __builtin_fmaf(a[2], p.z, a[3])));
    const float y = __builtin_fmaf(a[4], p.x, __builtin_fmaf(a[5], p.y, __builtin_fmaf(a[6], p.z, a[7])));
    S.z = __builtin_fmaf(a[8], p.x, __builtin_fmaf(a[9], p.y, __builtin_fmaf(a[10], p.z, a[11])));
    S.band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), L.band_slope, L.band_base);
    S.tau = S.band * LRM_TOL_TIE;
    const float m2 = __builtin_fmaf(y, y, x * x);
    const float rs = LRM_FAST_RSQ(m2);
    const float r = m2 * rs;
    const float cu = x * rs, su = y * rs;
    const float cM = L.yaw_cs[0], sM = L.yaw_cs[1], cm = L.yaw_cs[2], sm = L.yaw_cs[3];
    const float uM = __builtin_fmaf(x, cM, y * sM), wM = __builtin_fmaf(y, cM, -(x * sM));
    const float um = __builtin_fmaf(x, cm, y * sm), wm = __builtin_fmaf(y, cm, -(x * sm));
    S.wM = wM;
    S.wm = wm;
    const uint32_t pat = (lrm_f2u(wM) >> 31) | ((lrm_f2u(uM) >> 30) & 2u) | ((lrm_f2u(wm) >> 29) & 4u) |
                         ((lrm_f2u(um) >> 28) & 8u);
    constexpr uint32_t kLutD = lrm_tol_lut(false), kLutF = lrm_tol_lut(true);
    const uint32_t codeD = (kLutD >> (pat << 1)) & 3u, codeF = (kLutF >> (pat << 1)) & 3u;
    const bool inD = (pat & 5u) == 1u, inF = (pat & 5u) == 4u;
#if LRM_TOL_DIET
    const float ymin = lrm_min3_aa(wm, um, lrm_min3_aa(wM, uM, 3.0e38f));
#else
    const float ymin = fminf(fminf(fabsf(wM), fabsf(uM)), fminf(fabsf(wm), fabsf(um)));
#endif
    S.lu = (!(ymin > S.band) || !(r > LRM_TOL_RMIN)) ? LRM_TD_YAW : 0u;
    S.two = codeD != codeF;
    // offsets of the two candidates' planes: 0 for a meridian plane, w of the limit for a limit plane
    const float wD = codeD >= 2u ? (codeD == 3u ? wm : wM) : 0.f, wF = codeF >= 2u ? (codeF == 3u ? wm : wM) : 0.f;
    const bool firstD = inD || (!inF && fabsf(wD) <= fabsf(wF));
    const uint32_t code0 = firstD ? codeD : codeF, code1 = firstD ? codeF : codeD;
    S.in0 = inD || inF;
    auto rot = [&](uint32_t code, float& c, float& s, float& u, float& w, bool& lim) {
        lim = code >= 2u;
        const bool mn = code == 3u, neg = code == 1u;
        c = lim ? (mn ? cm : cM) : lrm_u2f(lrm_f2u(cu) ^ (neg ? 0x80000000u : 0u));
        s = lim ? (mn ? sm : sM) : lrm_u2f(lrm_f2u(su) ^ (neg ? 0x80000000u : 0u));
        u = __builtin_fmaf(x, c, y * s);
        w = lim ? __builtin_fmaf(y, c, -(x * s)) : 0.f;
    };
    rot(code0, S.c0, S.s0, S.u0, S.w0, S.lim0);
    rot(code1, S.c1, S.s1, S.u1, S.w1, S.lim1);
    return S;
}

// what follows a plane evaluation (du, dz, valid) of candidate `second ? 1 : 0`
LRM_HD LrmTolCand lrm_tol_candidate(const LrmTolPoint& S, bool second, float du, float dz, bool valid, uint32_t& d) {
    const bool lim = second ? S.lim1 : S.lim0;
    const float c = second ? S.c1 : S.c0, s = second ? S.s1 : S.s0, w = second ? S.w1 : S.w0;
    // a candidate clamped to a yaw limit whose plane point is valid collapses to the offset from that plane, unless
    // sqrt(du^2 + w^2 + dz^2) rounds to |w| (see lrm_dist_tol_t)
    if (lim && valid) {
        const float q = __builtin_fmaf(du, du, dz * dz), w2 = w * w;
        if (q > w2 * 9.6e-7f) du = dz = 0.f;
        else if (!(q < w2 * 2.9e-8f)) d |= LRM_TD_LIMIT;
    }
    LrmTolCand C;
    C.X = __builtin_fmaf(du, c, -(w * s));
    C.Y = __builtin_fmaf(du, s, w * c);
    C.Z = dz;
    C.n = __builtin_fmaf(du, du, __builtin_fmaf(w, w, dz * dz));
    C.flag = valid && !second && S.in0;
    return C;
}

// can the second candidate still win against the evaluated first one?
LRM_HD bool lrm_tol_need_second(const LrmTolLeg& L, const LrmTolPoint& S, const LrmTolCand& A) {
    const float ux = S.u1 - L.coxa_length;
    const float out = fmaxf(LRM_FAST_SQRT(__builtin_fmaf(ux, ux, S.z * S.z)) - L.r_outer, 0.f);
    const float lb = __builtin_fmaf(S.w1, S.w1, out * out);
    const float thr = S.tau * __builtin_fmaf(2.0f, LRM_FAST_SQRT(A.n), S.tau); // the tie band of the pick, squared domain
    return S.two && !A.flag && !(A.n < lb - thr);
}

// yaw-limit alternative of the first candidate, the pick, the way back to the caller's frame
LRM_HD bool lrm_tol_finish(const LrmTolLeg& L, const LrmTolPoint& S, LrmTolCand A, bool haveB, const LrmTolCand B,
                           LrmVec3& out, uint32_t& doubt) {
    uint32_t lu = 0;
    if (A.flag) { // one_leg.cu:258-274: the nearer limit plane wins over the in-plane boundary when it is closer
        const float cM = L.yaw_cs[0], sM = L.yaw_cs[1], cm = L.yaw_cs[2], sm = L.yaw_cs[3];
        const float aM = fabsf(S.wM), am = fabsf(S.wm);
        const float dl = fminf(aM, am), dl2 = dl * dl;
        const float thr = S.tau * __builtin_fmaf(2.0f, dl, S.tau);
        if (!(fabsf(A.n - dl2) > thr) || (!(A.n < dl2 - thr) && !(fabsf(aM - am) > S.tau))) lu |= LRM_TD_LIMIT;
        if (A.n > dl2) {
            const bool useM = aM < am;
            const float wl = useM ? S.wM : S.wm, sl = useM ? sM : sm, cl = useM ? cM : cm;
            A.X = -(wl * sl);
            A.Y = wl * cl;
            A.Z = 0.f;
            A.n = dl2;
        }
    }
    bool useA = true;
    if (haveB) { // both invalid (a valid first candidate never asks for the second): the shorter one
        const float nmin = LRM_FAST_SQRT(fminf(A.n, B.n));
        const float thr = S.tau * __builtin_fmaf(2.0f, nmin, S.tau);
        if (!(fabsf(A.n - B.n) > thr)) lu |= LRM_TD_PICK;
        useA = A.n < B.n;
    }
    const float vx = useA ? A.X : B.X, vy = useA ? A.Y : B.Y, vz = useA ? A.Z : B.Z;
    const float* b = L.back;
    out.x = __builtin_fmaf(b[0], vx, __builtin_fmaf(b[1], vy, b[2] * vz));
    out.y = __builtin_fmaf(b[3], vx, __builtin_fmaf(b[4], vy, b[5] * vz));
    out.z = __builtin_fmaf(b[6], vx, __builtin_fmaf(b[7], vy, b[8] * vz));
    doubt |= lu;
    return A.flag;
}

// the whole evaluation of one point, staged (host builds, the GPU's middle kernel of the plane-table variant)
LRM_HD bool lrm_dist_tol(const LrmTolLeg& L, const LrmTolTables T, LrmVec3& p, uint32_t& doubt) {
    const LrmTolPoint S = lrm_tol_prologue(L, p);
    uint32_t lu = S.lu;
    float du, dz;
    bool valid;
    lrm_tol_plane(L, T, S.u0, S.z, S.band, S.tau, du, dz, valid, lu);
    const LrmTolCand A = lrm_tol_candidate(S, false, du, dz, valid, lu);
    LrmTolCand B = A;
    const bool need = lrm_tol_need_second(L, S, A);
    if (need) {
        lrm_tol_plane(L, T, S.u1, S.z, S.band, S.tau, du, dz, valid, lu);
        B = lrm_tol_candidate(S, true, du, dz, valid, lu);
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    if (need) lu |= LRM_TD_SECOND; // statistic
#endif
    doubt |= lu;
    return lrm_tol_finish(L, S, A, need, B, p, doubt);
}

// lrm_point_fast.h -- filtered evaluation (LRM_MODE_FAST) of the reach / distance path.
//
// Idea (floating-point filters, as in exact geometric predicates): every value that reaches
// an OUTPUT is computed with the strict, reference-order arithmetic of lrm_point.h; every
// DECISION on the way (which region, is the point inside a circle, is a clamp point valid,
// which boundary is nearest, which yaw candidate wins) is first taken with cheap arithmetic
// (FMA, squared distances instead of sqrt, cross products instead of atan2f, hardware
// rsqrt/sqrt) together with a conservative bound on how far that arithmetic can be from the
// strict one.  A decision inside its band is re-taken by the strict code at the smallest
// enclosing granularity, so results are bit-identical to the strict mode by construction; the
// bands only decide how often the slow path runs.
//
// Shape of the code: every test is expressed as a signed distance (mm) to its decision
// boundary.  The decision is a sign, accumulated with v_max; the doubt is the smallest magnitude,
// accumulated with v_min and compared ONCE with the band.  No compare/select chains (a
// v_cmp + v_cndmask pair costs about twice a plain VALU op on gfx950).
//
// What the strict path spends and this one does not:
//   reachability   2 atan2f + 1 sincosf + 4 sqrt            -> 0 (decisions only), ~70 VALU ops
//   distance, per yaw candidate
//     16 sqrt-bearing circle validations of clamp points    -> 12 "arc" dot products
//     4 + n_corners exact clamps (sqrt + div each)          -> 1 exact clamp (the winner's)
//     1 exact atan2f for the region                         -> 3 cross products
//   kept exact in the distance: 1 atan2f, 1 sincosf per candidate, the winner's clamp, the
//   yaw-limit alternative when it is taken or in doubt, and the final comparison of the two
//   candidates' strict norms (for |yaw| < 30 deg the "flipped" candidate is the same
//   configuration up to rounding and the reference's pick is decided by rounding noise).
//
// Error model (u = 2^-24).  Distance filter: px, pz and the rotated coordinates are the strict
// values themselves, only the decision arithmetic differs:
//   m  = |p - c|^2 by FMA : relative error <= 2u;  v_sqrt_f32 / v_rsq_f32 : <= 1 ulp;
//   strict d = r - sqrt(m) is exact by Sterbenz near the boundary, its sqrt carries 0.5 ulp
//   => |d_fast - d_strict| <= 4u * mag <= 2.4e-7 * S,  S = |px| + |pz| + fast_scale.
//   Band: LRM_BAND_DIST * S with LRM_BAND_DIST = 2e-6 (>= 8x).  Cross products t = cosC*pz - sinC*px:
//   error <= 4u (|px| + |pz|), strict atan2f within 1 ulp of the true angle (2e-7 |p| in t): >= 4x.
//   Arc form of the clamp validity: <= 12u * fast_scale against LRM_BAND_DIST * 2 * fast_scale (>= 5x).
// Reach filter: the coxa-frame point itself is approximate (one FMA affine map instead of the
// strict qtRotate / z-rotation / translation / pitch chain: <= 22u (|p|_1 + body) apart) and the
// plane abscissa is sgn(x) sqrt(x^2 + y^2) instead of x cos(a) - y sin(a) (<= 8u r apart):
//   band = LRM_BAND * (fast_scale + (sqrt(3) + 1.5) (|p|_1 + body)), LRM_BAND = 4e-6 (>= 4.5x, linear in |p|_1).
// Empirical margin (tests/test_capi_cpu.py keeps the bands honest on every fixture, including the
// boundary-hugging one): with both bands at 1e-7 there are still zero mismatches on 6e6 points x 6
// leg/orientation cases; the first mismatches appear at 2e-8 (100-200x smaller than the bands in use).
// The distance band is the tighter one because its doubts cost more: a wave re-runs the strict
// plane evaluation (~800 instructions) when ANY of its 64 lanes is in doubt.
#pragma once
#include "lrm_point.h"

#ifndef LRM_BAND
#define LRM_BAND 4.0e-6f
#endif
// Unroll factor of the four-circle loop of lrm_plane_dist_fast.  4 keeps all four circle records
// (80 table values) in flight; 2 halves that and lets the distance kernels run 7 waves/SIMD, which
// is worth more (see LRM_DIST_MIN_WAVES in lrm_kernels.hip).
#ifndef LRM_CIRCLE_UNROLL
#define LRM_CIRCLE_UNROLL 2
#endif
#ifndef LRM_BAND_DIST
#define LRM_BAND_DIST 2.0e-6f
#endif

// Device kernels that read the leg from the kernarg segment define LRM_FRESH as lrm_fresh: the
// constants are then re-fetched (s_load) per phase instead of staying live in SGPRs throughout.
#ifndef LRM_FRESH
#define LRM_FRESH(L) (L)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define LRM_FAST_SQRT(v) __builtin_amdgcn_sqrtf(v)
#define LRM_FAST_RSQ(v) __builtin_amdgcn_rsqf(v)
#else
#define LRM_FAST_SQRT(v) sqrtf(v)
#define LRM_FAST_RSQ(v) (1.0f / sqrtf(v))
#endif

LRM_HD uint32_t lrm_umed3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
#endif
}

// ---------------------------------------------------------------------------------------
// reachability: decisions only.  Lean form: one affine map into the coxa frame, every test
// turned into a signed distance-to-the-decision-boundary in mm ("value"); the decision is the
// sign of the largest value, the doubt is the smallest |value| against one band.  No
// compare/select chains: v_max / v_min accumulators.
// ---------------------------------------------------------------------------------------

// "atan2f(y, x) > C" for a constant C in (-pi, pi) from the cross product t = cosC*y - sinC*x
// (> 0: (x, y) is counter-clockwise of direction C by less than pi), with the wrap of atan2f at
// +-pi handled:   C >= 0: t > 0 and y >= 0;   C < 0: y >= 0 or t > 0.
// Doubt: |t| small (the point is near the ray), or x < 0 with y = +-0 (atan2f jumps from +pi to -pi).
//
// find_region (circles.cu.h:48-78) from the three cross products and y: the boolean form above
// depends only on four signs (and on the signs of the three constants), so the host tabulates it (LrmCompiledLeg::region_lut, 2 bits per
// sign pattern) and the per-point code gathers sign bits with shifts/and/or (full-rate VALU
// operations on gfx950; a v_cmp + v_cndmask pair costs 4x as much).  t = +-0 reads as "positive"
// or "negative" by its sign bit; every caller treats |t| <= band (> 0) as a doubt, so the
// difference with `t > 0` never reaches a result.  Returns upper * 2 + fully_extended.
LRM_HD uint32_t lrm_region_from_signs(uint32_t lut, float t_mid, float t_s0, float t_s1, float y) {
    const uint32_t pat = (lrm_f2u(t_mid) >> 31) | ((lrm_f2u(t_s0) >> 30) & 2u) | ((lrm_f2u(t_s1) >> 29) & 4u) |
                         ((lrm_f2u(y) >> 28) & 8u);
    return (lut >> (pat << 1)) & 3u;
}

// (x, y, z): the point in the coxa frame (after place_over_coxa), approximate; `band` (mm) covers
// the distance between this evaluation and the strict one for every test below.
LRM_HD bool lrm_reach_coxa_lean(const LrmCompiledLeg& L, const LrmCompiledLeg::LeanCircle* lean, float x, float y,
                                float z, float band, uint32_t& unc) {
    const uint32_t sx = lrm_f2u(x) & 0x80000000u;
    const float ax = fabsf(x);
    const float ay = lrm_u2f(lrm_f2u(y) ^ sx); // the reference mirrors the point into x >= 0 (one_leg.cu:291-296)
    const float r_xy = LRM_FAST_SQRT(__builtin_fmaf(ay, ay, ax * ax));
    const float px = lrm_u2f(lrm_f2u(r_xy) | sx) - L.coxa_length;
    // yaw limits (both inside (-pi/2, pi/2), ax >= 0): above <=> t_max > 0, below <=> t_min < 0
    const float t_max = __builtin_fmaf(L.dir_cos[3], ay, -(L.dir_sin[3] * ax));
    const float t_min = __builtin_fmaf(L.dir_cos[4], ay, -(L.dir_sin[4] * ax));
    const float coxa_v = fmaxf(t_max, -t_min);
    const float coxa_m = fminf(fabsf(t_max), fabsf(t_min));
    // region (circles.cu.h:48-78)
    const float t_mid = __builtin_fmaf(L.dir_cos[0], z, -(L.dir_sin[0] * px));
    const float t_s0 = __builtin_fmaf(L.dir_cos[1], z, -(L.dir_sin[1] * px));
    const float t_s1 = __builtin_fmaf(L.dir_cos[2], z, -(L.dir_sin[2] * px));
    const uint32_t reg = lrm_region_from_signs(L.region_lut, t_mid, t_s0, t_s1, z);
    // doubt of the region: distance to any of the three rays, or to the atan2f wrap ray (x < 0, y = +-0)
    float macc = fminf(fminf(fabsf(t_mid), fabsf(t_s0)), fminf(fabsf(t_s1), fmaxf(px, fabsf(z))));
    const LrmCompiledLeg::LeanCircle* c = lean + reg * LRM_N_CIRCLES;
    float vacc = -3.0e38f;
#pragma unroll
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        const LrmCompiledLeg::LeanCircle ci = c[i];
        const float dx = px - ci.x, dy = z - ci.y;
        const float v = __builtin_fmaf(__builtin_fmaf(dy, dy, dx * dx), ci.gs, ci.c);
        vacc = fmaxf(vacc, v);
        macc = fminf(macc, fabsf(v));
    }
    const bool coxa_ok = coxa_v < 0.f;
    // outside the yaw range the answer is "no" whatever the circles say
    unc |= ((!(coxa_m > band)) | (coxa_ok & !(macc > band))) ? 1u : 0u;
    return coxa_ok & (vacc < 0.f);
}

LRM_HD bool lrm_reach_global_fast(const LrmCompiledLeg& L, const LrmCompiledLeg::LeanCircle* lean, LrmVec3 p,
                                  uint32_t& unc) {
    const float* a = L.aff_global;
    const float x = __builtin_fmaf(a[0], p.x, __builtin_fmaf(a[1], p.y, __builtin_fmaf(a[2], p.z, a[3])));
    const float y = __builtin_fmaf(a[4], p.x, __builtin_fmaf(a[5], p.y, __builtin_fmaf(a[6], p.z, a[7])));
    const float z = __builtin_fmaf(a[8], p.x, __builtin_fmaf(a[9], p.y, __builtin_fmaf(a[10], p.z, a[11])));
    // non-finite input: the band is nan/inf and every "> band" test fails closed
    const float band = __builtin_fmaf(fabsf(p.x) + fabsf(p.y) + fabsf(p.z), L.band_slope, L.band_base);
    return lrm_reach_coxa_lean(L, lean, x, y, z, band, unc);
}

LRM_HD bool lrm_reachable_rotate_leg_fast(const LrmCompiledLeg& L, const LrmCompiledLeg::LeanCircle* lean,
                                          LrmVec3 t, LrmVec3 body, uint32_t& unc) {
    t.x -= body.x; // same float subtractions as the strict code
    t.y -= body.y;
    t.z -= body.z;
    const float* a = L.aff_pair;
    const float mag = fabsf(t.x) + fabsf(t.y) + fabsf(t.z);
    const float gx = __builtin_fmaf(L.grav_row[0], t.x, __builtin_fmaf(L.grav_row[1], t.y, L.grav_row[2] * t.z));
    const float band = __builtin_fmaf(mag, L.band_slope, L.band_base);
    // gravity side (several_leg.cu:58-61): strict gx < 0 -> not reachable
    unc |= !(fabsf(gx) > band) ? 2u : 0u;
    if (gx < 0.f) return false;
    const float x = __builtin_fmaf(a[0], t.x, __builtin_fmaf(a[1], t.y, __builtin_fmaf(a[2], t.z, a[3])));
    const float y = __builtin_fmaf(a[4], t.x, __builtin_fmaf(a[5], t.y, __builtin_fmaf(a[6], t.z, a[7])));
    const float z = __builtin_fmaf(a[8], t.x, __builtin_fmaf(a[9], t.y, __builtin_fmaf(a[10], t.z, a[11])));
    return lrm_reach_coxa_lean(L, lean, x, y, z, band, unc);
}

// ---------------------------------------------------------------------------------------
// distance_global: filtered decisions + strict arithmetic for the winner
// ---------------------------------------------------------------------------------------

// Tables of the lean distance filter as the per-point code sees them (LDS on the device,
// the LrmCompiledLeg members on the host).
struct LrmDistTables {
    const LrmCircle* lists;                        // [16] strict circle lists
    const LrmCompiledLeg::DistCircle* dist;        // [16]
    const LrmCircle* corners;                      // [LRM_N_CORNERS]
    uint32_t force_strict;                         // self-test (LRM_TOL_SELFTEST): every plane evaluation takes the strict path
};

#if defined(__HIP_DEVICE_COMPILE__)
// lrm_plane_dist (one_leg.cu:91-145, :167-208) of ONE plane point, evaluated by the whole wave: (x, y) is wave-uniform, every
// lane returns the same result, bit for bit what lrm_plane_dist gives.  The strict evaluation is one dependent chain of
// ~30 IEEE square roots and divisions (4 clamps x 4 validations + the corner points); here lane l < 16 clamps onto circle
// l / 4 and validates the clamp point against circle l % 4, lanes 16 .. 16 + n_corners - 1 take one corner point each, and
// the reference's sequential "keep the earlier candidate unless a later one is strictly closer" is the minimum of |d| with
// ties to the lowest candidate number -- candidates sit in the lanes in the reference's order.
// ALL 64 lanes of the wave must call it together.
__device__ __forceinline__ bool lrm_plane_dist_coop(const LrmCompiledLeg& L, const LrmCircle* lists, float& x, float& y, int lane) {
    x -= L.coxa_length;
    const LrmCircle* list = lists + lrm_region(L, x, y) * LRM_N_CIRCLES;
    const bool is_circle = lane < 4 * LRM_N_CIRCLES;
    const int ci = lane - 4 * LRM_N_CIRCLES, nco = L.n_corners;
    const bool is_corner = ci >= 0 && ci < nco;
    const int ck = is_corner ? ci : 0;
    const LrmCircle cl = list[is_circle ? (lane >> 2) : 0];
    const float ccx = is_circle ? cl.x : L.corner_x[ck], ccy = is_circle ? cl.y : L.corner_y[ck], ccr = is_circle ? cl.r : 0.f;
    const bool att = is_circle ? (cl.attract != 0.f) : true;
    float cx = x, cy = y, d;
    bool valid;
    lrm_clamp_on(ccx, ccy, ccr, att, cx, cy, d, valid);
    const bool okj = lrm_circle_valid(list[lane & 3], cx, cy);
    const unsigned long long okm = __ballot(okj), vm = __ballot(valid);
    const bool clamp_ok = ((okm >> (lane & ~3)) & 15ull) == 15ull;
    const bool overall = (vm & 0xffffull) == 0xffffull; // lrm_clamp_on's `valid` of the four circles (each held by four lanes)
    const float ad = fabsf(d);
    const bool cand = (is_circle ? clamp_ok : (is_corner && !overall)) && (999999999999999.9f > ad); // `fabsf(best_d) > fabsf(d)` against the initial best_d
    float key = cand ? ad : __builtin_inff();
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) key = fminf(key, __shfl_xor(key, off)); // candidates live in lanes 0 .. 25
    const unsigned long long eq = __ballot(cand && ad == key) & 0xffffffffull;
    float bx = 0.f, by = 0.f;
    if (eq) {
        const int w = __builtin_ctzll(eq);
        bx = lrm_u2f((uint32_t)__builtin_amdgcn_readlane((int)lrm_f2u(cx), w));
        by = lrm_u2f((uint32_t)__builtin_amdgcn_readlane((int)lrm_f2u(cy), w));
    }
    x -= bx;
    y -= by;
    return overall;
}
#endif

// eval_plane_circles<DIST> + multi_circle_clamp with filtered decisions.  (x, y) in/out as in
// lrm_plane_dist.  Every test is a signed distance to its decision boundary (mm): decisions are
// signs, doubts are the smallest magnitudes against one band (no compare/select chains); the
// candidates are ranked as integer keys (distance bits with the candidate number in the 4 low
// bits) through a 3-deep sorting network.  A decision inside its band sends this ONE call to the
// strict lrm_plane_dist; a near-tie between the two nearest boundaries is resolved by evaluating
// just those two with the strict clamp.  `unc` is only a statistic here.
// kCoop (device, the fix-up kernel of LRM_MODE_TOL): ALL 64 lanes of the wave call this together; a lane whose filter is in
// doubt does not run the strict evaluation alone (4.7 us of one lane's time, the tail of that launch) -- the wave runs it
// for the lane (lrm_plane_dist_coop), one request after the other.
template <bool kCoop = false>
LRM_HD bool lrm_plane_dist_fast(const LrmCompiledLeg& L, const LrmDistTables T, float& x, float& y, uint32_t& unc) {
    const float x_in = x, y_in = y;
    x -= L.coxa_length;
    const float band = LRM_BAND_DIST * (fabsf(x) + fabsf(y) + L.fast_scale);
    // region (circles.cu.h:48-78)
    const float t_mid = __builtin_fmaf(L.dir_cos[0], y, -(L.dir_sin[0] * x));
    const float t_s0 = __builtin_fmaf(L.dir_cos[1], y, -(L.dir_sin[1] * x));
    const float t_s1 = __builtin_fmaf(L.dir_cos[2], y, -(L.dir_sin[2] * x));
    const int reg = (int)lrm_region_from_signs(L.region_lut, t_mid, t_s0, t_s1, y) * LRM_N_CIRCLES;
    float macc = fminf(fminf(fabsf(t_mid), fabsf(t_s0)), fminf(fabsf(t_s1), fmaxf(x, fabsf(y))));
    float maccq = 3.0e38f, magmin = 3.0e38f, vacc = -3.0e38f;
    const LrmCompiledLeg::DistCircle* dt = T.dist + reg;
#if LRM_CIRCLE_UNROLL == 4
    uint32_t key[LRM_N_CIRCLES];
#pragma unroll
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        const LrmCompiledLeg::DistCircle d = dt[i];
        const float vx = x - d.x, vy = y - d.y;
        const float m = __builtin_fmaf(vy, vy, vx * vx);
        const float pv = __builtin_fmaf(m, d.gs, d.c); // validity of the point itself
        vacc = fmaxf(vacc, pv);
        macc = fminf(macc, fabsf(pv));
        const float rs = LRM_FAST_RSQ(m);
        const float mag = m * rs;
        magmin = fminf(magmin, mag);
        const float ad = fabsf(d.r - mag);
        // validity of the clamp point against the other three circles, in arc form
        const float q0 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[0].ex, vy * d.arc[0].ey), rs, d.arc[0].Q);
        const float q1 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[1].ex, vy * d.arc[1].ey), rs, d.arc[1].Q);
        const float q2 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[2].ex, vy * d.arc[2].ey), rs, d.arc[2].Q);
        const float okv = fmaxf(fmaxf(q0, q1), q2);
        maccq = fminf(maccq, fminf(fminf(fabsf(q0), fabsf(q1)), fabsf(q2)));
        // an invalid clamp ranks as +inf (exponent all ones, mantissa = candidate number only)
        key[i] = ((okv < 0.f) ? (lrm_f2u(ad) & ~15u) : 0x7f800000u) | (uint32_t)i;
    }
    const bool overall = vacc < 0.f;
    // sorted triple (a <= b <= c) of candidate keys
    uint32_t a = key[0] < key[1] ? key[0] : key[1];
    uint32_t b = key[0] < key[1] ? key[1] : key[0];
    uint32_t c = 0x7f80000fu;
    auto insert = [&](uint32_t k) { // a <= b <= c stay the three smallest: four operations
        const uint32_t bk = b > k ? b : k;       // max(a, b, k) since a <= b
        c = c < bk ? c : bk;
        b = lrm_umed3(a, b, k);
        a = a < k ? a : k;
    };
    insert(key[2]);
    insert(key[3]);
#else
    // rolled variant (fewer table values live at once): the keys are inserted as they are produced
    uint32_t a = 0x7f80000fu, b = 0x7f80000fu, c = 0x7f80000fu;
    auto insert = [&](uint32_t k) { // a <= b <= c stay the three smallest: four operations
        const uint32_t bk = b > k ? b : k;       // max(a, b, k) since a <= b
        c = c < bk ? c : bk;
        b = lrm_umed3(a, b, k);
        a = a < k ? a : k;
    };
#pragma unroll LRM_CIRCLE_UNROLL
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        const LrmCompiledLeg::DistCircle d = dt[i];
        const float vx = x - d.x, vy = y - d.y;
        const float m = __builtin_fmaf(vy, vy, vx * vx);
        const float pv = __builtin_fmaf(m, d.gs, d.c);
        vacc = fmaxf(vacc, pv);
        macc = fminf(macc, fabsf(pv));
        const float rs = LRM_FAST_RSQ(m);
        const float mag = m * rs;
        magmin = fminf(magmin, mag);
        const float ad = fabsf(d.r - mag);
        const float q0 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[0].ex, vy * d.arc[0].ey), rs, d.arc[0].Q);
        const float q1 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[1].ex, vy * d.arc[1].ey), rs, d.arc[1].Q);
        const float q2 = __builtin_fmaf(__builtin_fmaf(vx, d.arc[2].ex, vy * d.arc[2].ey), rs, d.arc[2].Q);
        const float okv = fmaxf(fmaxf(q0, q1), q2);
        maccq = fminf(maccq, fminf(fminf(fabsf(q0), fabsf(q1)), fabsf(q2)));
        insert(((okv < 0.f) ? (lrm_f2u(ad) & ~15u) : 0x7f800000u) | (uint32_t)i);
    }
    const bool overall = vacc < 0.f;
#endif
    // corner points only matter when the origin is invalid (one_leg.cu:109-116)
    const uint32_t corner_keep = overall ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < LRM_N_CORNERS; i++) {
        if (i < L.n_ucorners) { // wave-uniform
            const float vx = x - L.ucorner_x[i], vy = y - L.ucorner_y[i];
            const float ad = LRM_FAST_SQRT(__builtin_fmaf(vy, vy, vx * vx));
            insert((((lrm_f2u(ad) & ~15u) & corner_keep) | (0x7f800000u & ~corner_keep)) | (uint32_t)(LRM_N_CIRCLES + i));
        }
    }
    const float av = lrm_u2f(a & ~15u), bv = lrm_u2f(b & ~15u), cv = lrm_u2f(c & ~15u);
    // the keys drop 4 mantissa bits (< 2e-6 relative); the tie band grows by that much
    const float tie = __builtin_fmaf(av, 4.0e-6f, 2.0f * band);
    uint32_t lu = 0;
    lu |= !(macc > band) ? 1u : 0u;               // region or point-in-circle decision in doubt
    lu |= !(maccq > L.band_q) ? 4u : 0u;          // clamp-point validity in doubt
    lu |= !(magmin > 0.01f) ? 16u : 0u;           // at a circle centre the strict clamp switches direction (one_leg.cu:54-58)
    lu |= !(cv - av > tie) ? 32u : 0u;            // no candidate at all (inf - inf), or a three-way near-tie
    unc |= lu;
    unc |= !(macc > 2.0f * band) ? 512u : 0u;     // statistic + the fused reach mask below: `overall` has less than twice the margin
    lu |= T.force_strict;
#if defined(__HIP_DEVICE_COMPILE__)
    if (kCoop) {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        unsigned long long req = __ballot(lu != 0u);
        bool mine = false, res = false;
        float rx = 0.f, ry = 0.f;
        while (req) { // wave-uniform
            const int w = __builtin_ctzll(req);
            req &= req - 1ull;
            float qx = lrm_u2f((uint32_t)__builtin_amdgcn_readlane((int)lrm_f2u(x_in), w));
            float qy = lrm_u2f((uint32_t)__builtin_amdgcn_readlane((int)lrm_f2u(y_in), w));
            const bool ov = lrm_plane_dist_coop(L, T.lists, qx, qy, lane);
            if (lane == w) { mine = true; res = ov; rx = qx; ry = qy; }
        }
        if (mine) {
            x = rx;
            y = ry;
            return res;
        }
    } else
#endif
    if (lu) { // strict evaluation of this call (rare: a few 1e-4 of the calls)
        x = x_in;
        y = y_in;
        return lrm_plane_dist(L, T.lists, x, y);
    }
    // strict arithmetic for the winner (and the runner-up when the filter cannot separate them)
    const uint32_t win = a & 15u, win2 = b & 15u;
    float bx = x, by = y, d1;
    {
        const LrmCircle w = (win < LRM_N_CIRCLES) ? T.lists[reg + win] : T.corners[win - LRM_N_CIRCLES];
        bool v;
        lrm_clamp_on(w.x, w.y, w.r, w.attract != 0.f, bx, by, d1, v);
    }
    if (!(bv - av > tie)) {
        unc |= 256u; // statistic: local exact tie-break
        const LrmCircle w = (win2 < LRM_N_CIRCLES) ? T.lists[reg + win2] : T.corners[win2 - LRM_N_CIRCLES];
        float b2x = x, b2y = y, d2;
        bool v;
        lrm_clamp_on(w.x, w.y, w.r, w.attract != 0.f, b2x, b2y, d2, v);
        // the strict loop keeps the earlier entry unless the later one is strictly closer
        const bool first_is_win = win < win2;
        const float df = first_is_win ? d1 : d2, dl = first_is_win ? d2 : d1;
        const bool later_wins = fabsf(df) > fabsf(dl);
        const bool take2 = (first_is_win == later_wins);
        bx = take2 ? b2x : bx;
        by = take2 ? b2y : by;
    }
    x -= bx;
    y -= by;
    return overall;
}

// finish_finding_closest<bool> (one_leg.cu:215-278); `angle` is the strict atan2f value, so
// every comparison on it is the strict comparison.
template <bool kCoop = false>
LRM_HD bool lrm_finish_closest_fast(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& p, float angle,
                                    uint32_t& unc) {
    const bool mega = (angle > L.mega_hi) || (angle < L.mega_lo);
    float sat;
    if (mega) sat = (angle > 0) ? angle - LRM_PI_F : angle + LRM_PI_F;
    else sat = fmaxf(fminf(angle, L.max_coxa), L.min_coxa);
    const bool saturated = sat != angle;
    const float limit = (angle > L.coxa_mid) ? L.max_coxa : L.min_coxa;
    float s, c;
    lrm_sincosf(-sat, &s, &c);
    float buffer = p.x * s;
    p.x = p.x * c - p.y * s;
    p.y = buffer + p.y * c;
    const LrmVec3 save = p;
    const bool was_valid = lrm_plane_dist_fast<kCoop>(L, T, p.x, p.z, unc);
    if (was_valid && !mega) {
        // Is the nearer yaw-limit half-plane closer than the in-plane boundary?  Filter first:
        // |save.x*sin(th) + save.y*cos(th)| against |p|, th = -(limit - sat).
        const float th = -(limit - sat);
        const float Sxy = fabsf(save.x) + fabsf(save.y);
        float nrm, d_lim;
        {
#pragma clang fp contract(fast)
            nrm = LRM_FAST_SQRT(p.x * p.x + p.y * p.y + p.z * p.z);
#if defined(__HIP_DEVICE_COMPILE__)
            const float s2 = __sinf(th), c2 = __cosf(th); // hardware sin/cos: abs error < 2e-6
#else
            const float s2 = sinf(th), c2 = cosf(th);
#endif
            d_lim = fabsf(save.x * s2 + save.y * c2);
        }
        const float band2 = LRM_BAND * (Sxy + fabsf(p.z) + L.fast_scale) + 8.0e-6f * Sxy;
        const bool near = !(fabsf(nrm - d_lim) > band2);
        if (near || nrm > d_lim) {
            // strict arithmetic (one_leg.cu:258-272): decides when `near`, and produces the output
            unc |= near ? 64u : 0u;
            // A candidate clamped to a yaw limit has th = -(limit - sat) = -0: sincosf(+-0) = (+-0, 1)
            // exactly, and those are most of the lanes that come here with the flipped candidate --
            // the polynomial only runs when some lane of the wave has a real angle.
            float s2 = th, c2 = 1.0f;
            if (th != 0.f) lrm_sincosf(th, &s2, &c2);
            const float sy = save.x * s2 + save.y * c2;
            LrmVec3 lim = {0.f, sy, 0.f};
            if (lrm_norm3(p) > lrm_norm3(lim)) {
                const float b2 = lim.y * s2;
                lim.y = -lim.x * s2 + lim.y * c2;
                lim.x = lim.x * c2 + b2;
                p = lim;
            }
        }
    }
    buffer = p.y * s;
    p.y = -p.x * s + p.y * c;
    p.x = p.x * c + buffer;
    return was_valid && !saturated;
}

// What the fused reach+distance evaluation learns about reachability_global on the way (see
// lrm_reach_from_dist).
struct LrmDistByproduct {
    bool res, resflip;      // finish_finding_closest of the direct / flipped candidate
    uint32_t unc_flip;      // doubt bits of the flipped candidate's plane evaluation
    float ax, ang, ang_flip; // coxa-frame x, the two candidate yaws
};

LRM_HD bool lrm_dist_circles_fast(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& r, uint32_t& unc,
                                  LrmDistByproduct* by = nullptr) {
    LrmVec3 a = r;
    a.x -= L.body;
    float buffer = a.x * L.sin_pitch;
    a.x = a.x * L.cos_pitch - a.z * L.sin_pitch;
    a.z = buffer + a.z * L.cos_pitch;
    LrmVec3 b = a;
    const float ax = a.x;
    const float ang = lrm_atan2f(a.y, a.x);
    const float ang_flip = (ang > 0) ? ang - LRM_PI_F : ang + LRM_PI_F;
    // one copy of the candidate evaluation, executed twice (halves the code the wave walks through)
    bool res = false, resflip = false;
    uint32_t unc_flip = 0;
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        LrmVec3 p = k ? b : a;
        uint32_t u = 0;
        const bool r_k = lrm_finish_closest_fast(LRM_FRESH(L), T, p, k ? ang_flip : ang, u);
        unc |= u;
        if (k) { b = p; resflip = r_k; unc_flip = u; }
        else { a = p; res = r_k; }
    }
    if (by) *by = LrmDistByproduct{res, resflip, unc_flip, ax, ang, ang_flip};
    // The two candidates are often the same configuration up to rounding (yaw within 30 deg of
    // the axis: one of them is "mega-saturated" onto the other): the strict comparison of the
    // strict norms is the only way to pick the same one.
    const bool use_direct = (res == resflip) ? (lrm_norm3(a) < lrm_norm3(b)) : res;
    r = use_direct ? a : b;
    const LrmCompiledLeg& Le = LRM_FRESH(L);
    buffer = r.x * Le.sin_pitch_rev;
    r.x = r.x * Le.cos_pitch_rev - r.z * Le.sin_pitch_rev;
    r.z = buffer + r.z * Le.cos_pitch_rev;
    return res || resflip;
}

// reachability_global (one_leg.cu:280-319) from the by-products of distance_global on the same point.
//
// Both functions bring the point into the coxa frame with the same operations (qtRotate, z-rotation,
// place_over_coxa), so they see the same (x, y, z) bit for bit.  reachability mirrors the point into
// x >= 0, takes angle = atan2f(y, x) there, rejects it outside [min_coxa, max_coxa], cancels the yaw
// with sincosf(-angle) and validates (fx - coxa_length, z) against the region's four circles.
//  * sign bit of x clear: that is, operation for operation, the distance's DIRECT candidate -- same
//    atan2f call, "saturated" <=> outside the yaw range, same sincosf argument when unsaturated, same
//    rotated abscissa, and `was_valid` is the AND of the same four circle predicates
//    (force_clamp_on_circle's `valid` = distance_to_circumf's, one_leg.cu:31-63) in the same region.
//    So reach == res, identically (no band involved).
//  * sign bit set: the mirrored yaw atan2f(-y, -x) and the flipped candidate's ang -+ pi differ by
//    rounding (<= 2 ulp of pi), so res_flip is a FILTER for reach: it is the answer when the yaw is
//    not within 2e-6 rad of a limit and, if the candidate is unsaturated, its plane evaluation kept
//    twice the usual margin (the abscissa moves by < 1e-6 relative between the two yaws); otherwise
//    the caller re-evaluates with the strict lrm_reach_global.
//  * nan yaw (nan coordinates): doubt.
LRM_HD bool lrm_reach_from_dist(const LrmCompiledLeg& L, const LrmDistByproduct& by, bool& doubt) {
    const bool flipped = lrm_f2u(by.ax) >> 31;
    const float lim_margin = fminf(fabsf(by.ang_flip - L.max_coxa), fabsf(by.ang_flip - L.min_coxa));
    const bool flip_doubt = !(lim_margin > 2.0e-6f) || ((by.unc_flip & (1u | 512u)) != 0u);
    doubt = !(fabsf(by.ang) <= 4.0f) || (flipped && flip_doubt);
    return flipped ? by.resflip : by.res;
}

LRM_HD bool lrm_dist_global_fast(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& p, uint32_t& unc,
                                 LrmDistByproduct* by = nullptr) {
    LrmVec3 u = lrm_qrot(L.inv_rot, p);
    float buffer = u.x * L.sin_body;
    u.x = u.x * L.cos_body - u.y * L.sin_body;
    u.y = buffer + u.y * L.cos_body;
    const bool r = lrm_dist_circles_fast(L, T, u, unc, by);
    const LrmCompiledLeg& Le = LRM_FRESH(L);
    buffer = u.x * -Le.sin_body;
    u.x = u.x * Le.cos_body - u.y * -Le.sin_body;
    u.y = buffer + u.y * Le.cos_body;
    p = lrm_qrot(Le.fwd_rot, u);
    return r;
}

// ---------------------------------------------------------------------------------------
// Self-contained filtered entry points: the strict re-evaluation is inside, so callers get
// the strict result bit for bit whatever the bands decide.
// ---------------------------------------------------------------------------------------
LRM_HD bool lrm_reach_global_filtered(const LrmCompiledLeg& L, const LrmCircle* lists,
                                      const LrmCompiledLeg::LeanCircle* lean, LrmVec3 p) {
    uint32_t unc = 0;
    bool r = lrm_reach_global_fast(L, lean, p, unc);
    if (unc) r = lrm_reach_global(L, lists, p);
    return r;
}

LRM_HD bool lrm_reachable_rotate_leg_filtered(const LrmCompiledLeg& L, const LrmCircle* lists,
                                              const LrmCompiledLeg::LeanCircle* lean, LrmVec3 t, LrmVec3 body) {
    uint32_t unc = 0;
    bool r = lrm_reachable_rotate_leg_fast(L, lean, t, body, unc);
    if (unc) r = lrm_reachable_rotate_leg(L, lists, t, body);
    return r;
}

LRM_HD bool lrm_dist_global_filtered(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& p) {
    uint32_t stat = 0; // the distance filter resolves its own doubts (see lrm_plane_dist_fast)
    return lrm_dist_global_fast(L, T, p, stat);
}

// reachability_global AND distance_global of one point: the mask comes out of the distance
// evaluation (lrm_reach_from_dist), the strict reachability only runs for its rare doubts.
LRM_HD bool lrm_reach_dist_global_filtered(const LrmCompiledLeg& L, const LrmDistTables T, LrmVec3& p, bool& reach) {
    const LrmVec3 p_in = p;
    uint32_t stat = 0;
    LrmDistByproduct by;
    const bool v = lrm_dist_global_fast(L, T, p, stat, &by);
    bool doubt;
    reach = lrm_reach_from_dist(L, by, doubt);
    if (doubt) reach = lrm_reach_global(L, T.lists, p_in);
    return v;
}

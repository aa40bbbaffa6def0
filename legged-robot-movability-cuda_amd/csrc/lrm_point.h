// lrm_point.h -- per-point reachability / distance in REFERENCE ORDER (LRM_MODE_STRICT).
//
// One source for the device kernels and for the CPU entry points (the reference does the
// same with its `__host__ __device__` functions).  Compile with -ffp-contract=off: every
// float operation below is a separate IEEE operation in the order the reference performs
// it, so that together with lrm_exact_math.h the result is bit-identical to the
// reference's host path.  All leg-only quantities come precomputed in LrmCompiledLeg.
//
// `lists` points at the 4x4 circle table (LDS copy on the device, L.lists on the host).
#pragma once
#include "lrm_exact_math.h"
#include "lrm_types.h"

#define LRM_MARGIN_F 0.001f // (double)v < 0.001  <=>  v < 0.001f for every float v
#define LRM_PI_F 3.14159265358979323846264338327950288419716939937510582097f

struct LrmVec3 {
    float x, y, z;
};

// qtRotate with precomputed coefficient sums (unified_math_cuda.cu.h:13-27)
LRM_HD LrmVec3 lrm_qrot(const float* m, LrmVec3 v) {
    LrmVec3 r;
    r.x = 2.0f * (m[0] * v.x + m[1] * v.y + m[2] * v.z) + v.x;
    r.y = 2.0f * (m[3] * v.x + m[4] * v.y + m[5] * v.z) + v.y;
    r.z = 2.0f * (m[6] * v.x + m[7] * v.y + m[8] * v.z) + v.z;
    return r;
}

// distance_to_circumf, one_leg.cu:31-41
LRM_HD bool lrm_circle_valid(const LrmCircle c, float x, float y) {
    x -= c.x;
    y -= c.y;
    const float mag = lrm_sqrtf(x * x + y * y);
    const float d = c.r - mag;
    const bool inside = !(lrm_f2u(d) >> 31);
    return (inside == (c.attract != 0.f)) || (fabsf(d) < LRM_MARGIN_F);
}

// multi_circle_validate<true>(.., MAX_CIRCLES), one_leg.cu:65-89
LRM_HD bool lrm_all_valid(const LrmCircle* list, float x, float y) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < LRM_N_CIRCLES; i++) ok = ok && lrm_circle_valid(list[i], x, y);
    return ok;
}

// find_region, circles.cu.h:48-78 -> index of the circle list
LRM_HD int lrm_region(const LrmCompiledLeg& L, float x, float y) {
    const float angle = lrm_atan2f(y, x);
    const bool upper = angle > L.region_mid;
    const bool more = angle > (upper ? L.full_sat[1] : L.full_sat[0]);
    const bool fe = upper != more;
    return (upper ? 2 : 0) + (fe ? 1 : 0);
}

// reachability_circles, one_leg.cu:280-319 (point already in "leg 0" body frame)
LRM_HD bool lrm_reach_circles(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3 p) {
    // place_over_coxa, one_leg.cu:9-24
    p.x -= L.body;
    float buffer = p.x * L.sin_pitch;
    p.x = p.x * L.cos_pitch - p.z * L.sin_pitch;
    p.z = buffer + p.z * L.cos_pitch;
    // one_leg.cu:290-303: yaw of the point mirrored into x >= 0
    const bool flip = lrm_f2u(p.x) >> 31;
    const float ax = flip ? -p.x : p.x;
    const float ay = flip ? -p.y : p.y;
    const float angle = lrm_atan2f(ay, ax);
    if ((angle > L.max_coxa) || (angle < L.min_coxa)) return false;
    // cancel_coxa_rotation, one_leg.cu:146-156
    float s, c;
    lrm_sincosf(-angle, &s, &c);
    const float fx = p.x * c - p.y * s;
    // eval_plane_circles<REACH>, one_leg.cu:167-183
    const float px = fx - L.coxa_length;
    const int reg = lrm_region(L, px, p.z);
    return lrm_all_valid(lists + reg * LRM_N_CIRCLES, px, p.z);
}

// reachability_global, one_leg_global.cu:103-130
LRM_HD bool lrm_reach_global(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3 p) {
    LrmVec3 u = lrm_qrot(L.inv_rot, p);
    const float buffer = u.x * L.sin_body; // z_rotateInPlace, one_leg_global.cu:25-31
    u.x = u.x * L.cos_body - u.y * L.sin_body;
    u.y = buffer + u.y * L.cos_body;
    return lrm_reach_circles(L, lists, u);
}

// reachable_rotate_leg, several_leg.cu:48-67
LRM_HD bool lrm_reachable_rotate_leg(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3 t,
                                     LrmVec3 body) {
    t.x -= body.x;
    t.y -= body.y;
    t.z -= body.z;
    const LrmVec3 g = lrm_qrot(L.inv_rot, t);
    const float gx = g.x * L.cos_body - g.y * L.sin_body; // rotateInPlace(.., -body_angle)
    if (gx < 0) return false;
    const float buffer = t.x * L.sin_body;
    t.x = t.x * L.cos_body - t.y * L.sin_body;
    t.y = buffer + t.y * L.cos_body;
    return lrm_reach_circles(L, lists, t);
}

// force_clamp_on_circle, one_leg.cu:42-63
LRM_HD void lrm_clamp_on(float cx, float cy, float cr, bool attract, float& x, float& y, float& d,
                         bool& valid) {
    x -= cx;
    y -= cy;
    float mag = lrm_sqrtf(x * x + y * y);
    d = cr - mag;
    const bool inside = !(lrm_f2u(d) >> 31);
    valid = (inside == attract) || (fabsf(d) < LRM_MARGIN_F);
    if (mag < LRM_MARGIN_F) {
        x = 1;
        y = 0;
        mag = 1;
    }
    const float k = cr / mag;
    x = cx + x * k;
    y = cy + y * k;
}

// eval_plane_circles<DIST> + multi_circle_clamp, one_leg.cu:91-145, :167-208
LRM_HD bool lrm_plane_dist(const LrmCompiledLeg& L, const LrmCircle* lists, float& x, float& y) {
    x -= L.coxa_length;
    const LrmCircle* list = lists + lrm_region(L, x, y) * LRM_N_CIRCLES;
    bool overall = true;
    float best_x = 0, best_y = 0;
    float best_d = 999999999999999.9f;
#pragma unroll
    for (int i = 0; i < LRM_N_CIRCLES; i++) {
        const LrmCircle c = list[i];
        float cx = x, cy = y, d;
        bool valid;
        lrm_clamp_on(c.x, c.y, c.r, c.attract != 0.f, cx, cy, d, valid);
        const bool clamp_ok = lrm_all_valid(list, cx, cy);
        overall = overall && valid;
        if (clamp_ok && (fabsf(best_d) > fabsf(d))) {
            best_d = d;
            best_x = cx;
            best_y = cy;
        }
    }
    if (!overall) { // corner points only matter when the origin is invalid (one_leg.cu:109-116)
        for (int i = 0; i < L.n_corners; i++) {
            float cx = x, cy = y, d;
            bool valid;
            lrm_clamp_on(L.corner_x[i], L.corner_y[i], 0.f, true, cx, cy, d, valid);
            if (fabsf(best_d) > fabsf(d)) {
                best_d = d;
                best_x = cx;
                best_y = cy;
            }
        }
    }
    x -= best_x;
    y -= best_y;
    return overall;
}

LRM_HD float lrm_norm3(LrmVec3 v) { return lrm_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

// finish_finding_closest<bool>, one_leg.cu:215-278
LRM_HD bool lrm_finish_closest(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3& p,
                               float angle) {
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
    LrmVec3 save = p;
    const bool was_valid = lrm_plane_dist(L, lists, p.x, p.z);
    if (was_valid && !mega) {
        float s2, c2;
        lrm_sincosf(-(limit - sat), &s2, &c2);
        // cancel_coxa_rotation(save, ..) then save.x = save.z = 0 (one_leg.cu:262-265)
        const float sy = save.x * s2 + save.y * c2;
        const float d_clamped = lrm_norm3(p);
        LrmVec3 lim = {0.f, sy, 0.f};
        const float d_limit = lrm_norm3(lim);
        if (d_clamped > d_limit) {
            // restore_coxa_rotation(save, c2, s2) on (0, sy, 0)
            const float b2 = lim.y * s2;
            lim.y = -lim.x * s2 + lim.y * c2;
            lim.x = lim.x * c2 + b2;
            p = lim;
        }
    }
    // restore_coxa_rotation, one_leg.cu:158-165
    buffer = p.y * s;
    p.y = -p.x * s + p.y * c;
    p.x = p.x * c + buffer;
    return was_valid && !saturated;
}

// distance_circles, one_leg.cu:321-341
LRM_HD bool lrm_dist_circles(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3& r) {
    LrmVec3 a = r;
    a.x -= L.body;
    float buffer = a.x * L.sin_pitch;
    a.x = a.x * L.cos_pitch - a.z * L.sin_pitch;
    a.z = buffer + a.z * L.cos_pitch;
    LrmVec3 b = a;
    const float ang = lrm_atan2f(a.y, a.x);
    const float ang_flip = (ang > 0) ? ang - LRM_PI_F : ang + LRM_PI_F;
    const bool res = lrm_finish_closest(L, lists, a, ang);
    const bool resflip = lrm_finish_closest(L, lists, b, ang_flip);
    const bool use_direct = (res == resflip) ? (lrm_norm3(a) < lrm_norm3(b)) : res;
    r = use_direct ? a : b;
    // place_over_coxa<Reverse>, one_leg.cu:9-24
    buffer = r.x * L.sin_pitch_rev;
    r.x = r.x * L.cos_pitch_rev - r.z * L.sin_pitch_rev;
    r.z = buffer + r.z * L.cos_pitch_rev;
    return res || resflip;
}

// distance_global, one_leg_global.cu:74-101
LRM_HD bool lrm_dist_global(const LrmCompiledLeg& L, const LrmCircle* lists, LrmVec3& p) {
    LrmVec3 u = lrm_qrot(L.inv_rot, p);
    float buffer = u.x * L.sin_body;
    u.x = u.x * L.cos_body - u.y * L.sin_body;
    u.y = buffer + u.y * L.cos_body;
    const bool r = lrm_dist_circles(L, lists, u);
    // z_unrotateInPlace, one_leg_global.cu:33-39
    buffer = u.x * -L.sin_body;
    u.x = u.x * L.cos_body - u.y * -L.sin_body;
    u.y = buffer + u.y * L.cos_body;
    p = lrm_qrot(L.fwd_rot, u);
    return r;
}

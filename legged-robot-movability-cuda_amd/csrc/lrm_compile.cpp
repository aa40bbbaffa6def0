// lrm_compile.cpp -- hoists every leg-only / orientation-only quantity of the reference's
// per-point code into one LrmCompiledLeg block (host, once per (leg, quaternion)).
//
// The reference recomputes all of this for every point: 4 circles (insert_circles,
// circles.cu.h:337-383: ~8 sin/cos + 1 sqrt), <=10 corner points (insert_intersecv2,
// circles.cu.h:417-476: 40 sin/cos), the oriented tibia limits (rotate_leg_data,
// one_leg_global.cu:48-60) and three sincosf of leg constants.  Computing them here with the
// same expressions and the same libm calls yields the same floats, so hoisting does not
// change a single result bit.  Compile without FMA contraction / fast-math.
#include "lrm_compile.h"
#include "lrm_point_fast.h" // LRM_BAND, LRM_BAND_DIST
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr float kPi = 3.14159265358979323846264338327950288419716939937510582097f;
constexpr double kEps = 0.001; // circles.cu.h:7 (a double literal in the reference)

struct Quat {
    float x, y, z, w;
};

// qtInvert, unified_math_cuda.cu.h:29-34
Quat q_invert(Quat q) {
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return Quat{q.x / n2, -q.y / n2, -q.z / n2, -q.w / n2};
}

// qtMultiply, unified_math_cuda.cu.h:40-46
Quat q_mul(Quat a, Quat b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

// quatFromVectAngle, unified_math_cuda.cu.h:48-57, for the z axis
Quat q_about_z(float angle) {
    float s, c;
    sincosf(angle / 2, &s, &c);
    const float mag = sqrtf(0.f * 0.f + 0.f * 0.f + 1.f * 1.f);
    return Quat{s, c * 0.f / mag, c * 0.f / mag, c * 1.f / mag};
}

// the nine coefficient sums of qtRotate, unified_math_cuda.cu.h:13-27
void rot_coefficients(Quat q, float m[9]) {
    const float t2 = q.x * q.y, t3 = q.x * q.z, t4 = q.x * q.w;
    const float t5 = -q.y * q.y, t6 = q.y * q.z, t7 = q.y * q.w;
    const float t8 = -q.z * q.z, t9 = q.z * q.w, t10 = -q.w * q.w;
    m[0] = t8 + t10; m[1] = t6 - t4; m[2] = t3 + t7;
    m[3] = t4 + t6;  m[4] = t5 + t10; m[5] = t9 - t2;
    m[6] = t7 - t3;  m[7] = t2 + t9; m[8] = t5 + t8;
}

// pitch component of rpyFromQuat, unified_math_cuda.cu.h:59-83
float pitch_of(Quat q) {
    const double sinp = 2 * (q.w * q.y - q.z * q.x); // float product, widened
    if (std::fabs(sinp) >= 1) return copysignf((float)(M_PI / 2), (float)sinp);
    return (float)std::asin(sinp);
}

LrmCircle circle(float x, float y, float r, bool attract) { return LrmCircle{x, y, r, attract ? 1.f : 0.f}; }

// circles.cu.h:80-135 + leg_geometry.cu.h:12-50
LrmCircle inner(const LrmLegDimensions& l) {
    const float x = l.femur_length + l.tibia_length * cosf(l.min_angle_tibia);
    const float y = l.tibia_length * sinf(l.min_angle_tibia);
    return circle(0.f, 0.f, sqrtf(x * x + y * y), false);
}
LrmCircle outer(const LrmLegDimensions& l) { return circle(0.f, 0.f, l.tibia_length + l.femur_length, true); }
LrmCircle from_above(const LrmLegDimensions& l, bool positive) {
    const float a = positive ? l.tibia_absolute_pos : l.tibia_absolute_neg;
    return circle(l.tibia_length * cosf(a), l.tibia_length * sinf(a), l.femur_length, false);
}
LrmCircle winglet(const LrmLegDimensions& l, bool lower_side) {
    const float a = lower_side ? l.min_angle_femur : l.max_angle_femur;
    return circle(cosf(a) * l.femur_length, sinf(a) * l.femur_length, l.tibia_length, false);
}

// The leg-only half of find_region (circles.cu.h:56-68) for a given UpperRegion bit.
struct SideFlags {
    bool femur_limits;       // FemurAngleLimitation
    bool femur_limits_other; // FemurAngleLimitation_other
    float full_sat;          // full_sat_limit
};
SideFlags side_flags(const LrmLegDimensions& d, bool upper) {
    const float femur_limit = upper ? d.max_angle_femur : d.min_angle_femur;
    const float abs_limit = upper ? d.tibia_absolute_pos : d.tibia_absolute_neg;
    const float femur_limit_o = !upper ? d.max_angle_femur : d.min_angle_femur;
    const float abs_limit_o = !upper ? d.tibia_absolute_pos : d.tibia_absolute_neg;
    SideFlags f;
    f.femur_limits = (!upper) != (femur_limit < abs_limit);
    f.femur_limits_other = (!upper) != (femur_limit_o < abs_limit_o);
    f.full_sat = f.femur_limits ? femur_limit : abs_limit;
    return f;
}

// insert_circles (MegaClamp == 0), circles.cu.h:337-383, for one of the 4 possible regions
void circle_list(const LrmLegDimensions& l, bool upper, bool fully_ext, LrmCircle out[4]) {
    const SideFlags f = side_flags(l, upper);
    out[0] = inner(l);
    LrmCircle* tail = out + 1; // [0] fromabove_neg slot, [1] fromabove_pos slot, [2] winglet slot
    tail[0] = from_above(l, false);
    tail[1] = from_above(l, true);
    const int excluded = upper ? 0 : 1;
    if (f.femur_limits_other) tail[excluded] = winglet(l, /*lower_side=*/upper);
    tail[excluded].attract = 0.f;
    const int other = upper ? 1 : 0;
    tail[2] = winglet(l, /*lower_side=*/!upper);
    tail[other].attract = f.femur_limits ? 0.f : 1.f;
    tail[2].attract = f.femur_limits ? 1.f : 0.f;
    if (fully_ext) tail[(tail[other].attract != 0.f) ? other : 2] = outer(l);
}

// insert_intersecv2, circles.cu.h:417-476
int corner_points(const LrmLegDimensions& l, float* xs, float* ys) {
    const float fem[10] = {l.min_angle_femur, l.min_angle_femur, l.min_angle_femur,
                           l.tibia_absolute_neg - l.min_angle_tibia,
                           l.tibia_absolute_neg - l.max_angle_tibia,
                           l.max_angle_femur, l.max_angle_femur, l.max_angle_femur,
                           l.tibia_absolute_pos - l.min_angle_tibia,
                           l.tibia_absolute_pos - l.min_angle_tibia};
    const float tib[10] = {l.max_angle_tibia, l.min_angle_tibia, l.tibia_absolute_neg - fem[2],
                           l.tibia_absolute_neg - fem[3], l.tibia_absolute_neg - fem[4],
                           l.min_angle_tibia, l.max_angle_tibia, l.tibia_absolute_pos - fem[7],
                           l.tibia_absolute_pos - fem[8], l.tibia_absolute_pos - fem[9]};
    int n = 0;
    for (int i = 0; i < 10; i++) {
        const float f = fem[i], t = tib[i], a = f + t;
        const bool ok = ((double)f < (double)l.max_angle_femur + kEps) && ((double)f > (double)l.min_angle_femur - kEps) &&
                        ((double)t < (double)l.max_angle_tibia + kEps) && ((double)t > (double)l.min_angle_tibia - kEps) &&
                        ((double)a < (double)l.tibia_absolute_pos + kEps) && ((double)a > (double)l.tibia_absolute_neg - kEps);
        if (!ok) continue;
        const float xf = l.femur_length * cosf(f), yf = l.femur_length * sinf(f);
        const float xt = l.tibia_length * cosf(a), yt = l.tibia_length * sinf(a);
        xs[n] = xf + xt;
        ys[n] = yf + yt;
        n++;
    }
    return n;
}

} // namespace

void lrm_host_rotate_leg_data(const float quat[4], const LrmLegDimensions& leg, LrmLegDimensions* out) {
    const Quat q{quat[0], quat[1], quat[2], quat[3]};
    const Quat qa = q_about_z(leg.body_angle);
    const float pitch = pitch_of(q_mul(q_mul(qa, q), q_invert(qa)));
    *out = leg;
    out->tibia_absolute_pos -= pitch;
    out->tibia_absolute_neg -= pitch;
}

void lrm_host_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia,
                          float tibia2femur, float femur2tip, float coxa_angle_deg,
                          float femur_angle_deg, float tibia_angle_deg, float tib_abs_pos,
                          float tib_abs_neg, LrmLegDimensions* out) {
    // leg_factory, static_variables.cpp:6-42
    LrmLegDimensions leg;
    std::memset(&leg, 0, sizeof leg);
    leg.coxa_pitch = coxa_pitch_deg / 180.f * kPi;
    leg.body = body2coxa;
    leg.coxa_length = coxa2tibia;
    leg.femur_length = tibia2femur;
    leg.tibia_length = femur2tip;
    leg.tibia_absolute_pos = tib_abs_pos / 180.0f * kPi - leg.coxa_pitch;
    leg.tibia_absolute_neg = (-180.0f - tib_abs_neg) / 180.0f * kPi - leg.coxa_pitch;
    leg.max_angle_coxa = kPi / 180.0f * coxa_angle_deg;
    leg.min_angle_coxa = -kPi / 180.0f * coxa_angle_deg;
    leg.max_angle_femur = kPi / 180.0f * femur_angle_deg;
    leg.min_angle_femur = -kPi / 180.0f * femur_angle_deg;
    leg.max_angle_tibia = kPi / 180.0f * tibia_angle_deg;
    leg.min_angle_tibia = -kPi / 180.0f * tibia_angle_deg;
    leg.body_angle = azimut;
    *out = leg;
}

void lrm_compile_leg(const LrmLegDimensions& leg_in, const float quat[4], int apply_leg_rotation,
                     LrmCompiledLeg* out) {
    std::memset(out, 0, sizeof *out);
    LrmLegDimensions l = leg_in;
    if (apply_leg_rotation) lrm_host_rotate_leg_data(quat, leg_in, &l);
    const Quat q{quat[0], quat[1], quat[2], quat[3]};

    for (int u = 0; u < 2; u++)
        for (int fe = 0; fe < 2; fe++) circle_list(l, u != 0, fe != 0, out->lists[u * 2 + fe]);
    out->n_corners = corner_points(l, out->corner_x, out->corner_y);

    rot_coefficients(q_invert(q), out->inv_rot);
    rot_coefficients(q, out->fwd_rot);
    sincosf(-l.body_angle, &out->sin_body, &out->cos_body);
    out->body = l.body;
    sincosf(-l.coxa_pitch, &out->sin_pitch, &out->cos_pitch);
    sincosf(l.coxa_pitch, &out->sin_pitch_rev, &out->cos_pitch_rev);
    out->coxa_length = l.coxa_length;
    out->max_coxa = l.max_angle_coxa;
    out->min_coxa = l.min_angle_coxa;
    out->mega_hi = l.max_angle_coxa + kPi / 2;
    out->mega_lo = l.min_angle_coxa - kPi / 2;
    out->coxa_mid = (l.max_angle_coxa + l.min_angle_coxa) / 2;
    // circles.cu.h:52-54 (std::max / std::min)
    const float lo = (l.tibia_absolute_neg < l.min_angle_femur) ? l.min_angle_femur : l.tibia_absolute_neg;
    const float hi = (l.max_angle_femur < l.tibia_absolute_pos) ? l.max_angle_femur : l.tibia_absolute_pos;
    out->region_mid = (lo + hi) / 2;
    out->full_sat[0] = side_flags(l, false).full_sat;
    out->full_sat[1] = side_flags(l, true).full_sat;

    // ---- filtered-evaluation constants (double precision on the host, rounded once) ----
    const double margin = 0.001;
    double scale = 0;
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < LRM_N_CIRCLES; i++) {
            const LrmCircle& ci = out->lists[k][i];
            auto& f = out->flists[k][i];
            const double r = ci.r;
            if (ci.attract != 0.f) {
                f.T = (float)((r + margin) * (r + margin));
                f.sg = 1.f;
            } else {
                const double lo = r - margin;
                f.T = (lo > 0) ? (float)(lo * lo) : -1.f; // r <= margin: always valid (m > -1)
                f.sg = -1.f;
            }
            f.g = (float)(2 * (r + margin));
            f.pad = 0.f;
            const double sc = std::fabs((double)ci.x) + std::fabs((double)ci.y) + r;
            if (sc > scale) scale = sc;
        }
    out->fast_scale = (float)scale;
    const float dirs[5] = {out->region_mid, out->full_sat[0], out->full_sat[1], out->max_coxa, out->min_coxa};
    for (int i = 0; i < 5; i++) {
        out->dir_cos[i] = (float)std::cos((double)dirs[i]);
        out->dir_sin[i] = (float)std::sin((double)dirs[i]);
    }
    // region table of the filters: the boolean form of "atan2f(y, x) > C" (lrm_gt_from_t) evaluated
    // for the 16 sign patterns of (t_mid, t_s0, t_s1, y)
    out->region_lut = 0;
    for (unsigned pat = 0; pat < 16; pat++) {
        const bool ypos = !(pat & 8u);
        auto gt = [&](unsigned bit, float c) {
            const bool tp = !(pat & bit), c_nonneg = c >= 0.f;
            return (tp && ypos) || (!c_nonneg && (tp || ypos));
        };
        const bool upper = gt(1u, out->region_mid);
        const bool more = upper ? gt(4u, out->full_sat[1]) : gt(2u, out->full_sat[0]);
        const unsigned reg = (upper ? 2u : 0u) + ((upper != more) ? 1u : 0u);
        out->region_lut |= reg << (2 * pat);
    }
    // The cross-product form of the coxa range test needs both limits well inside
    // (-pi/2, pi/2) (the yaw is measured on the point mirrored into x >= 0); the region
    // constants must be finite and away from +-pi.  Anything else: strict path.
    const double half_pi = 1.5707963267948966, pi = 3.141592653589793;
    bool ok = std::isfinite(scale) && scale > 0 && scale < 2000.0; // own-circle clamp validity needs 4u*r << margin
    ok = ok && std::fabs((double)out->max_coxa) < half_pi - 1e-3 && std::fabs((double)out->min_coxa) < half_pi - 1e-3;
    ok = ok && out->min_coxa < out->max_coxa;
    for (int i = 0; i < 3; i++) ok = ok && std::isfinite(dirs[i]) && std::fabs((double)dirs[i]) < pi - 1e-3;
    for (int k = 0; k < 4 && ok; k++)
        for (int i = 0; i < LRM_N_CIRCLES; i++) ok = ok && out->lists[k][i].r > 0.01f; // no degenerate circle
    out->fast_ok = ok ? 1 : 0;
    // corner points without bit-identical repeats (the reference lists one configuration twice,
    // circles.cu.h:447-450; a repeat can never win the "strictly closer" test of one_leg.cu:135)
    out->n_ucorners = 0;
    for (int i = 0; i < out->n_corners; i++) {
        bool dup = false;
        for (int j = 0; j < out->n_ucorners; j++)
            dup = dup || (out->ucorner_x[j] == out->corner_x[i] && out->ucorner_y[j] == out->corner_y[i]);
        if (!dup) {
            out->ucorner_x[out->n_ucorners] = out->corner_x[i];
            out->ucorner_y[out->n_ucorners] = out->corner_y[i];
            out->n_ucorners++;
        }
    }

    // lean reach filter: affine maps (double on the host, from the same float coefficients the
    // strict code multiplies with) and 16-byte circle records
    {
        double Rq[3][3], Rz[3][3] = {{out->cos_body, -(double)out->sin_body, 0}, {out->sin_body, out->cos_body, 0}, {0, 0, 1}};
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) Rq[r][k] = 2.0 * out->inv_rot[3 * r + k] + (r == k ? 1.0 : 0.0);
        const double Rp[3][3] = {{out->cos_pitch, 0, -(double)out->sin_pitch}, {0, 1, 0}, {out->sin_pitch, 0, out->cos_pitch}};
        double RzRq[3][3], G[3][3], P[3][3];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                RzRq[r][k] = 0;
                for (int j = 0; j < 3; j++) RzRq[r][k] += Rz[r][j] * Rq[j][k];
            }
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                G[r][k] = P[r][k] = 0;
                for (int j = 0; j < 3; j++) {
                    G[r][k] += Rp[r][j] * RzRq[j][k];
                    P[r][k] += Rp[r][j] * Rz[j][k];
                }
            }
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) {
                out->aff_global[4 * r + k] = (float)G[r][k];
                out->aff_pair[4 * r + k] = (float)P[r][k];
            }
            const float tr = (float)(-Rp[r][0] * (double)l.body);
            out->aff_global[4 * r + 3] = tr;
            out->aff_pair[4 * r + 3] = tr;
        }
        for (int k = 0; k < 3; k++) out->grav_row[k] = (float)RzRq[0][k];
        // Per-leg bounding sphere of the pair test.  In the coxa frame a reachable foothold has its
        // (mirrored) yaw in [min_coxa, max_coxa] and lies within L = femur + tibia (+ slack) of the
        // femur joint in its meridian plane: (r - c)^2 + z^2 <= L^2 with r >= 0, c = coxa_length, or,
        // for a point behind the coxa axis (the reference mirrors it and evaluates px = -r - c),
        // (r' + c)^2 + z^2 <= L^2 at yaw + pi.  For a centre at radius rho on the bisector of the yaw
        // range (half-width al) the squared distance is linear in r on each part, so its maximum is
        //   max( rho^2 + L^2 - c^2 + 2 (L - c)+ (rho - c)+ ,  (c + L)^2 - 2 rho (c + L) cos(al) + rho^2 ).
        {
            const double c = l.coxa_length, L = ((double)l.femur_length + (double)l.tibia_length + 1.0) * 1.0001;
            const double th = 0.5 * ((double)out->max_coxa + (double)out->min_coxa);
            const double ca = std::cos(0.5 * ((double)out->max_coxa - (double)out->min_coxa));
            auto radius2 = [&](double rho) {
                const double f1 = rho * rho + L * L - c * c + 2.0 * std::max(L - c, 0.0) * std::max(rho - c, 0.0);
                const double f2 = (c + L) * (c + L) - 2.0 * rho * (c + L) * ca + rho * rho;
                return std::max(f1, f2);
            };
            double rho = 0.0, r2 = (std::fabs(c) + L) * (std::fabs(c) + L); // the whole ball around the coxa origin
            if (ca > 0.05 && c > 0 && std::isfinite(th)) {
                const double cand[3] = {c, 2.0 * c * L / (std::max(L - c, 0.0) + (c + L) * ca), c / ca};
                for (double rc : cand)
                    if (radius2(rc) < r2) {
                        r2 = radius2(rc);
                        rho = rc;
                    }
            }
            // centre in the coxa frame -> relative to the body position: x_coxa = Rp (Rz t - (body, 0, 0))
            const double cc[3] = {rho * std::cos(th), rho * std::sin(th), 0.0};
            double v[3];
            for (int k = 0; k < 3; k++) v[k] = Rp[0][k] * cc[0] + Rp[1][k] * cc[1] + Rp[2][k] * cc[2];
            v[0] += (double)l.body;
            for (int k = 0; k < 3; k++) out->pair_center[k] = (float)(Rz[0][k] * v[0] + Rz[1][k] * v[1] + Rz[2][k] * v[2]);
            const double rr = std::sqrt(r2) + 1.0; // + 1 mm: float rounding of the centre and of the test itself
            out->pair_r2 = (float)(rr * rr * 1.0001);
        }
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < LRM_N_CIRCLES; i++) {
                auto& q = out->lean[k][i];
                q.x = out->lists[k][i].x;
                q.y = out->lists[k][i].y;
                const double gs = (double)out->flists[k][i].sg / (double)out->flists[k][i].g;
                q.gs = (float)gs;
                q.c = (float)(-(double)out->flists[k][i].T * gs);
            }
    }

    // lean distance filter tables
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < LRM_N_CIRCLES; i++) {
            auto& d = out->dist_tab[k][i];
            const LrmCircle& ci = out->lists[k][i];
            d.x = ci.x; d.y = ci.y; d.gs = out->lean[k][i].gs; d.c = out->lean[k][i].c;
            d.r = ci.r; d.pad[0] = d.pad[1] = 0.f;
            int a = 0;
            for (int j = 0; j < LRM_N_CIRCLES; j++) {
                if (j == i) continue;
                const LrmCircle& cj = out->lists[k][j];
                const double ex = (double)ci.x - (double)cj.x, ey = (double)ci.y - (double)cj.y;
                const double gsj = (double)out->flists[k][j].sg / (double)out->flists[k][j].g;
                const double P = 2.0 * (double)ci.r * gsj;
                d.arc[a].ex = (float)(ex * P); // the direction comes pre-scaled by P: one multiply less per arc
                d.arc[a].ey = (float)(ey * P);
                d.arc[a].Q = (float)((ex * ex + ey * ey + (double)ci.r * (double)ci.r - (double)out->flists[k][j].T) * gsj);
                a++;
            }
        }
    for (int i = 0; i < LRM_N_CORNERS; i++)
        out->corner_tab[i] = (i < out->n_ucorners) ? circle(out->ucorner_x[i], out->ucorner_y[i], 0.f, true)
                                                   : circle(0.f, 0.f, 0.f, true);
    out->band_q = (float)((double)LRM_BAND_DIST * 2.0 * (double)out->fast_scale);

    // One band for every test of the lean reach filter, linear in the L1 size of the input
    // point: the coxa-frame coordinates obey |x|+|y|+|z| <= sqrt(3) (|p|_1 + body), the affine map
    // is within 22u (|p|_1 + body) of the strict chain, the hardware sqrt within 8u r:
    //   S = fast_scale + (sqrt(3) + 1.5) (|p|_1 + body),  band = LRM_BAND * S.
    {
        const double kBand = (double)LRM_BAND;
        const double slope = 1.7320508 + 1.5;
        out->band_base = (float)(kBand * ((double)out->fast_scale + slope * std::fabs((double)l.body)));
        out->band_slope = (float)(kBand * slope);
    }

    // Nothing farther than the stretched leg (+1 mm and 1e-4 relative slack, three orders of
    // magnitude above the float rounding of the strict evaluation) can pass the attractive
    // circle test, so pairs beyond this radius are skipped without changing any result.
    const float reach = l.body + l.coxa_length + l.femur_length + l.tibia_length + 1.0f;
    out->reach_r2_max = reach * reach * 1.0001f;
}

// ---------------------------------------------------------------------------------------------
// Tolerance mode tables (lrm_point_tol.h)
// ---------------------------------------------------------------------------------------------
namespace {

// Signed distance (mm) of the clamp point of circle i at direction `th` to the validity boundary
// of the other three circles of its list: max_j sg_j (|q - c_j| - thr_j), q = c_i + r_i (cos th, sin th);
// the clamp point passes multi_circle_validate (one_leg.cu:65-89) <=> the value is negative
// (it is always valid for its own circle: |d| ~ 0 < CIRCLE_MARGIN).
double clamp_validity_dir(const LrmCircle* list, int i, double c, double s) { // (c, s) = the direction's cosine and sine
    const double margin = 0.001;
    const double qx = (double)list[i].x + (double)list[i].r * c;
    const double qy = (double)list[i].y + (double)list[i].r * s;
    double g = -1e300;
    for (int j = 0; j < LRM_N_CIRCLES; j++) {
        if (j == i) continue;
        const bool attract = list[j].attract != 0.f;
        const double thr = attract ? (double)list[j].r + margin : (double)list[j].r - margin;
        const double ex = qx - (double)list[j].x, ey = qy - (double)list[j].y;
        const double mag = std::sqrt(ex * ex + ey * ey);
        g = std::max(g, attract ? mag - thr : thr - mag);
    }
    return g;
}
double clamp_validity(const LrmCircle* list, int i, double th) { return clamp_validity_dir(list, i, std::cos(th), std::sin(th)); }
// the directions of clamp_arc's safety net: the same 4096 for every circle of every leg
constexpr int kArcSamples = 4096;
struct ArcDirs {
    double c[kArcSamples], s[kArcSamples];
    ArcDirs() {
        for (int k = 0; k < kArcSamples; k++) {
            const double th = 6.283185307179586 * (k + 0.5) / kArcSamples;
            c[k] = std::cos(th);
            s[k] = std::sin(th);
        }
    }
};
const ArcDirs& arc_dirs() {
    static const ArcDirs d; // (thread-safe initialisation)
    return d;
}

// The directions for which the clamp point of circle i is valid, as ONE arc {u : u . m >= chw} when that is
// what the set is.  eps (mm): how far the reference's float clamp point and its float validation can be from
// the exact ones; directions whose clamp point is within eps of a validity boundary get a doubt band `bw` on
// w = u . m - chw.  Returns false when the set is not a single arc (or empty / everything) or when a
// boundary is approached without being crossed (grazing circles).
#define TOL_REJECT(why)                                                                               \
    do {                                                                                              \
        if (std::getenv("LRM_TOL_DEBUG")) std::fprintf(stderr, "lrm_compile_tol: circle %d: %s\n", i, why); \
        return false;                                                                                 \
    } while (0)
// [qa, qb] (qa <= qb, radians, qb - qa >= 2 pi: everything): the only directions the per-point code can ask
// about -- for a circle centred on the femur joint the clamp direction is the direction of the point itself,
// which find_region confines to the region's sector.
bool clamp_arc(const LrmCircle* list, int i, double eps, double qa, double qb, LrmTolLeg::Circle* out) {
    const double two_pi = 6.283185307179586;
    const bool everything = (qb - qa) >= two_pi;
    auto in_query = [&](double th) { // th in any branch
        if (everything) return true;
        double t = std::fmod(th - qa, two_pi);
        if (t < 0) t += two_pi;
        return t <= (qb - qa);
    };
    const double margin = 0.001;
    const double ri = list[i].r;
    double cand[2 * LRM_N_CIRCLES];
    int nc = 0;
    for (int j = 0; j < LRM_N_CIRCLES; j++) {
        if (j == i) continue;
        const double ex = (double)list[i].x - (double)list[j].x, ey = (double)list[i].y - (double)list[j].y;
        const double E = std::hypot(ex, ey);
        if (!(E > 0)) continue; // concentric: the validity does not depend on the direction
        const double thr = (list[j].attract != 0.f) ? (double)list[j].r + margin : (double)list[j].r - margin;
        const double kappa = (thr * thr - E * E - ri * ri) / (2.0 * ri * E);
        if (!(std::fabs(kappa) < 1.0)) continue;
        const double phi = std::atan2(ey, ex), a = std::acos(kappa);
        cand[nc++] = std::fmod(phi + a + 2 * two_pi, two_pi);
        cand[nc++] = std::fmod(phi - a + 2 * two_pi, two_pi);
    }
    std::sort(cand, cand + nc);
    // valid / invalid on the elementary intervals between consecutive candidates
    double arc_a = 0, arc_b = 0;
    int n_arcs = 0;
    bool all_valid = false;
    if (nc == 0) {
        all_valid = clamp_validity(list, i, 0.0) < 0;
    } else {
        bool val[2 * LRM_N_CIRCLES];
        for (int k = 0; k < nc; k++) {
            const double lo = cand[k], hi = (k + 1 < nc) ? cand[k + 1] : cand[0] + two_pi;
            val[k] = clamp_validity(list, i, 0.5 * (lo + hi)) < 0;
        }
        bool any = false, every = true;
        for (int k = 0; k < nc; k++) { any = any || val[k]; every = every && val[k]; }
        if (every) all_valid = true;
        else if (any) {
            // arcs = maximal cyclic runs of valid intervals
            for (int k = 0; k < nc; k++) {
                const int prev = (k + nc - 1) % nc;
                if (val[k] && !val[prev]) { // a run starts at cand[k]
                    int e = k;
                    while (val[(e + 1) % nc] && (e + 1 - k) < nc) e++;
                    const double a0 = cand[k];
                    double b0 = cand[(e + 1) % nc];
                    if (b0 <= a0) b0 += two_pi;
                    bool hit = everything;
                    if (!hit) { // does [a0, b0] meet [qa, qb] (mod 2 pi)?  Either holds an end of the other.
                        double t = std::fmod(a0 - qa, two_pi);
                        if (t < 0) t += two_pi;
                        hit = t <= (qb - qa);
                        t = std::fmod(qa - a0, two_pi);
                        if (t < 0) t += two_pi;
                        hit = hit || t <= (b0 - a0);
                    }
                    if (!hit) continue;
                    n_arcs++;
                    arc_a = a0;
                    arc_b = b0;
                }
            }
        }
    }
    if (std::getenv("LRM_TOL_DEBUG")) {
        std::fprintf(stderr, "circle %d (%.3f %.3f r %.3f a %.0f): %d cand, %d arcs [%f %f] all %d:", i, list[i].x, list[i].y, list[i].r, list[i].attract, nc, n_arcs, arc_a, arc_b, (int)all_valid);
        for (int k = 0; k < nc; k++) std::fprintf(stderr, " %.4f(%+.3g)", cand[k], clamp_validity(list, i, 0.5 * (cand[k] + ((k + 1 < nc) ? cand[k + 1] : cand[0] + two_pi))));
        std::fprintf(stderr, "\n");
    }
    out->bw = 0.f;
    double mid = 0, hw = 0;
    if (n_arcs == 0) {
        out->mx = 1.f; out->my = 0.f;
        out->chw = all_valid ? -2.f : 2.f;
        if (!all_valid && !everything && nc) {
            // no valid arc meets the query range; an arc elsewhere must not leak in: "never" is right as long as
            // the safety net below (restricted to the range) agrees
        }
    } else if (n_arcs == 1) {
        mid = 0.5 * (arc_a + arc_b);
        hw = 0.5 * (arc_b - arc_a);
        out->mx = (float)std::cos(mid);
        out->my = (float)std::sin(mid);
        out->chw = (float)std::cos(hw);
    } else {
        TOL_REJECT("the valid directions are more than one arc");
    }
    // doubt band around the two arc ends: the largest offset at which the clamp point is still within eps
    // of the validity boundary (geometric scan), doubled
    double dmax = 0;
    if (n_arcs == 1) {
        const double ends[2] = {arc_a, arc_b};
        double w_band = 0;
        for (double e : ends)
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                if (!in_query(e)) continue; // this end is never asked about
                double last = 0;
                for (double d = 1e-9; d < 0.05; d *= 1.25)
                    if (std::fabs(clamp_validity(list, i, e + sgn * d)) < eps) last = d;
                if (last > 0.02) TOL_REJECT("a validity boundary is nearly tangent at an arc end");
                last = 2.0 * last + 1e-9;
                dmax = std::max(dmax, last);
                w_band = std::max(w_band, std::fabs(std::cos(e + sgn * last - mid) - std::cos(hw)));
            }
        if (hw < 2 * dmax || (3.141592653589793 - hw) < 2 * dmax) TOL_REJECT("a sliver of an arc (or of a gap)");
        out->bw = (float)(w_band + 1e-6); // + the float evaluation of w itself (a few ulp of 1)
    }
    // safety net on a dense sample: the arc form agrees with the direct evaluation wherever that is certain
    const ArcDirs& dirs = arc_dirs();
    for (int s = 0; s < kArcSamples; s++) {
        const double th = two_pi * (s + 0.5) / kArcSamples;
        if (!in_query(th)) continue;
        const double g = clamp_validity_dir(list, i, dirs.c[s], dirs.s[s]);
        const double w = (double)out->mx * dirs.c[s] + (double)out->my * dirs.s[s] - (double)out->chw;
        if (std::fabs(g) < eps) {
            if (!(std::fabs(w) < (double)out->bw)) TOL_REJECT("an uncertain direction outside the doubt band");
        } else if ((g < 0) != (w >= 0)) {
            if (!(std::fabs(w) < (double)out->bw)) TOL_REJECT("the arc form disagrees with the direct evaluation");
        }
    }
    return true;
}

} // namespace

void lrm_compile_tol(const LrmCompiledLeg& L, LrmTolLeg* out) {
    std::memset(out, 0, sizeof *out);
    bool ok = L.fast_ok != 0;
    // How far the reference's float clamp point (force_clamp_on_circle one_leg.cu:42-63: sub, sqrt, div, mul, add)
    // and its float validation (distance_to_circumf :31-41) can be from the exact ones: a few ulp of the plane
    // coordinates each; 16u * fast_scale (~3e-4 mm) -- below CIRCLE_MARGIN, so that the designed tangencies
    // (a from-above circle touching the outer circle from inside, valid by the margin alone) stay certain.
    const double eps = 16.0 * 5.9604645e-8 * (double)L.fast_scale;
    const double pi = 3.141592653589793, slack = 2e-3;
    // the sector of directions find_region (circles.cu.h:48-78) assigns to list k = upper * 2 + fully_extended
    double q_lo[4], q_hi[4];
    {
        const double mid = L.region_mid, fs0 = L.full_sat[0], fs1 = L.full_sat[1];
        q_lo[0] = -pi;                 q_hi[0] = std::min(mid, fs0); // lower, not fully extended
        q_lo[1] = fs0;                 q_hi[1] = mid;                // lower, fully extended
        q_lo[2] = std::max(mid, fs1);  q_hi[2] = pi;                 // upper, not fully extended
        q_lo[3] = mid;                 q_hi[3] = fs1;                // upper, fully extended
    }
    double r_outer = 0;
    double qa_of[16], qb_of[16];
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < LRM_N_CIRCLES; i++) {
            auto& c = out->circ[k][i];
            const LrmCircle& ci = L.lists[k][i];
            c.x = ci.x; c.y = ci.y; c.gs = L.lean[k][i].gs; c.c = L.lean[k][i].c;
            c.r = ci.r;
            c.mx = 1.f; c.my = 0.f; c.chw = 2.f; c.bw = 0.f;
            double qa = 0, qb = 7; // everything
            if (ci.x == 0.f && ci.y == 0.f) { // centred on the femur joint: the clamp direction is the point's own
                qa = q_lo[k] - slack;
                qb = std::max(q_hi[k], q_lo[k]) + slack; // an empty sector (never selected) keeps a token range
            }
            qa_of[k * LRM_N_CIRCLES + i] = qa;
            qb_of[k * LRM_N_CIRCLES + i] = qb;
            out->feat[k * LRM_N_CIRCLES + i] = ci;
            r_outer = std::max(r_outer, std::hypot((double)ci.x, (double)ci.y) + (double)ci.r);
        }
    if (ok || std::getenv("LRM_TOL_DEBUG")) {
        // the arcs of the sixteen (list, circle) pairs are independent (each intersects three caps and verifies its arc form on 4096
        // directions: most of this function's ~1.5 ms): a few host threads; a caller that changes orientation per call pays this per call
        bool arc_ok[16];
        auto work = [&](int t, int nt) {
            for (int j = t; j < 16; j += nt) arc_ok[j] = clamp_arc(L.lists[j / LRM_N_CIRCLES], j % LRM_N_CIRCLES, eps, qa_of[j], qb_of[j], &out->circ[j / LRM_N_CIRCLES][j % LRM_N_CIRCLES]);
        };
        int nt = (int)std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : (nt > 4 ? 4 : nt);
        if (std::getenv("LRM_TOL_DEBUG")) nt = 1; // (its messages in order)
        if (const char* e = std::getenv("LRM_COMPILE_THREADS")) nt = std::max(1, std::atoi(e));
        if (nt == 1) work(0, 1);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; t++) pool.emplace_back(work, t, nt);
            for (auto& th : pool) th.join();
        }
        for (int j = 0; j < 16; j++) ok = arc_ok[j] && ok;
    }
    // corner points: exact repeats are already gone; near-repeats (closer than 1e-4 mm: a tenth of the
    // smallest tie band) would put every point whose nearest target they are in doubt -- keep the first,
    // as the reference's strict "closer than" does for exact ties
    out->n_corners = 0;
    for (int i = 0; i < L.n_ucorners; i++) {
        bool dup = false;
        for (int j = 0; j < out->n_corners; j++)
            dup = dup || std::hypot((double)out->feat[16 + j].x - (double)L.ucorner_x[i],
                                    (double)out->feat[16 + j].y - (double)L.ucorner_y[i]) < 1e-4;
        if (dup) continue;
        out->feat[16 + out->n_corners] = LrmCircle{L.ucorner_x[i], L.ucorner_y[i], 0.f, 1.f};
        r_outer = std::max(r_outer, std::hypot((double)L.ucorner_x[i], (double)L.ucorner_y[i]));
        out->n_corners++;
    }
    for (int i = out->n_corners; i < LRM_N_CORNERS; i++) out->feat[16 + i] = LrmCircle{0.f, 0.f, 0.f, 1.f};
    for (int i = 0; i < 12; i++) out->aff[i] = L.aff_global[i];
    {   // coxa-frame vector -> output frame: place_over_coxa<Reverse> (one_leg.cu:17), z_unrotateInPlace
        // (one_leg_global.cu:33-39), qtRotate(q, .) (one_leg_global.cu:99), composed in double
        const double c = L.cos_pitch_rev, s = L.sin_pitch_rev, cb = L.cos_body, sb = L.sin_body;
        const double Rp[3][3] = {{c, 0, -s}, {0, 1, 0}, {s, 0, c}};
        const double Rz[3][3] = {{cb, sb, 0}, {-sb, cb, 0}, {0, 0, 1}};
        double Rq[3][3], T[3][3];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) Rq[r][k] = 2.0 * L.fwd_rot[3 * r + k] + (r == k ? 1.0 : 0.0);
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                T[r][k] = 0;
                for (int j = 0; j < 3; j++) T[r][k] += Rz[r][j] * Rp[j][k];
            }
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) {
                double v = 0;
                for (int j = 0; j < 3; j++) v += Rq[r][j] * T[j][k];
                out->back[3 * r + k] = (float)v;
            }
    }
    out->yaw_cs[0] = L.dir_cos[3]; out->yaw_cs[1] = L.dir_sin[3];
    out->yaw_cs[2] = L.dir_cos[4]; out->yaw_cs[3] = L.dir_sin[4];
    for (int i = 0; i < 3; i++) { out->dir_cos[i] = L.dir_cos[i]; out->dir_sin[i] = L.dir_sin[i]; }
    out->region_lut = L.region_lut;
    out->coxa_length = L.coxa_length;
    out->band_base = L.band_base;
    out->band_slope = L.band_slope;
    out->r_outer = (float)(r_outer * 1.0001 + 0.01);
    out->tol_ok = ok ? 1 : 0;
}

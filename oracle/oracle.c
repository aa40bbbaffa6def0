/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the reference's HOST arithmetic for the batched 3-DoF leg
 * reachability / distance path.  Every function cites the reference lines it follows
 * (paths are relative to /root/reference).  The restatement keeps the reference's
 * operation order, its float/double promotions and its quirks, so that the result is
 * bit-identical to the reference sources compiled by g++ (oracle/_ref, see Makefile).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no -ffast-math, no -march=native).
 *
 * Promotion notes (they decide bits):
 *   - CIRCLE_MARGIN and EPS are `double` literals 0.001 (settings.h:9, circles.cu.h:7):
 *     every comparison against them is carried out in double.
 *   - PI / pI / pIgpu are `float` literals (settings.h:16, HeaderCPP.h:7, HeaderCUDA.h:20).
 *   - cos/sin/sqrt/abs on float arguments resolve to the float overloads under nvcc's
 *     host pass (global-namespace overloads from <math.h>), i.e. cosf/sinf/sqrtf/fabsf.
 *   - rpyFromQuat mixes float products with double atan2/asin (unified_math_cuda.cu.h:59-83).
 */
#define _GNU_SOURCE
#include "oracle.h"
#include <math.h>
#include <string.h>

#define ORC_MARGIN 0.001 /* double: settings.h:9 CIRCLE_MARGIN */
#define ORC_EPS 0.001    /* double: circles.cu.h:7 EPS */
static const float ORC_PI_F = 3.14159265358979323846264338327950288419716939937510582097f;

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float x, y, radius; int attract; } circ;
typedef struct { int upper, fully_ext, fem_lim, fem_lim_other; } region_t;

#define ORC_N_CIRCLES 4   /* circles.cu.h:8-14 MAX_CIRCLES */
#define ORC_N_CORNERS 10  /* circles.cu.h:15 MAX_INTERSECT */

/* std::max / std::min semantics (return first argument on ties / NaN in second) */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline float std_min(float a, float b) { return (b < a) ? b : a; }

/* ------------------------------------------------------------------------------------
 * unified_math_cuda.cu.h
 * ---------------------------------------------------------------------------------- */

/* qtRotate, unified_math_cuda.cu.h:13-27 (q.x plays the scalar part) */
static v3 qt_rotate(v4 q, v3 v) {
    float t2 = q.x * q.y;
    float t3 = q.x * q.z;
    float t4 = q.x * q.w;
    float t5 = -q.y * q.y;
    float t6 = q.y * q.z;
    float t7 = q.y * q.w;
    float t8 = -q.z * q.z;
    float t9 = q.z * q.w;
    float t10 = -q.w * q.w;
    v3 r;
    r.x = 2.0f * ((t8 + t10) * v.x + (t6 - t4) * v.y + (t3 + t7) * v.z) + v.x;
    r.y = 2.0f * ((t4 + t6) * v.x + (t5 + t10) * v.y + (t9 - t2) * v.z) + v.y;
    r.z = 2.0f * ((t7 - t3) * v.x + (t2 + t9) * v.y + (t5 + t8) * v.z) + v.z;
    return r;
}

/* qtInvert, unified_math_cuda.cu.h:29-34 */
static v4 qt_invert(v4 q) {
    float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    v4 r = {q.x / n2, -q.y / n2, -q.z / n2, -q.w / n2};
    return r;
}

/* qtMultiply, unified_math_cuda.cu.h:40-46 (here .w is the scalar part: kept as is) */
static v4 qt_multiply(v4 a, v4 b) {
    float w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    float x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    float y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    float z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    v4 r = {x, y, z, w};
    return r;
}

/* quatFromVectAngle, unified_math_cuda.cu.h:48-57 (sine lands in .x: kept as is) */
static v4 quat_from_vect_angle(v3 axis, float angle) {
    float s, c;
    sincosf(angle / 2, &s, &c);
    float mag = sqrtf(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
    v4 r = {s, c * axis.x / mag, c * axis.y / mag, c * axis.z / mag};
    return r;
}

/* rpyFromQuat, unified_math_cuda.cu.h:59-83: float products widened to double,
 * double atan2/asin, results narrowed to float.  Only .y (pitch) is consumed. */
static float pitch_from_quat(v4 q) {
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    double sinp = 2 * (w * y - z * x); /* float arithmetic, then widened */
    float pitch;
    if (fabs(sinp) >= 1)
        pitch = copysignf((float)(M_PI / 2), (float)sinp);
    else
        pitch = (float)asin(sinp);
    return pitch;
}

/* linorm (host branch), unified_math_cuda.cu.h:147-157: sqrt of (x*x + y*y) + z*z */
static inline float linorm(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

/* ------------------------------------------------------------------------------------
 * leg_geometry.cu.h + circle constructors circles.cu.h:80-135
 * ---------------------------------------------------------------------------------- */

/* inner_circle circles.cu.h:80-87 with min_femur_to_gripper_dist<LOWER_SIDE>
 * leg_geometry.cu.h:12-25 */
static circ inner_circle(const orc_leg_t* leg) {
    float ta = leg->min_angle_tibia;
    float x = leg->femur_length + leg->tibia_length * cosf(ta);
    float y = leg->tibia_length * sinf(ta);
    circ c = {0.f, 0.f, sqrtf(x * x + y * y), 0};
    return c;
}

/* outer_circle circles.cu.h:89-96, max_femur_to_gripper_dist leg_geometry.cu.h:27-30 */
static circ outer_circle(const orc_leg_t* leg) {
    circ c = {0.f, 0.f, leg->tibia_length + leg->femur_length, 1};
    return c;
}

/* fromabove_{pos,neg}_circle circles.cu.h:98-114 (attractivity left to the caller) */
static circ fromabove_circle(const orc_leg_t* leg, int positive) {
    float a = positive ? leg->tibia_absolute_pos : leg->tibia_absolute_neg;
    circ c = {leg->tibia_length * cosf(a), leg->tibia_length * sinf(a), leg->femur_length, 0};
    return c;
}

/* winglet_circle circles.cu.h:116-122 + saturated_femur leg_geometry.cu.h:32-37 */
static circ winglet_circle(const orc_leg_t* leg, int lower_side) {
    float a = lower_side ? leg->min_angle_femur : leg->max_angle_femur;
    circ c = {cosf(a) * leg->femur_length, sinf(a) * leg->femur_length, leg->tibia_length, 0};
    return c;
}

/* find_region circles.cu.h:48-78 */
static region_t find_region(float x, float y, const orc_leg_t* d) {
    region_t r;
    float angle = atan2f(y, x);
    float middle = (std_max(d->tibia_absolute_neg, d->min_angle_femur) +
                    std_min(d->tibia_absolute_pos, d->max_angle_femur)) / 2;
    r.upper = angle > middle;
    float femur_limit = r.upper ? d->max_angle_femur : d->min_angle_femur;
    float abs_limit = r.upper ? d->tibia_absolute_pos : d->tibia_absolute_neg;
    float femur_limit_o = (!r.upper) ? d->max_angle_femur : d->min_angle_femur;
    float abs_limit_o = (!r.upper) ? d->tibia_absolute_pos : d->tibia_absolute_neg;
    r.fem_lim = (!r.upper) ^ (femur_limit < abs_limit);
    r.fem_lim_other = (!r.upper) ^ (femur_limit_o < abs_limit_o);
    float full_sat = r.fem_lim ? femur_limit : abs_limit;
    int more_than_sat = angle > full_sat;
    r.fully_ext = r.upper ^ more_than_sat;
    return r;
}

/* insert_circles circles.cu.h:337-383, MegaClamp == 0 branch: always 4 entries */
static void insert_circles(const orc_leg_t* leg, region_t reg, circ* list) {
    list[0] = inner_circle(leg);
    circ* tail = list + 1;
    const int lower_side = !reg.upper;
    enum { NEG = 0, POS = 1, WING = 2 };
    tail[NEG] = fromabove_circle(leg, 0);
    tail[POS] = fromabove_circle(leg, 1);
    int excl = reg.upper ? NEG : POS;
    if (reg.fem_lim_other) tail[excl] = winglet_circle(leg, !lower_side);
    tail[excl].attract = 0;
    int other = (!reg.upper) ? NEG : POS;
    tail[WING] = winglet_circle(leg, lower_side);
    tail[other].attract = !reg.fem_lim;
    tail[WING].attract = reg.fem_lim;
    if (reg.fully_ext) {
        int idx = tail[other].attract ? other : WING;
        tail[idx] = outer_circle(leg);
    }
}

/* insert_intersecv2 circles.cu.h:417-476: <=10 corner points (entries 8 and 9 are the
 * same configuration in the reference: kept).  Returns the number appended. */
static int insert_corners(const orc_leg_t* leg, circ* tail) {
    float fem[10], tib[10];
    fem[0] = leg->min_angle_femur; tib[0] = leg->max_angle_tibia;
    fem[1] = leg->min_angle_femur; tib[1] = leg->min_angle_tibia;
    fem[2] = leg->min_angle_femur; tib[2] = leg->tibia_absolute_neg - fem[2];
    fem[3] = leg->tibia_absolute_neg - leg->min_angle_tibia; tib[3] = leg->tibia_absolute_neg - fem[3];
    fem[4] = leg->tibia_absolute_neg - leg->max_angle_tibia; tib[4] = leg->tibia_absolute_neg - fem[4];
    fem[5] = leg->max_angle_femur; tib[5] = leg->min_angle_tibia;
    fem[6] = leg->max_angle_femur; tib[6] = leg->max_angle_tibia;
    fem[7] = leg->max_angle_femur; tib[7] = leg->tibia_absolute_pos - fem[7];
    fem[8] = leg->tibia_absolute_pos - leg->min_angle_tibia; tib[8] = leg->tibia_absolute_pos - fem[8];
    fem[9] = leg->tibia_absolute_pos - leg->min_angle_tibia; tib[9] = leg->tibia_absolute_pos - fem[9];
    int n = 0;
    for (int i = 0; i < 10; i++) {
        float f = fem[i], t = tib[i];
        int fem_ok = ((double)f < (double)leg->max_angle_femur + ORC_EPS) &&
                     ((double)f > (double)leg->min_angle_femur - ORC_EPS);
        int tib_ok = ((double)t < (double)leg->max_angle_tibia + ORC_EPS) &&
                     ((double)t > (double)leg->min_angle_tibia - ORC_EPS);
        float ab = f + t;
        int abs_ok = ((double)ab < (double)leg->tibia_absolute_pos + ORC_EPS) &&
                     ((double)ab > (double)leg->tibia_absolute_neg - ORC_EPS);
        if (fem_ok && tib_ok && abs_ok) {
            float xf = leg->femur_length * cosf(f);
            float yf = leg->femur_length * sinf(f);
            float xt = leg->tibia_length * cosf(ab);
            float yt = leg->tibia_length * sinf(ab);
            tail[n].x = xf + xt;
            tail[n].y = yf + yt;
            tail[n].radius = 0.f;
            tail[n].attract = 1;
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------
 * one_leg.cu
 * ---------------------------------------------------------------------------------- */

/* place_over_coxa<Reverse> one_leg.cu:9-24 */
static void place_over_coxa(v3* p, const orc_leg_t* d, int reverse) {
    if (!reverse) p->x -= d->body;
    float s, c;
    if (reverse) sincosf(d->coxa_pitch, &s, &c);
    else sincosf(-d->coxa_pitch, &s, &c);
    float buffer = p->x * s;
    p->x = p->x * c - p->z * s;
    p->z = buffer + p->z * c;
}

/* distance_to_circumf one_leg.cu:31-41 */
static int circle_valid(const circ* c, float x, float y, float* dist_out) {
    x -= c->x;
    y -= c->y;
    float mag = sqrtf(x * x + y * y);
    float dist = c->radius - mag;
    int inside = !signbit(dist);
    if (dist_out) *dist_out = dist;
    return (inside == c->attract) || ((double)fabsf(dist) < ORC_MARGIN);
}

/* force_clamp_on_circle one_leg.cu:42-63 */
static void clamp_on_circle(const circ* c, float* x, float* y, float* dist, int* valid) {
    *x -= c->x;
    *y -= c->y;
    float mag = sqrtf(*x * *x + *y * *y);
    *dist = c->radius - mag;
    int inside = !signbit(*dist);
    *valid = (inside == c->attract) || ((double)fabsf(*dist) < ORC_MARGIN);
    if ((double)mag < ORC_MARGIN) {
        *x = 1;
        *y = 0;
        mag = 1;
    }
    float k = c->radius / mag;
    *x = c->x + *x * k;
    *y = c->y + *y * k;
}

/* multi_circle_validate<true> one_leg.cu:65-89 */
static int all_circles_valid(float x, float y, const circ* list, int n) {
    for (int i = 0; i < n; i++)
        if (!circle_valid(&list[i], x, y, NULL)) return 0;
    return 1;
}

/* multi_circle_clamp one_leg.cu:91-145 (CIRCLE_ARR_ORDERED, MegaClamp == 0) */
static int clamp_on_boundary(float* x, float* y, const circ* list, int n) {
    int overall = 1;
    float best_x = 0, best_y = 0;
    float best_d = 999999999999999.9; /* narrowed to float as in the reference */
    for (int i = 0; i < n; i++) {
        const circ* c = &list[i];
        float d;
        int valid;
        float cx = *x, cy = *y;
        clamp_on_circle(c, &cx, &cy, &d, &valid);
        int clamp_ok;
        int is_point = (double)fabsf(c->radius) < ORC_MARGIN;
        if (is_point) {
            if (overall) break; /* points only matter when the origin is invalid */
            clamp_ok = 1;
        } else {
            clamp_ok = all_circles_valid(cx, cy, list, ORC_N_CIRCLES);
            overall = overall && valid;
        }
        int closer = fabsf(best_d) > fabsf(d);
        if (clamp_ok && closer) {
            best_d = d;
            best_x = cx;
            best_y = cy;
        }
    }
    *x -= best_x;
    *y -= best_y;
    return overall;
}

/* cancel_coxa_rotation one_leg.cu:146-156 */
static void cancel_coxa(v3* p, float angle, float* c, float* s) {
    sincosf(-angle, s, c);
    float buffer = p->x * *s;
    p->x = p->x * *c - p->y * *s;
    p->y = buffer + p->y * *c;
}

/* restore_coxa_rotation one_leg.cu:158-165 */
static void restore_coxa(v3* p, float c, float s) {
    float buffer = p->y * s;
    p->y = -p->x * s + p->y * c;
    p->x = p->x * c + buffer;
}

/* eval_plane_circles<REACH_USECASE> one_leg.cu:167-183 */
static int plane_reach(float x, float y, const orc_leg_t* d) {
    x -= d->coxa_length;
    region_t reg = find_region(x, y, d);
    circ list[ORC_N_CIRCLES];
    insert_circles(d, reg, list);
    return all_circles_valid(x, y, list, ORC_N_CIRCLES);
}

/* eval_plane_circles<DIST_USECASE> one_leg.cu:167-208 */
static int plane_dist(float* x, float* y, const orc_leg_t* d) {
    *x -= d->coxa_length;
    region_t reg = find_region(*x, *y, d);
    circ list[ORC_N_CIRCLES + ORC_N_CORNERS];
    insert_circles(d, reg, list);
    int n = ORC_N_CIRCLES + insert_corners(d, list + ORC_N_CIRCLES);
    return clamp_on_boundary(x, y, list, n);
}

/* finish_finding_closest<bool> one_leg.cu:215-278 */
static int finish_closest(v3* p, const orc_leg_t* d, float angle) {
    int mega = angle > (d->max_angle_coxa + ORC_PI_F / 2) || angle < (d->min_angle_coxa - ORC_PI_F / 2);
    float sat;
    if (mega) sat = (angle > 0) ? angle - ORC_PI_F : angle + ORC_PI_F;
    else sat = fmaxf(fminf(angle, d->max_angle_coxa), d->min_angle_coxa);
    int saturated = sat != angle;
    float limit = (angle > (d->max_angle_coxa + d->min_angle_coxa) / 2) ? d->max_angle_coxa
                                                                        : d->min_angle_coxa;
    float c, s;
    cancel_coxa(p, sat, &c, &s);
    v3 save = *p;
    int was_valid = plane_dist(&p->x, &p->z, d);
    if (was_valid && !mega) {
        float c2, s2;
        cancel_coxa(&save, limit - sat, &c2, &s2);
        save.x = 0;
        save.z = 0;
        float d_clamped = linorm(*p);
        float d_limit = linorm(save);
        if (d_clamped > d_limit) {
            restore_coxa(&save, c2, s2);
            *p = save;
        }
    }
    restore_coxa(p, c, s);
    return was_valid && !saturated;
}

/* reachability_circles one_leg.cu:280-319 */
static int reach_circles(v3 p, const orc_leg_t* d) {
    place_over_coxa(&p, d, 0);
    int flip = signbit(p.x) ? 1 : 0; /* -0.0 flips too (one_leg.cu:291) */
    if (flip) { p.x *= -1; p.y *= -1; }
    float angle = atan2f(p.y, p.x);
    if (flip) { p.x *= -1; p.y *= -1; }
    if ((angle > d->max_angle_coxa) || (angle < d->min_angle_coxa)) return 0;
    float c, s;
    cancel_coxa(&p, angle, &c, &s);
    return plane_reach(p.x, p.z, d);
}

/* distance_circles one_leg.cu:321-341 */
static int dist_circles(v3* result, const orc_leg_t* d) {
    v3 closest = *result;
    place_over_coxa(&closest, d, 0);
    v3 closest_flip = closest;
    float a = atan2f(closest.y, closest.x);
    float a_flip = (a > 0) ? a - ORC_PI_F : a + ORC_PI_F;
    int res = finish_closest(&closest, d, a);
    int resflip = finish_closest(&closest_flip, d, a_flip);
    int use_direct = (!(res ^ resflip)) ? (linorm(closest) < linorm(closest_flip)) : res;
    *result = use_direct ? closest : closest_flip;
    place_over_coxa(result, d, 1);
    return res || resflip;
}

/* ------------------------------------------------------------------------------------
 * one_leg_global.cu
 * ---------------------------------------------------------------------------------- */

/* z_rotateInPlace one_leg_global.cu:25-31 */
static void z_rotate(v3* p, float a, float* c, float* s) {
    sincosf(a, s, c);
    float buffer = p->x * *s;
    p->x = p->x * *c - p->y * *s;
    p->y = buffer + p->y * *c;
}

/* z_unrotateInPlace one_leg_global.cu:33-39 */
static void z_unrotate(v3* p, float c, float s) {
    float buffer = p->x * -s;
    p->x = p->x * c - p->y * -s;
    p->y = buffer + p->y * c;
}

/* rotate_leg_data one_leg_global.cu:48-60 */
static orc_leg_t rotate_leg(v4 quat, orc_leg_t leg) {
    v3 zaxis = {0, 0, 1};
    v4 qa = quat_from_vect_angle(zaxis, leg.body_angle);
    v4 r = qt_multiply(qt_multiply(qa, quat), qt_invert(qa));
    float pitch = pitch_from_quat(r);
    leg.tibia_absolute_pos -= pitch;
    leg.tibia_absolute_neg -= pitch;
    return leg;
}

/* reachability_global one_leg_global.cu:103-130 (host branch) */
static int reach_global(v3 p, const orc_leg_t* dim, v4 quat) {
    orc_leg_t ol = rotate_leg(quat, *dim);
    v3 u = qt_rotate(qt_invert(quat), p);
    float c, s;
    z_rotate(&u, -ol.body_angle, &c, &s);
    return reach_circles(u, &ol);
}

/* distance_global one_leg_global.cu:74-101 (host branch) */
static int dist_global(v3* p, const orc_leg_t* dim, v4 quat) {
    orc_leg_t ol = rotate_leg(quat, *dim);
    v3 u = qt_rotate(qt_invert(quat), *p);
    float c, s;
    z_rotate(&u, -ol.body_angle, &c, &s);
    int r = dist_circles(&u, &ol);
    z_unrotate(&u, c, s);
    *p = qt_rotate(quat, u);
    return r;
}

/* ------------------------------------------------------------------------------------
 * static_variables.cpp
 * ---------------------------------------------------------------------------------- */

/* leg_factory static_variables.cpp:6-42 (margins are unused by the reference) */
void orc_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia,
                     float tibia2femur, float femur2tip, float coxa_angle_deg,
                     float femur_angle_deg, float tibia_angle_deg, float tib_abs_pos,
                     float tib_abs_neg, orc_leg_t* out) {
    orc_leg_t leg;
    memset(&leg, 0, sizeof leg);
    leg.coxa_pitch = coxa_pitch_deg / 180.f * ORC_PI_F;
    leg.body = body2coxa;
    leg.coxa_length = coxa2tibia;
    leg.femur_length = tibia2femur;
    leg.tibia_length = femur2tip;
    leg.tibia_absolute_pos = tib_abs_pos / 180.0f * ORC_PI_F - leg.coxa_pitch;
    leg.tibia_absolute_neg = (-180.0f - tib_abs_neg) / 180.0f * ORC_PI_F - leg.coxa_pitch;
    leg.max_angle_coxa = ORC_PI_F / 180.0f * coxa_angle_deg;
    leg.min_angle_coxa = -ORC_PI_F / 180.0f * coxa_angle_deg;
    leg.max_angle_femur = ORC_PI_F / 180.0f * femur_angle_deg;
    leg.min_angle_femur = -ORC_PI_F / 180.0f * femur_angle_deg;
    leg.max_angle_tibia = ORC_PI_F / 180.0f * tibia_angle_deg;
    leg.min_angle_tibia = -ORC_PI_F / 180.0f * tibia_angle_deg;
    leg.body_angle = azimut;
    *out = leg;
}

/* get_moonbot_leg static_variables.cpp:44-67 */
void orc_get_moonbot_leg(float azimut, orc_leg_t* out) {
    orc_leg_factory(azimut, 181, 0, 65.5f, 129, 160, 60.0f, 90.0f, 120.0f, -5, -5, out);
}

/* get_M2_leg static_variables.cpp:69-93 */
void orc_get_M2_leg(float azimut, orc_leg_t* out) {
    orc_leg_factory(azimut, 181, -45, 65.5f, 129, 135, 60.0f, 90.0f, 120.0f, -5, -5, out);
}

/* ------------------------------------------------------------------------------------
 * several_leg.cu / collision.cu.h (device-only in the reference: restated from source)
 * ---------------------------------------------------------------------------------- */

/* reachable_rotate_leg several_leg.cu:48-67 */
static int reachable_rotate_leg(v3 target, v3 body, v4 q, const orc_leg_t* dim) {
    float c, s;
    target.x -= body.x;
    target.y -= body.y;
    target.z -= body.z;
    v3 g = qt_rotate(qt_invert(q), target);
    z_rotate(&g, -dim->body_angle, &c, &s); /* rotateInPlace several_leg.cu:26-33 */
    if (g.x < 0) return 0;
    z_rotate(&target, -dim->body_angle, &c, &s);
    return reach_circles(target, dim);
}

/* ------------------------------------------------------------------------------------
 * exported wrappers
 * ---------------------------------------------------------------------------------- */
static inline v3 ld3(const float* p) { v3 v = {p[0], p[1], p[2]}; return v; }
static inline v4 ld4(const float* p) { v4 v = {p[0], p[1], p[2], p[3]}; return v; }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

int orc_reachability_circles(const float p[3], const orc_leg_t* leg) { return reach_circles(ld3(p), leg); }
int orc_distance_circles(float p[3], const orc_leg_t* leg) {
    v3 v = ld3(p);
    int r = dist_circles(&v, leg);
    st3(p, v);
    return r;
}
int orc_reachability_global(const float p[3], const orc_leg_t* leg, const float quat[4]) {
    return reach_global(ld3(p), leg, ld4(quat));
}
int orc_distance_global(float p[3], const orc_leg_t* leg, const float quat[4]) {
    v3 v = ld3(p);
    int r = dist_global(&v, leg, ld4(quat));
    st3(p, v);
    return r;
}

/* reachability_kernel_cpu one_leg_global.cu:132-137 */
void orc_reach(const float* xyz, size_t n, const orc_leg_t* leg, const float quat[4], uint8_t* out) {
    v4 q = ld4(quat);
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)reach_global(ld3(xyz + 3 * i), leg, q);
}

/* distance_kernel_cpu one_leg_global.cu:139-147 */
void orc_dist(const float* xyz, size_t n, const orc_leg_t* leg, const float quat[4], float* dout,
              uint8_t* vout) {
    v4 q = ld4(quat);
    for (size_t i = 0; i < n; i++) {
        v3 v = ld3(xyz + 3 * i);
        int r = dist_global(&v, leg, q);
        st3(dout + 3 * i, v);
        if (vout) vout[i] = (uint8_t)r;
    }
}

void orc_rotate_leg_data(const float quat[4], const orc_leg_t* leg, orc_leg_t* out) {
    *out = rotate_leg(ld4(quat), *leg);
}
void orc_qt_rotate(const float q[4], const float v[3], float out[3]) { st3(out, qt_rotate(ld4(q), ld3(v))); }
void orc_qt_invert(const float q[4], float out[4]) {
    v4 r = qt_invert(ld4(q));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_qt_multiply(const float a[4], const float b[4], float out[4]) {
    v4 r = qt_multiply(ld4(a), ld4(b));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_quat_from_vect_angle(const float axis[3], float angle, float out[4]) {
    v4 r = quat_from_vect_angle(ld3(axis), angle);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

int orc_reachable_rotate_leg(const float t[3], const float b[3], const float q[4], const orc_leg_t* leg) {
    return reachable_rotate_leg(ld3(t), ld3(b), ld4(q), leg);
}

/* reach_mem_kernel semantics several_leg.cu:92-129: out[leg][body] = any over targets */
void orc_reach_any(const float* bodies, size_t nb, const float* targets, size_t nt,
                   const orc_leg_t* legs, size_t nlegs, const float quat[4], uint8_t* out) {
    v4 q = ld4(quat);
    for (size_t l = 0; l < nlegs; l++)
        for (size_t b = 0; b < nb; b++) {
            uint8_t any = 0;
            v3 body = ld3(bodies + 3 * b);
            for (size_t t = 0; t < nt && !any; t++)
                any = (uint8_t)reachable_rotate_leg(ld3(targets + 3 * t), body, q, &legs[l]);
            out[l * nb + b] = any;
        }
}

/* in_sphere collision.cu.h:5-10 */
int orc_in_sphere(float radius, const float c[3], const float t[3]) {
    float dx = c[0] - t[0], dy = c[1] - t[1], dz = c[2] - t[2];
    return sqrtf(dx * dx + dy * dy + dz * dz) < radius;
}

/* in_cylinder collision.cu.h:12-23 */
int orc_in_cylinder(float radius, float plus_z, float minus_z, const float c[3], const float t[3]) {
    float distz = t[2] - c[2];
    float dx = t[0] - c[0], dy = t[1] - c[1];
    int radial = sqrtf(dx * dx + dy * dy + 0.f) < radius;
    return radial && (distz < plus_z) && (distz > minus_z);
}

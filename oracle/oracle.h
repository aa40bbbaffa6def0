/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the reference's host (CPU) arithmetic for
 * the reach / distance hot path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (liblrm.so) never
 * links or calls it.
 *
 * Parity pin: bit-for-bit against the reference's own sources compiled with g++
 * (oracle/_ref/libref.so, built by oracle/Makefile where /root/reference exists) and
 * against the committed fixtures under tests/golden/ that were generated from it.
 */
#ifndef LRM_ORACLE_H
#define LRM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same field order and size (14 x f32 = 56 B) as reference HeaderCPP.h:19-52 */
typedef struct {
    float body_angle, body, coxa_pitch, coxa_length, tibia_length, femur_length;
    float tibia_absolute_pos, tibia_absolute_neg;
    float max_angle_coxa, min_angle_coxa;
    float max_angle_tibia, min_angle_tibia;
    float max_angle_femur, min_angle_femur;
} orc_leg_t;

/* leg factories: static_variables.cpp:6-93 */
void orc_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia,
                     float tibia2femur, float femur2tip, float coxa_angle_deg,
                     float femur_angle_deg, float tibia_angle_deg, float tib_abs_pos,
                     float tib_abs_neg, orc_leg_t* out);
void orc_get_M2_leg(float azimut, orc_leg_t* out);
void orc_get_moonbot_leg(float azimut, orc_leg_t* out);

/* one point, one_leg.cu:280-341 (no body orientation) */
int orc_reachability_circles(const float p[3], const orc_leg_t* leg);
int orc_distance_circles(float p_inout[3], const orc_leg_t* leg);

/* one point, one_leg_global.cu:74-130 (quat = {x,y,z,w} with .x the scalar slot for
 * qtRotate, exactly as the reference treats it) */
int orc_reachability_global(const float p[3], const orc_leg_t* leg, const float quat[4]);
int orc_distance_global(float p_inout[3], const orc_leg_t* leg, const float quat[4]);

/* array loops, one_leg_global.cu:132-147 generalised to any quaternion; xyz is AoS */
void orc_reach(const float* xyz, size_t n, const orc_leg_t* leg, const float quat[4],
               uint8_t* mask_out);
void orc_dist(const float* xyz, size_t n, const orc_leg_t* leg, const float quat[4],
              float* dxyz_out, uint8_t* valid_out /* may be NULL */);

/* rotate_leg_data, one_leg_global.cu:48-60 */
void orc_rotate_leg_data(const float quat[4], const orc_leg_t* leg, orc_leg_t* out);

/* quaternion helpers (unified_math_cuda.cu.h:13-57), exposed for the sweep tests */
void orc_qt_rotate(const float q[4], const float v[3], float out[3]);
void orc_qt_invert(const float q[4], float out[4]);
void orc_qt_multiply(const float a[4], const float b[4], float out[4]);
void orc_quat_from_vect_angle(const float axis[3], float angle, float out[4]);

/* reachable_rotate_leg, several_leg.cu:48-67 (one body, one target, one leg) */
int orc_reachable_rotate_leg(const float target[3], const float body[3], const float quat[4],
                             const orc_leg_t* leg);

/* brute-force body x target "any" per leg: semantics of reach_mem_kernel
 * (several_leg.cu:92-129).  out[leg*nb + b] = 1 iff some target is reachable. */
void orc_reach_any(const float* bodies, size_t nb, const float* targets, size_t nt,
                   const orc_leg_t* legs, size_t nlegs, const float quat[4], uint8_t* out);

/* in_sphere / in_cylinder, collision.cu.h:5-23 (norm3df restated as sqrtf of the sum) */
int orc_in_sphere(float radius, const float c[3], const float t[3]);
int orc_in_cylinder(float radius, float plus_z, float minus_z, const float c[3],
                    const float t[3]);

#ifdef __cplusplus
}
#endif
#endif

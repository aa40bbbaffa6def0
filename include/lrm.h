/*
 * lrm.h -- C ABI of liblrm.so: the MI355X (gfx950) implementation of the batched 3-DoF
 * (yaw-pitch-pitch) leg reachability / distance path and of the body x target
 * positionability aggregation.
 *
 * Every entry point names the interface of the reference (2lian/Legged-Robot-Movability-Cuda,
 * paths relative to its root) that it replaces.  Plain pointers and sizes only.
 * All functions returning `int` return 0 on success and a negative LRM_E* code on failure;
 * lrm_last_error() then holds a message.  Nothing here ever computes a GPU entry point on
 * the CPU: without a usable HIP device the *_dev / host-buffer GPU calls fail with LRM_ENODEV.
 *
 * Threading: like the reference's harness (single host thread, cross_compiled.cu) the library is meant to be called
 * from one host thread at a time: the arithmetic mode and the caches of compiled tables are process-wide.  Device
 * workspaces (bounding boxes and compiled-leg slots of the pair kernels, doubt queues of LRM_MODE_TOL) are kept per
 * device -- the doubt queues per (device, stream) -- created on first use and reused; a compiled-leg slot is only
 * rewritten after the launch that read it has completed (an event per slot), so the *_dev entry points may be
 * queued on several streams.  The bounding boxes of the pair kernels are one buffer per device: do not run two
 * pair launches on different clouds concurrently on one device.
 *
 * Units: millimetres and radians, float32 arithmetic (reference convention).
 * Quaternions are float[4] = {x,y,z,w} in the reference's own (inconsistent) convention:
 * qtRotate/qtInvert read [0] as the scalar part (identity = {1,0,0,0}, settings.h:51).
 */
#ifndef LRM_H
#define LRM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LRM_OK 0
#define LRM_EINVAL (-1) /* bad argument (null pointer, length mismatch, too many legs...) */
#define LRM_ENODEV (-2) /* no HIP device / HIP runtime error                               */
#define LRM_ENOMEM (-3) /* device or host allocation failed                               */

/* Arithmetic mode of the GPU kernels.
 * LRM_MODE_STRICT: reference operation order, no FMA contraction, glibc-exact atan2f/sincosf:
 *                  bit-identical to the reference's host path (reachability_kernel_cpu /
 *                  distance_kernel_cpu, one_leg_global.cu:132-147).
 * LRM_MODE_FAST:   the same outputs, bit for bit, faster.  Clouds of >= 2e5 points on legs with a plane table:
 *                  csrc/lrm_point_xtab.h -- every DECISION (clamp target, validity, which yaw candidate) comes
 *                  from the plane table of the tolerance mode with its error bands, every VALUE from the
 *                  reference's own operations in its own order (one exact atan2f, one exact sincosf and one strict
 *                  clamp per evaluated candidate); a point with a decision inside its band is re-evaluated by the
 *                  filtered code in a second small launch.  Otherwise the filtered evaluation
 *                  (csrc/lrm_point_fast.h): decisions with cheap arithmetic plus conservative bands, values that
 *                  reach an output with the strict arithmetic, a decision inside its band re-taken by the strict
 *                  code.  Legs outside the filter's eligibility silently use the strict kernels. */
#define LRM_MODE_STRICT 0
#define LRM_MODE_FAST 1
/* LRM_MODE_TOL:    contract-tolerance mode (csrc/lrm_point_tol.h): the reach mask and the distance's
 *                  validity byte stay bit-identical to LRM_MODE_STRICT; the distance VECTOR is computed with
 *                  FP32 FMA / v_rsq_f32 arithmetic (no atan2f / sincosf / IEEE sqrt) and lands on the same
 *                  boundary feature as the reference's, with the error bound (FROZEN, tests/tolcheck.py)
 *                      |d - d_ref| <= 1e-5 * max(|d_ref|, (|p| + body) / 8)      per point.
 *                  This is a floored reading of "within 1e-5 relative" (BASELINE.json): literally relative for
 *                  every vector longer than 1/8 of the coordinate scale, an absolute bound of about 10 ulp of
 *                  the coordinates (~1e-3 mm at most; measured <= 3e-4 mm) below.  Why a floor: d is a difference
 *                  of float32 positions the reference itself forms after `x -= body` (one_leg.cu:13), so it
 *                  carries ~1e-4 mm of rounding noise in ANY float32 implementation, the reference's own
 *                  -use_fast_math CUDA build included; a purely relative bound is unattainable for
 *                  |d| << 10 mm.  The literal bound |d - d_ref| <= 1e-5 |d_ref| holds for every vector of at
 *                  least 16 mm (asserted on the GPU); the fraction of shorter vectors that miss it is reported by
 *                  bench.py ("tolerance_check").  Callers that need the literal text use LRM_MODE_TOL_REL (below) or LRM_MODE_FAST (tolerance 0).
 *                  Points with any decision inside its error band are re-evaluated by the LRM_MODE_FAST code in a
 *                  second small launch and are bit-identical.  Applies to the distance / fused entry points, host
 *                  buffers (lrm_dist, lrm_reach_dist: the apply_kernel boundary) and device buffers alike;
 *                  reach-only and pair kernels run as in LRM_MODE_FAST.  Legs outside the mode's eligibility use
 *                  LRM_MODE_FAST. */
#define LRM_MODE_TOL 2
/* LRM_MODE_TOL_REL: LRM_MODE_TOL with the LITERAL bound of BASELINE.json on every vector: reach mask and validity byte
 *                  bit-identical, |d - d_ref| <= 1e-5 |d_ref| for every point.  A vector that comes out shorter than
 *                  max(19 mm, 0.034 (|p|_1 + body)) -- the absolute error of the tolerance arithmetic grows with the coordinates --
 *                  has its value chain recomputed with the reference's own operations from the decisions the tolerance
 *                  evaluation took (csrc/lrm_point_xtab.h: lrm_xtab_replay, inside the same kernel: bit-identical, relative
 *                  error 0).  Every longer vector is within 1e-5 relative by LRM_MODE_TOL's own arithmetic (measured 7.6e-6 at
 *                  most, at 17.55 mm under the first threshold of 17 mm; asserted: tests/test_gpu_tol.py).  About 5 % of a cloud filling the leg's bounding cube
 *                  is replayed (+ the 0.5 % of doubtful points of LRM_MODE_TOL through the fix-up launch); a cloud that hugs the
 *                  workspace's surface is replayed whole and runs at about half the speed.  The bench headline. */
#define LRM_MODE_TOL_REL 3

/* LegDimensions, HeaderCPP.h:19-52: 14 x f32 = 56 bytes, this field order. */
typedef struct LrmLegDimensions {
    float body_angle;
    float body;
    float coxa_pitch;
    float coxa_length;
    float tibia_length;
    float femur_length;
    float tibia_absolute_pos;
    float tibia_absolute_neg;
    float max_angle_coxa;
    float min_angle_coxa;
    float max_angle_tibia;
    float min_angle_tibia;
    float max_angle_femur;
    float min_angle_femur;
} LrmLegDimensions;

/* ---- library state ------------------------------------------------------------------ */
const char* lrm_version(void);
const char* lrm_last_error(void);
int lrm_device_count(void);          /* number of HIP devices, 0 if none (never fails)      */
int lrm_set_device(int ordinal);     /* cudaSetDevice analogue; the reference uses device 0 */
int lrm_set_mode(int mode);          /* LRM_MODE_*; process-wide; default LRM_MODE_FAST     */
int lrm_get_mode(void);

/* ---- leg factories: static_variables.cpp:6-93 ---------------------------------------- */
void lrm_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia,
                     float tibia2femur, float femur2tip, float coxa_angle_deg,
                     float femur_angle_deg, float tibia_angle_deg, float tib_abs_pos,
                     float tib_abs_neg, LrmLegDimensions* out);
void lrm_get_M2_leg(float azimut, LrmLegDimensions* out);      /* static_variables.cpp:69-93 */
void lrm_get_moonbot_leg(float azimut, LrmLegDimensions* out); /* static_variables.cpp:44-67 */

/* rotate_leg_data, one_leg_global.cu:48-60 (host helper; also several_leg.cu:743-760) */
void lrm_rotate_leg_data(const float quat[4], const LrmLegDimensions* leg, LrmLegDimensions* out);

/* ---- host-buffer drop-ins ------------------------------------------------------------
 * Replace apply_kernel<float3,LegDimensions,bool|float3> (cross_compiled.cu:33-79) for the
 * kernels reachability_global_kernel / distance_global_kernel (one_leg_global.cu:149-166):
 * device buffers are allocated and freed inside the call, input is copied H2D, a warm-up
 * launch runs, the kernel alone is timed with events, results are copied D2H.
 * xyz is AoS float3 (12-byte stride), mask is one byte per point (C++ bool), *ms receives
 * the kernel-only milliseconds.  `quat` = NULL means the reference's quatTest {1,0,0,0}. */
int lrm_reach(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
              uint8_t* mask_out, float* ms);
int lrm_dist(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
             float* dxyz_aos_out, uint8_t* valid_out /* may be NULL */, float* ms);
/* one launch producing both outputs (mask is reachability_global's, not distance's bool) */
int lrm_reach_dist(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                   uint8_t* mask_out, float* dxyz_aos_out, float* ms);

/* Host buffers in the reference's on-disk layout: one f32 array per component, as
 * dist_input_t{x,y,z}.bin / out_dist_x{x,y,z}.bin (several_leg.cpp:126-131, :201-219,
 * math_util.cpp:46-89).  Same semantics as lrm_reach / lrm_dist without the AoS detour. */
int lrm_reach_soa(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                  const float* quat, uint8_t* mask_out, float* ms);
int lrm_dist_soa(const float* x, const float* y, const float* z, size_t n, const LrmLegDimensions* leg,
                 const float* quat, float* dx, float* dy, float* dz, uint8_t* valid_out /* may be NULL */,
                 float* ms);

/* ---- CPU path: apply_reach_cpu / apply_dist_cpu, cross_compiled.cu:163-181 -------------
 * Single-threaded host loops over the same per-point code the kernels run (the reference
 * compiles one `__host__ __device__` source twice in the same way).  These are explicit CPU
 * entry points of the reference API, never a fallback for the GPU ones. */
int lrm_reach_cpu(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                  uint8_t* mask_out, double* ms);
int lrm_dist_cpu(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                 float* dxyz_aos_out, uint8_t* valid_out /* may be NULL */, double* ms);

/* ---- "RBDL-equivalent" CPU baseline: apply_RBDL, rbdl_benchmark.cpp:18-111 / RBDL_benchmark.h:5 ------
 * The reference times RBDL's Levenberg-Marquardt position IK on the same targets (bench.cpp:158, 3 repeats,
 * setting_bench.h:7).  RBDL is an external, unpinned dependency that is absent here: this is the same iteration
 * (same chain incl. the /400 scaling, max_steps = 10, <= 5 starts) with closed-form kinematics.  PARITY UNPINNED:
 * a timing baseline only; mask_out[i] = the solver converged (no joint limits, no coxa pitch, as the
 * reference's RBDL model).  *ms = chrono milliseconds of the loop. */
int lrm_rbdl_equiv_cpu(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, uint8_t* mask_out, double* ms);

/* ---- device-resident entry points (launch only: no copy of the clouds, the caller synchronises) ----
 * Pointers are device pointers; coordinates are SoA (one f32 array per component: the same
 * layout the reference keeps on disk, several_leg.cpp:126-131).  `stream` is a hipStream_t
 * (NULL = default stream).  `n` need not be a multiple of anything; 16-byte aligned arrays
 * (hipMalloc / torch allocations are) take the vectorised kernels, anything else a scalar
 * variant with identical results.
 * LRM_MODE_STRICT / LRM_MODE_FAST: nothing but the launch (no allocation, no host synchronisation).
 * LRM_MODE_TOL (distance / fused calls): the first call for a (leg, orientation) compiles the mode's tables on the
 * host (~0.3 ms) and builds the plane table for clouds of >= 2e5 points on the device (< 1 ms on the call's stream, one small
 * read-back; LRM_TOLTAB_HOST=1: the host builder, ~30 ms), and allocates the doubt
 * queues of this (device, stream); a later call with a larger n regrows the queues (hipFree + hipMalloc: a device-wide
 * synchronisation).  lrm_tol_prepare does all of that ahead of time, after which the calls only launch -- graph
 * capture and latency-critical loops call it first.  The table cache holds 64 (leg, orientation) pairs, least
 * recently used out.  lrm_release_workspaces frees every cached device buffer (queues, tables, the host pipeline's
 * buffers and streams, the multi-device communicators); the next call re-creates what it needs. */
int lrm_tol_prepare(const LrmLegDimensions* leg, const float* quat, size_t n_max, void* stream);
/* Milliseconds the most recent plane-table build of this process took (the table of a (leg, orientation) is built by the first
 * call that needs it, or by lrm_tol_prepare, and cached); -1 when none has been built yet.  bench.py reports it as
 * config.table_build_ms next to ms_per_step. */
int lrm_tol_table_build_ms(float* ms_out);
void lrm_release_workspaces(void);
int lrm_reach_dev(const float* x, const float* y, const float* z, size_t n,
                  const LrmLegDimensions* leg, const float* quat, uint8_t* mask, void* stream);
/* as lrm_reach_dev, plus a wave-ballot bit mask: bit (i & 63) of bits[i >> 6]
 * (ceil(n/64) words; either output may be NULL) */
int lrm_reach_bits_dev(const float* x, const float* y, const float* z, size_t n,
                       const LrmLegDimensions* leg, const float* quat, uint8_t* mask,
                       uint64_t* bits, void* stream);
int lrm_dist_dev(const float* x, const float* y, const float* z, size_t n,
                 const LrmLegDimensions* leg, const float* quat, float* dx, float* dy, float* dz,
                 uint8_t* valid /* may be NULL */, void* stream);
int lrm_reach_dist_dev(const float* x, const float* y, const float* z, size_t n,
                       const LrmLegDimensions* leg, const float* quat, uint8_t* mask, float* dx,
                       float* dy, float* dz, void* stream);
/* as lrm_reach_dist_dev, plus the wave-ballot bit mask of the reach mask (the shard payload
 * of the multi-GPU gather); either mask output may be NULL */
int lrm_reach_dist_bits_dev(const float* x, const float* y, const float* z, size_t n,
                            const LrmLegDimensions* leg, const float* quat, uint8_t* mask,
                            uint64_t* bits, float* dx, float* dy, float* dz, void* stream);
/* AoS device variants (what apply_kernel launches on its device copies) */
int lrm_reach_aos_dev(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
                      uint8_t* mask, void* stream);
int lrm_dist_aos_dev(const float* xyz, size_t n, const LrmLegDimensions* leg, const float* quat,
                     float* dxyz, uint8_t* valid /* may be NULL */, void* stream);

/* ---- body x target aggregation ---------------------------------------------------------
 * Replaces reach_mem_kernel + launch_opti_mem_reach_kernel (several_leg.cu:92-192) for all
 * legs in ONE launch: out[l*nb + b] = 1 iff some target t satisfies
 * reachable_rotate_leg(t, body b, quat, legs[l]) (several_leg.cu:48-67); otherwise 0 (the
 * whole output is written).  `legs` are used as given (several_leg.cu:743-760 rotates the
 * limits on the host before the launch: use lrm_rotate_leg_data for that).
 * If all_legs_out != NULL it receives the AND over legs per body (agregateReachability,
 * several_leg.cu:681-697, generalised from 4 to nlegs legs). nlegs <= LRM_MAX_LEGS. */
#define LRM_MAX_LEGS 8
int lrm_reach_any_dev(const float* bx, const float* by, const float* bz, size_t nb,
                      const float* tx, const float* ty, const float* tz, size_t nt,
                      const LrmLegDimensions* legs, size_t nlegs, const float* quat,
                      uint8_t* out_leg_body, uint8_t* all_legs_out, void* stream);
/* host-buffer form of robot_full_struct's pipeline (several_leg.cu:326-877; AoS in, as its
 * Array<float3> arguments); quats is nquat x 4; body_mask_out[b] = 1 iff for SOME orientation
 * EVERY leg (limits rotated per orientation, bodies and targets rotated by the quaternion) has a
 * reachable target.  reference_culls != 0 additionally applies multi_rot_estimator's culls: the
 * one-time spheres (r = 60 collision, r = 400 far body / far target, :413-502) and the
 * per-orientation cylinder pair of eliminateFarAndColliding (:504-559).  reference_culls == 2 applies only the
 * per-orientation pair: for callers that shard the bodies over several GPUs and evaluate the one-time culls
 * themselves (the far-target cull depends on ALL surviving bodies; lrm_amd/shard.py).  *ms = kernel time. */
int lrm_positionability(const float* bodies_aos, size_t nb, const float* targets_aos, size_t nt,
                        const LrmLegDimensions* legs, size_t nlegs, const float* quats,
                        size_t nquat, int reference_culls, uint8_t* body_mask_out, float* ms);

/* lrm_positionability's orientation sweep on clouds that already live on the device (SoA float32), device-resident
 * masks in and out: no copies of the clouds, no reordering (feed Morton-ordered clouds).  reference_culls: 0 none, 2 the
 * per-orientation cylinder culls (the caller has applied multi_rot_estimator's one-time culls, several_leg.cu:413-502,
 * as the sharded drivers do).  active_in (device, may be NULL = every body): bodies with 0 are not tested;
 * accepted_out[nb] (device).  Null stream; returns when the device has finished; *ms = the sweep's kernel time. */
int lrm_positionability_dev(const float* bx, const float* by, const float* bz, size_t nb, const float* tx, const float* ty,
                            const float* tz, size_t nt, const LrmLegDimensions* legs, size_t nlegs, const float* quats,
                            size_t nquat, int reference_culls, const uint8_t* active_in, uint8_t* accepted_out, float* ms);
/* Morton (Z-curve) order of a host cloud: order_out[k] = index of the k-th point along the curve.
 * The pair kernels (lrm_reach_any_dev, lrm_any_in_*_dev) skip whole 1024-target tiles by bounding
 * box; feeding them clouds (and centres / bodies) in this order makes the boxes compact in any
 * orientation.  Results do not depend on the order.  lrm_positionability does this itself. */
int lrm_morton_order(const float* xyz_aos, size_t n, uint64_t* order_out);

/* in_sphere / in_cylinder any-reductions: launch_optimized_mem_in_sphere /
 * launch_optimized_mem_in_cylinder (collision.cu:68-98, :148-168):
 * out[c] = 1 iff some target lies in the sphere / cylinder centred on centre c. */
int lrm_any_in_sphere_dev(const float* cx, const float* cy, const float* cz, size_t nc,
                          const float* tx, const float* ty, const float* tz, size_t nt,
                          float radius, uint8_t* out, void* stream);
int lrm_any_in_cylinder_dev(const float* cx, const float* cy, const float* cz, size_t nc,
                            const float* tx, const float* ty, const float* tz, size_t nt,
                            float radius, float plus_z, float minus_z, uint8_t* out, void* stream);

/* ---- octree-culled positionability: apply_oct, several_leg_octree.cu:391-488 ----------------
 * Breadth-first refinement of the body-position box: per level every child box is tested against
 * every foothold (and the orientation samples for small boxes) with distance_global for each
 * mounted leg (validity_child, several_leg_octree.cu:19-151); valid leaves' centres are returned
 * in depth-first child order (extractValidAsArray, octree_util.cu:128-180).  The compile-time knobs
 * of settings.h:15-46 are a struct here; lrm_octree_default_settings() fills in the committed
 * values (root box +-5000 mm, MINBOXSIZE 100, 3x3x3 orientation samples, 4 legs mounted every
 * pi/4, LegNumberForStab 4, MAX_DEPTH 1).  Level-synchronous flat arrays replace the reference's
 * device-side cudaMalloc and kernel-launching kernels. */
typedef struct LrmOctreeSettings {
    float box_center[3];        /* settings.h:24 BoxCenter        */
    float box_size[3];          /* settings.h:26 BoxSize (half)   */
    float min_box;              /* settings.h:17 MINBOXSIZE       */
    float enable_rot_below;     /* settings.h:33 EnableRotBelow   */
    float convex_radius;        /* settings.h:34 convexRadius     */
    int32_t angle_sample[3];    /* settings.h:35 AngleSample      */
    float angle_minmax[6];      /* settings.h:38 AngleMinMax      */
    int32_t leg_count;          /* settings.h:41 LegCount         */
    float leg_mount[8];         /* settings.h:42 LegMount         */
    int32_t leg_number_for_stab;/* settings.h:46 LegNumberForStab */
    int32_t max_depth;          /* settings.h:15 MAX_DEPTH        */
} LrmOctreeSettings;
void lrm_octree_default_settings(LrmOctreeSettings* out);
/* footholds: AoS float3 (Array<float3> input of apply_oct); centers_out: room for `capacity` float3;
 * *n_out = number of valid leaves (if > capacity the call fails with LRM_EINVAL and *n_out tells the
 * size to retry with); settings = NULL -> defaults; *ms = kernel time.
 * From 3e5 footholds on (LRM_MODE_FAST) the work items' decisions come from the plane tables of the (leg, orientation) pairs, built
 * on the device on the first call that meets a pair (~0.4 ms each) and kept for the process; the tree is the same. */
int lrm_apply_oct(const float* footholds_aos, size_t n, const LrmLegDimensions* dim,
                  const LrmOctreeSettings* settings, float* centers_out, size_t capacity, size_t* n_out,
                  float* ms);
/* The same tree on several GPUs, one process each (the reference is single-device).  `exchange(flags, n, user)` must
 * replace flags[0..n) by their element-wise BITWISE OR over all ranks and return 0 (non-zero: the call fails); it is
 * called by every rank once with n = 1 before the first level and once per level.  A rank that fails locally still
 * enters the exchange its peers wait in with 0xffffffff in every word, and a rank that reads 0xffffffff fails too.
 * lrm_apply_oct_sharded: every rank holds all footholds, the children of a level are dealt round-robin to the ranks.
 * lrm_apply_oct_partitioned(_dev): every rank holds ITS part of the footholds (any disjoint split of the cloud; a
 *   spatial one keeps the work local) and evaluates every child against it -- a child's flags are ORs over
 *   footholds, so the OR over the ranks is exact: BASELINE config 5 without a replica of the 1e8-point cloud per GPU.
 * Every rank returns all valid leaves. */
typedef int (*LrmOctExchange)(uint32_t* flags, size_t n, void* user);
int lrm_apply_oct_sharded(const float* footholds_aos, size_t n, const LrmLegDimensions* dim,
                          const LrmOctreeSettings* settings, float* centers_out, size_t capacity, size_t* n_out,
                          float* ms, int rank, int world, LrmOctExchange exchange, void* user);
int lrm_apply_oct_partitioned(const float* local_footholds_aos, size_t n_local, const LrmLegDimensions* dim,
                              const LrmOctreeSettings* settings, float* centers_out, size_t capacity, size_t* n_out,
                              float* ms, LrmOctExchange exchange, void* user);
int lrm_apply_oct_partitioned_dev(const float* fx, const float* fy, const float* fz, size_t n_local, const LrmLegDimensions* dim,
                                  const LrmOctreeSettings* settings, float* centers_out, size_t capacity, size_t* n_out,
                                  float* ms, LrmOctExchange exchange, void* user);
/* The same with the footholds already on the device, one array per component (the layout of the other *_dev entry points;
 * the reference has no such call: apply_oct uploads its Array<float3> every time, several_leg_octree.cu:408-414).  The arrays
 * are read only (a sorted copy is made).  rank / world / exchange as lrm_apply_oct_sharded (0, 1, NULL, NULL on one GPU). */
int lrm_apply_oct_dev(const float* fx, const float* fy, const float* fz, size_t n, const LrmLegDimensions* dim,
                      const LrmOctreeSettings* settings, float* centers_out, size_t capacity, size_t* n_out, float* ms,
                      int rank, int world, LrmOctExchange exchange, void* user);
const char* lrm_octree_last_error(void);
/* Trace of the octree calls of this thread (tests): after lrm_dbg_oct_trace(1) every evaluated child of every level is
 * recorded as 12 floats {c[3], h[3], parent h[3], flag bits the kernel returned (1 reach, 2 leaf, 4 edge),
 * parent_valid + 2 * rotations + 4 * skipped, depth}; lrm_dbg_oct_trace_read copies them (out may be NULL to ask for
 * the count); lrm_dbg_oct_trace(0) stops and clears. */
int lrm_dbg_oct_trace(int enable);
int lrm_dbg_oct_trace_read(float* out, size_t capacity_records, size_t* n_out);

/* ---- several GPUs behind the apply_kernel boundary (one process, one host thread) ------------------------------
 * New capability: the reference runs on device 0 only (several_leg.cu:800; apply_kernel cross_compiled.cu:33-79).
 * lrm_shard_bounds: owner `rank` of `world` gets items [lo, hi) of n, boundaries multiples of `align` (64 for point
 *   clouds: no 64-point ballot word straddles two owners); ceil(n / world) rounded up to `align` per owner, the last
 *   owners may be short or empty.  The Python side (lrm_amd.shard.shard_bounds) uses the same arithmetic.
 * lrm_reach_dist_multi: lrm_reach_dist with the cloud cut into those shards over `ndev` devices (`devices`: their
 *   ordinals, NULL = 0 .. ndev-1): per device one stream, its slice of the input, the fused kernels of the current
 *   mode, the reach bytes packed into ballot words; the words are all-gathered with RCCL (ncclCommInitAll once per
 *   device set + one grouped ncclAllGather per call; librccl.so is opened on first use with ndev > 1), so that every
 *   device holds the bit-packed mask of the whole cloud (BASELINE config 4's exchange step).  mask_out[n] /
 *   dxyz_out[3 n] as lrm_reach_dist; bits_out: NULL or ceil(n / 64) words, the gathered mask as devices[0] holds it;
 *   ms_per_dev: NULL or [ndev] kernel milliseconds per device.  lrm_multi_release frees the cached communicators. */
int lrm_shard_bounds(size_t n, int world, int rank, size_t align, size_t* lo_out, size_t* hi_out);
int lrm_reach_dist_multi(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat, int ndev,
                         const int* devices, uint8_t* mask_out, float* dxyz_out, uint64_t* bits_out, float* ms_per_dev);
void lrm_multi_release(void);

/* ---- diagnostics -------------------------------------------------------------------------
 * The glibc-exact atan2f / sincosf of the strict kernels (csrc/lrm_exact_math.h) applied to
 * arrays: at2[i] = atan2f(a[i], b[i]); (sn[i], cs[i]) = sincosf(a[i]).  Host build and device
 * build; tests compare both with the platform libm. */
int lrm_dbg_exact_math_host(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs);
int lrm_dbg_exact_math_dev(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs,
                           void* stream);
/* The device's correctly rounded square root (csrc/lrm_exact_math.h, lrm_sqrtf) against the
 * compiler's IEEE sqrtf on ALL 2^32 float bit patterns: writes the number of patterns whose results
 * differ bitwise (nan payloads included) and the first such pattern.  Synchronous. */
int lrm_dbg_sqrt_check_dev(uint64_t* mismatches_out, uint32_t* first_bad_out);

/* The filtered (LRM_MODE_FAST) per-point evaluation run on the host WITHOUT its strict
 * fallback, plus the per-point "uncertain" flags that would trigger the fallback.  Any output
 * pointer may be NULL.  Fails with LRM_EINVAL for a leg the filter does not support. */
int lrm_dbg_fast_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                      uint8_t* mask_out, uint8_t* mask_uncertain_out, float* dxyz_aos_out,
                      uint8_t* valid_out, uint8_t* dist_uncertain_out);
/* The contract-tolerance evaluation (LRM_MODE_TOL, csrc/lrm_point_tol.h) on the host WITHOUT the bit-exact
 * re-evaluation of its doubtful points: mask_out = reach / validity flag, doubt_out = LRM_TD_* bits (0: the
 * outputs are final).  Fails with LRM_EINVAL for a leg the mode does not support. */
int lrm_dbg_tol_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                     uint8_t* mask_out, float* dxyz_aos_out, uint32_t* doubt_out);
/* As lrm_dbg_tol_host with the plane table with deferred decisions (csrc/lrm_toltab.cpp) in place of the full plane
 * evaluation; doubt bit 0x100 = a cell without an answer.  stats_out[5] (or NULL): rows, validity rows, refined cells, bytes,
 * points whose second yaw candidate had to be evaluated (its lower bound did not exclude it). */
int lrm_dbg_toltab_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                        uint8_t* mask_out, float* dxyz_out, uint32_t* doubt_out, uint32_t* stats_out);
/* The bit-exact table-guided evaluation (csrc/lrm_point_xtab.h: decisions from the plane table, values in the reference's
 * operation order) on the host, WITHOUT the re-evaluation of its doubtful points: every point with doubt 0 carries the mask and
 * the vector of lrm_reach_cpu / lrm_dist_cpu bit for bit (tests/test_xtab_cpu.py).  stats_out[2] (or NULL): points whose
 * second value chain ran, table bytes. */
int lrm_dbg_xtab_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                      uint8_t* mask_out, float* dxyz_out, uint32_t* doubt_out, uint32_t* stats_out);
/* LRM_MODE_TOL_REL's two steps on the host: the tolerance evaluation with the plane table, then -- for every point without doubt --
 * the strict replay of the winner's value chain from the decisions the first step took (csrc/lrm_point_xtab.h: lrm_xtab_replay):
 * those vectors equal lrm_dist_cpu bit for bit (tests/test_xtab_cpu.py).  A point in doubt keeps the tolerance vector. */
int lrm_dbg_replay_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                        uint8_t* mask_out, float* dxyz_out, uint32_t* doubt_out);
/* The plane table of (leg, quat) as the HOST builder (device = 0: csrc/lrm_toltab.cpp) or the DEVICE builder (device = 1:
 * csrc/lrm_toltab_dev.hip, on the current device) makes it: its bytes into out[cap] (when they fit; out may be NULL), its size, the
 * build's milliseconds.  The two must agree byte for byte (tests/test_gpu_toltab.py). */
int lrm_dbg_toltab_build(const LrmLegDimensions* leg, const float* quat, int device, uint8_t* out, size_t cap, size_t* size_out,
                         float* ms_out);
/* The plane table's lower bound of the in-plane distance at n plane points xz[2 n] (abscissa - coxa_length, z), next to the
 * full plane evaluation there: distance sqrt(du^2 + dz^2), validity, doubt bits.  The bound must not exceed the distance
 * of an invalid point and must be 0 at a valid one (tests/test_tol_cpu.py). */
int lrm_dbg_toltab_bounds(const float* xz, size_t n, const LrmLegDimensions* leg, const float* quat, float* lb_out,
                          float* dist_out, uint8_t* valid_out, uint32_t* doubt_out);
/* Counting build only (csrc: -DLRM_PAIR_COUNT, tools/c3_evidence.py): what the wave-per-body pair kernel evaluated since the
 * last call: out[0] full (leg, target) evaluations, [1] leg bounding-sphere tests, [2] footholds inside a body's reach sphere,
 * [3] footholds loaded.  LRM_EINVAL in an ordinary build. */
int lrm_dbg_pair_counts(uint64_t out[4]);
/* After a distance / fused call on device buffers in LRM_MODE_TOL: the points of that call, how many of them its main
 * kernel queued for the bit-exact fix-up launch, and how many workgroups overflowed their queue segment (all their
 * points are re-evaluated).  Synchronises that device. */
int lrm_dbg_tol_queue_counts(uint64_t* n_points, uint64_t* n_queued, uint64_t* n_overflowed);
/* 1 if (leg, quat) is eligible for LRM_MODE_TOL, else 0 */
int lrm_dbg_tol_ok(const LrmLegDimensions* leg, const float* quat);
/* The per-leg bounding sphere the pair kernels use to skip batches of footholds:
 * out4 = {cx, cy, cz (relative to the body position), squared radius}.  Tests check that every
 * pair the strict reachable_rotate_leg accepts lies inside. */
int lrm_dbg_pair_sphere(const LrmLegDimensions* leg, const float* quat, float* out4);

/* The reach mask the fused filtered kernel derives from its distance evaluation
 * (lrm_reach_from_dist, csrc/lrm_point_fast.h), WITHOUT the strict fallback, and the per-point
 * doubt flag that would trigger it. */
int lrm_dbg_fused_reach_host(const float* xyz_aos, size_t n, const LrmLegDimensions* leg, const float* quat,
                             uint8_t* mask_out, uint8_t* doubt_out);

#ifdef __cplusplus
}
#endif
#endif /* LRM_H */

// lrm_launch.h -- kernel launch functions (lrm_kernels.hip) used by the C ABI (lrm_capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include "lrm_types.h"

hipError_t lrm_launch_warmup(size_t n, hipStream_t st);
hipError_t lrm_launch_reach_soa(const float* x, const float* y, const float* z, size_t n,
                                const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, bool fast, hipStream_t st);
// op: 1 = distance (mask = distance's validity byte, may be null), 2 = reach mask + distance
hipError_t lrm_launch_dist_soa(int op, const float* x, const float* y, const float* z, size_t n,
                               const LrmCompiledLeg& L, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                               float* dz, bool fast, hipStream_t st);
// LRM_MODE_TOL (lrm_tol_kernels.hip): the tolerance kernel + the fix-up of its doubtful points, two launches on
// `st`.  workspace: lrm_tol_queue_words(n) uint32 of device memory owned by the caller for the duration of both
// launches (contents are rewritten by every call; no initialisation needed).
#ifndef LRM_TOL_SEG_CAP
#define LRM_TOL_SEG_CAP 128 // doubt slots per workgroup of the main kernel (768 points): 17 %.  32 overflowed on the reference's planar bench grid (7 % in doubt: it contains the coxa axis and the symmetry plane)
#endif
#define LRM_TOL_SEG_CAP_WORDS LRM_TOL_SEG_CAP
#ifndef LRM_TOL_TAB_SEG_CAP
#define LRM_TOL_TAB_SEG_CAP 256 // doubt slots per workgroup of the table kernel (1536 points): 17 %
#endif
// flags of the tolerance-mode launches (bits 0 and 1 are LRM_TOL_SELFTEST's)
#define LRM_TOLF_SHORT 4u     // LRM_MODE_TOL_REL: every vector shorter than LRM_TOL_REL_MM is queued for the bit-exact fix-up
#ifndef LRM_TOL_REL_MM
#define LRM_TOL_REL_MM 19.0f  // the literal bound |d - d_ref| <= 1e-5 |d_ref| is asserted from 16 mm on (17 mm until the third campaign of round 4 found 7.6e-6 at 17.55 mm: the threshold and the band count below went up by an eighth, 0.9 us per 1e7 points)
#endif
#ifndef LRM_TOL_REL_BANDS
#define LRM_TOL_REL_BANDS 2250.0f // ... and from 2250 decision bands on (34 mm at |p|_1 + body = 1 m): the error grows with the coordinates
#endif
size_t lrm_tol_queue_words(size_t n);
hipError_t lrm_launch_dist_tol_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmTolLeg& TL, uint8_t* mask,
                                   float* dxyz, uint32_t* workspace, uint32_t flags, hipStream_t st);
hipError_t lrm_launch_dist_tol(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz,
                               uint32_t* workspace, uint32_t flags, hipStream_t st);
// Table variant (dist_tab_kernel + the same fix-up): tab_dev = device copy of lrm_build_tol_tab's table for TL.
size_t lrm_tol_tab_queue_words(size_t n);
size_t lrm_tol_tab_segments(size_t n, bool rel); // the workspace starts with one count per segment (rel: LRM_MODE_TOL_REL's grid)
struct LrmXtabLeg; // lrm_point_xtab.h
hipError_t lrm_launch_dist_tab(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L,
                               const LrmTolLeg& TL, const LrmXtabLeg& X, const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy,
                               float* dz, uint32_t* workspace, uint32_t flags, hipStream_t st);
hipError_t lrm_launch_dist_tab_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmTolLeg& TL, const LrmXtabLeg& X, const uint8_t* tab_dev,
                                   uint8_t* mask, float* dxyz, uint32_t* workspace, uint32_t flags, hipStream_t st);
// LRM_MODE_FAST through the plane table (dist_xtab_kernel, lrm_point_xtab.h: decisions from the table, values in the reference's
// order, bit-identical to LRM_MODE_STRICT) + tol_fixup_kernel for its doubtful points.  Workspace: lrm_tol_tab_queue_words(n).
hipError_t lrm_launch_dist_xtab(int op, const float* x, const float* y, const float* z, size_t n, const LrmCompiledLeg& L, const LrmXtabLeg& X,
                                const uint8_t* tab_dev, uint8_t* mask, uint64_t* bits, float* dx, float* dy, float* dz, uint32_t* workspace, hipStream_t st);
hipError_t lrm_launch_dist_xtab_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, const LrmXtabLeg& X, const uint8_t* tab_dev,
                                    uint8_t* mask, float* dxyz, uint32_t* workspace, hipStream_t st);
hipError_t lrm_launch_reach_aos(const float* xyz, size_t n, const LrmCompiledLeg& L, uint8_t* mask, bool fast,
                                hipStream_t st);
hipError_t lrm_launch_dist_aos(int op, const float* xyz, size_t n, const LrmCompiledLeg& L, uint8_t* mask,
                               float* dxyz, bool fast, hipStream_t st);
hipError_t lrm_launch_reach_any(const float* bx, const float* by, const float* bz, size_t nb, const float* tx,
                                const float* ty, const float* tz, size_t nt, const LrmCompiledLeg* legs_dev,
                                int nlegs, float* tile_boxes /* 17 x ntiles x 6 floats of workspace, or null */,
                                bool boxes_ready /* the workspace already holds this cloud's boxes */,
                                const uint8_t* body_active /* null = all */, uint8_t* out_leg_body,
                                uint8_t* all_legs_out, bool fast, hipStream_t st);
hipError_t lrm_launch_any_in_shape(int shape, const float* cx, const float* cy, const float* cz, size_t nc,
                                   const float* tx, const float* ty, const float* tz, size_t nt, float radius,
                                   float plus_z, float minus_z, float* tile_boxes /* workspace or null */,
                                   bool boxes_ready, uint8_t* out, hipStream_t st);
hipError_t lrm_launch_sqrt_check(unsigned long long* counters_dev /* [2]: mismatches, first bad pattern + 1 */,
                                 hipStream_t st);
hipError_t lrm_launch_exact_math(const float* a, const float* b, size_t n, float* at2, float* sn, float* cs,
                                 hipStream_t st);
hipError_t lrm_launch_rotate_soa(const float* sx, const float* sy, const float* sz, size_t n, const LrmCompiledLeg* rot_dev,
                                 float* dx, float* dy, float* dz, hipStream_t st);
hipError_t lrm_launch_sweep_update(const uint8_t* all_legs, const uint8_t* cyl_validate, const uint8_t* cyl_eliminate,
                                   int use_culls, size_t nb, uint8_t* active, uint8_t* accepted, hipStream_t st);
// evaluation counters of reach_any_wave_kernel in a -DLRM_PAIR_COUNT build (read and reset); hipErrorNotSupported otherwise
hipError_t lrm_pair_counts(unsigned long long out[4]);

// The plane table built on the device (lrm_toltab_dev.hip): 0 ok (*tab_dev_out = a fresh hipMalloc-ed table, the caller's), 1 this
// leg has no table, 2 the device builder does not take this leg (use lrm_build_tol_tab), < 0 a negated hipError_t.
int lrm_build_tol_tab_dev(const LrmTolLeg& L, hipStream_t st, uint8_t** tab_dev_out, size_t* bytes_out, float* ms_out);
void lrm_toltab_dev_release();

// lrm_compile.h -- host side: (LegDimensions, quaternion) -> LrmCompiledLeg.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "lrm_types.h"

// apply_leg_rotation != 0: the tibia limits are first rotated by the pitch of the body
// orientation seen from the leg (rotate_leg_data, one_leg_global.cu:48-60), as
// reachability_global / distance_global do.  == 0: the leg is used as given
// (reach_mem_kernel receives legs the host already rotated, several_leg.cu:743-760).
void lrm_compile_leg(const LrmLegDimensions& leg, const float quat[4], int apply_leg_rotation,
                     LrmCompiledLeg* out);

// host helpers shared with the C ABI
void lrm_host_rotate_leg_data(const float quat[4], const LrmLegDimensions& leg, LrmLegDimensions* out);
void lrm_host_leg_factory(float azimut, float body2coxa, float coxa_pitch_deg, float coxa2tibia,
                          float tibia2femur, float femur2tip, float coxa_angle_deg,
                          float femur_angle_deg, float tibia_angle_deg, float tib_abs_pos,
                          float tib_abs_neg, LrmLegDimensions* out);

// Tolerance mode (LRM_MODE_TOL, lrm_point_tol.h): the block its per-point code reads, derived from an
// already compiled leg.  out->tol_ok == 0 when the leg's geometry does not fit the mode's assumptions
// (the valid part of some circle is not one arc, a clamp-validity boundary grazes instead of crossing,
// the leg is not eligible for the filters at all): callers then use LRM_MODE_FAST for that leg.
void lrm_compile_tol(const LrmCompiledLeg& L, LrmTolLeg* out);

// The plane table with deferred decisions (lrm_types.h: LrmTolTabHeader | uint16 coarse[LRM_TT_N^2] | uint16 fine[16 (n_fine + 1)]).
// false: the leg needs more distinct rows than a cell code can name -- the caller uses the kernels without a table.
bool lrm_build_tol_tab(const LrmTolLeg& L, std::vector<uint8_t>* out);

// order[k] = index of the k-th point of the AoS cloud along a Morton (Z) curve over its bounding box (lrm_capi.cpp)
void lrm_host_morton_order(const float* xyz_aos, size_t n, std::vector<size_t>* order);

// ---- shared pieces of the two table builders (lrm_toltab.cpp; lrm_toltab_dev.hip uses them on the host side of its build) ----
struct LrmTbInput;
bool lrm_tb_number_rows(const LrmTbInput& in, uint64_t used_rows, uint32_t used_vrows, uint8_t* row_num, uint8_t* vrow_num, LrmTolTabHeader* hd);
void lrm_tb_layout(const uint32_t n_fine[2], LrmTolTabHeader* hd, size_t* n_cells);

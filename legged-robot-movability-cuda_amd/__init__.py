"""MI355X-native reach / distance / positionability path (drop-in for the hot path of
2lian/Legged-Robot-Movability-Cuda).  The product is csrc/ -> liblrm.so (HIP kernels + the
C ABI of include/lrm.h); this package is the Python binding used by tests and bench.py.
The C++ mirror of the reference's host interface is include/lrm_compat.hpp.
"""
from ._capi import (  # noqa: F401
    LIB_PATH, LrmError, MODE_FAST, MODE_STRICT, MODE_TOL, MODE_TOL_REL, build, lib, load, leg_factory, get_M2_leg,
    get_moonbot_leg, rotate_leg_data, apply_reach, apply_dist, apply_reach_dist,
    apply_reach_cpu, apply_dist_cpu, apply_rbdl_equiv, positionability, set_mode, get_mode, device_count,
    exported_symbols, declared_symbols, dbg_fast_host, dbg_fused_reach_host, dbg_tol_host, dbg_tol_ok, dbg_toltab_host, dbg_toltab_bounds, dbg_xtab_host, dbg_replay_host, dbg_toltab_build, shard_bounds as c_shard_bounds, apply_reach_dist_multi, tol_prepare, last_table_build_ms, release_workspaces, dbg_tol_queue_counts, dbg_pair_counts, dbg_oct_trace, dbg_oct_trace_read, dbg_pair_sphere, apply_oct, apply_oct_sharded, apply_oct_partitioned, octree_default_settings,
    OctreeSettings, morton_order,
)
from . import device  # noqa: F401
from . import shard  # noqa: F401

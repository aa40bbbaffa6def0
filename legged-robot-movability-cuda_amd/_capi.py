"""ctypes binding of liblrm.so (include/lrm.h).  No compute happens in Python, and there
is no fallback: if the shared library is missing, load() raises."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LRM_LIB_PATH: load another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("LRM_LIB_PATH") or os.path.join(_HERE, "liblrm.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lrm.h")
MODE_STRICT, MODE_FAST, MODE_TOL, MODE_TOL_REL = 0, 1, 2, 3
_lib = None


class LrmError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/ into liblrm.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def declared_symbols():
    """Every function name declared in include/lrm.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lrm_[a-z0-9_]+)\s*\(", text)))


def exported_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], check=True,
                         capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LrmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the HIP path)")
    try:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7; importing it
        # first makes liblrm.so bind to that copy, so torch's device pointers and streams are
        # valid in our launches.  Without torch the library binds to /opt/rocm's runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, fp = C.c_void_p, C.c_size_t, C.c_float
    L.lrm_version.restype = C.c_char_p
    L.lrm_last_error.restype = C.c_char_p
    sig = {
        "lrm_device_count": [], "lrm_set_device": [C.c_int], "lrm_set_mode": [C.c_int], "lrm_get_mode": [],
        "lrm_reach": [vp, sz, vp, vp, vp, vp],
        "lrm_dist": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_reach_dist": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_reach_soa": [vp, vp, vp, sz, vp, vp, vp, vp],
        "lrm_dist_soa": [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp],
        "lrm_reach_cpu": [vp, sz, vp, vp, vp, vp],
        "lrm_dist_cpu": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_rbdl_equiv_cpu": [vp, sz, vp, vp, vp],
        "lrm_reach_dev": [vp, vp, vp, sz, vp, vp, vp, vp],
        "lrm_reach_bits_dev": [vp, vp, vp, sz, vp, vp, vp, vp, vp],
        "lrm_dist_dev": [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp],
        "lrm_reach_dist_dev": [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp],
        "lrm_reach_dist_bits_dev": [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp],
        "lrm_reach_aos_dev": [vp, sz, vp, vp, vp, vp],
        "lrm_dist_aos_dev": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_reach_any_dev": [vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp],
        "lrm_positionability": [vp, sz, vp, sz, vp, sz, vp, sz, C.c_int, vp, vp],
        "lrm_morton_order": [vp, sz, vp],
        "lrm_dbg_fast_host": [vp, sz, vp, vp, vp, vp, vp, vp, vp],
        "lrm_dbg_fused_reach_host": [vp, sz, vp, vp, vp, vp],
        "lrm_dbg_tol_host": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_dbg_tol_ok": [vp, vp],
        "lrm_dbg_tol_queue_counts": [vp, vp, vp],
        "lrm_dbg_toltab_host": [vp, sz, vp, vp, vp, vp, vp, vp],
        "lrm_dbg_toltab_bounds": [vp, sz, vp, vp, vp, vp, vp, vp],
        "lrm_dbg_xtab_host": [vp, sz, vp, vp, vp, vp, vp, vp],
        "lrm_dbg_replay_host": [vp, sz, vp, vp, vp, vp, vp],
        "lrm_dbg_toltab_build": [vp, vp, C.c_int, vp, sz, vp, vp],
        "lrm_shard_bounds": [sz, C.c_int, C.c_int, sz, vp, vp],
        "lrm_dbg_pair_counts": [vp],
        "lrm_tol_prepare": [vp, vp, sz, vp],
        "lrm_tol_table_build_ms": [vp],
        "lrm_positionability_dev": [vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, sz, C.c_int, vp, vp, vp],
        "lrm_dbg_oct_trace": [C.c_int],
        "lrm_dbg_oct_trace_read": [vp, sz, vp],
        "lrm_reach_dist_multi": [vp, sz, vp, vp, C.c_int, vp, vp, vp, vp, vp],
        "lrm_dbg_pair_sphere": [vp, vp, vp],
        "lrm_dbg_exact_math_host": [vp, vp, sz, vp, vp, vp],
        "lrm_dbg_exact_math_dev": [vp, vp, sz, vp, vp, vp, vp],
        "lrm_dbg_sqrt_check_dev": [vp, vp],
        "lrm_any_in_sphere_dev": [vp, vp, vp, sz, vp, vp, vp, sz, fp, vp, vp],
        "lrm_any_in_cylinder_dev": [vp, vp, vp, sz, vp, vp, vp, sz, fp, fp, fp, vp, vp],
    }
    for name, argtypes in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if name.startswith("lrm_dbg_"):  # an older library variant in an A/B run (LRM_LIB_PATH): diagnostics may be missing
                continue
            raise
        fn.argtypes = argtypes
        fn.restype = C.c_int
    L.lrm_leg_factory.argtypes = [fp] * 11 + [vp]
    L.lrm_leg_factory.restype = None
    for name in ("lrm_get_M2_leg", "lrm_get_moonbot_leg"):
        getattr(L, name).argtypes = [fp, vp]
        getattr(L, name).restype = None
    L.lrm_octree_default_settings.argtypes = [vp]
    L.lrm_octree_default_settings.restype = None
    L.lrm_apply_oct.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp]
    L.lrm_apply_oct.restype = C.c_int
    L.lrm_octree_last_error.restype = C.c_char_p
    L.lrm_apply_oct_sharded.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lrm_apply_oct_sharded.restype = C.c_int
    L.lrm_apply_oct_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lrm_apply_oct_dev.restype = C.c_int
    L.lrm_apply_oct_partitioned.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp]
    L.lrm_apply_oct_partitioned.restype = C.c_int
    L.lrm_apply_oct_partitioned_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, vp]
    L.lrm_apply_oct_partitioned_dev.restype = C.c_int
    L.lrm_rotate_leg_data.argtypes = [vp, vp, vp]
    L.lrm_rotate_leg_data.restype = None
    L.lrm_multi_release.argtypes = []
    L.lrm_multi_release.restype = None
    L.lrm_release_workspaces.argtypes = []
    L.lrm_release_workspaces.restype = None
    _lib = L
    return L


def lib():
    return load()


def check(rc):
    if rc != 0:
        raise LrmError(f"liblrm error {rc}: {load().lrm_last_error().decode()}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def _quat(q):
    return None if q is None else _f32(q, (4,))


def device_count():
    return load().lrm_device_count()


def set_mode(mode):
    check(load().lrm_set_mode(mode))


def get_mode():
    return load().lrm_get_mode()


# ---- leg factories (static_variables.cpp:6-93) -------------------------------------------
def leg_factory(azimut, body2coxa, coxa_pitch_deg, coxa2tibia, tibia2femur, femur2tip,
                coxa_angle_deg, femur_angle_deg, tibia_angle_deg, tib_abs_pos, tib_abs_neg):
    out = np.zeros(14, np.float32)
    load().lrm_leg_factory(azimut, body2coxa, coxa_pitch_deg, coxa2tibia, tibia2femur, femur2tip,
                           coxa_angle_deg, femur_angle_deg, tibia_angle_deg, tib_abs_pos, tib_abs_neg,
                           _ptr(out))
    return out


def get_M2_leg(azimut=0.0):
    out = np.zeros(14, np.float32)
    load().lrm_get_M2_leg(azimut, _ptr(out))
    return out


def get_moonbot_leg(azimut=0.0):
    out = np.zeros(14, np.float32)
    load().lrm_get_moonbot_leg(azimut, _ptr(out))
    return out


def rotate_leg_data(quat, leg):
    out = np.zeros(14, np.float32)
    load().lrm_rotate_leg_data(_ptr(_f32(quat, (4,))), _ptr(_f32(leg, (14,))), _ptr(out))
    return out


# ---- host-buffer drop-ins (apply_kernel, cross_compiled.cu:33-79) -------------------------
def apply_reach(xyz, leg, quat=None):
    """-> (mask uint8[n], kernel milliseconds); GPU."""
    xyz = _f32(xyz, (-1, 3))
    mask = np.zeros(len(xyz), np.uint8)
    ms = C.c_float(0)
    check(load().lrm_reach(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask),
                           C.addressof(ms)))
    return mask, ms.value


def apply_dist(xyz, leg, quat=None):
    """-> (distance float32[n,3], validity uint8[n], kernel milliseconds); GPU."""
    xyz = _f32(xyz, (-1, 3))
    d = np.zeros_like(xyz)
    v = np.zeros(len(xyz), np.uint8)
    ms = C.c_float(0)
    check(load().lrm_dist(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(d), _ptr(v),
                          C.addressof(ms)))
    return d, v, ms.value


def apply_reach_dist(xyz, leg, quat=None):
    xyz = _f32(xyz, (-1, 3))
    d = np.zeros_like(xyz)
    m = np.zeros(len(xyz), np.uint8)
    ms = C.c_float(0)
    check(load().lrm_reach_dist(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(m), _ptr(d),
                                C.addressof(ms)))
    return m, d, ms.value


def tol_prepare(leg, quat=None, n_max=0, stream=None):
    """lrm_tol_prepare: LRM_MODE_TOL's tables and queues for (leg, orientation) and clouds of up to n_max points, ahead of time"""
    check(load().lrm_tol_prepare(_ptr(_f32(leg, (14,))), _ptr(_quat(quat)), n_max, stream))


def last_table_build_ms():
    """milliseconds the most recent plane-table build took (-1.0: none yet)"""
    ms = C.c_float(0)
    check(load().lrm_tol_table_build_ms(C.byref(ms)))
    return float(ms.value)


def release_workspaces():
    load().lrm_release_workspaces()


def shard_bounds(n, world, rank, align=64):
    """lrm_shard_bounds: the C ABI's shard arithmetic (equal to lrm_amd.shard.shard_bounds)"""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    check(load().lrm_shard_bounds(n, world, rank, align, C.addressof(lo), C.addressof(hi)))
    return int(lo.value), int(hi.value)


def apply_reach_dist_multi(xyz, leg, quat=None, ndev=1, devices=None, want_bits=True):
    """lrm_reach_dist_multi: the fused kernels over ndev devices of this process + RCCL gather of the bit words
    -> (mask, vectors, gathered words or None, kernel ms per device)"""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    d = np.zeros_like(xyz)
    m = np.zeros(n, np.uint8)
    bits = np.zeros((n + 63) // 64, np.uint64) if want_bits else None
    ms = np.zeros(ndev, np.float32)
    devs = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
    check(load().lrm_reach_dist_multi(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), ndev,
                                      None if devs is None else _ptr(devs), _ptr(m), _ptr(d),
                                      None if bits is None else _ptr(bits), _ptr(ms)))
    return m, d, bits, ms


# ---- CPU entry points (apply_reach_cpu / apply_dist_cpu, cross_compiled.cu:163-181) --------
def apply_reach_cpu(xyz, leg, quat=None):
    xyz = _f32(xyz, (-1, 3))
    mask = np.zeros(len(xyz), np.uint8)
    ms = C.c_double(0)
    check(load().lrm_reach_cpu(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask),
                               C.addressof(ms)))
    return mask, ms.value


def apply_dist_cpu(xyz, leg, quat=None):
    xyz = _f32(xyz, (-1, 3))
    d = np.zeros_like(xyz)
    v = np.zeros(len(xyz), np.uint8)
    ms = C.c_double(0)
    check(load().lrm_dist_cpu(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(d), _ptr(v),
                              C.addressof(ms)))
    return d, v, ms.value


def apply_rbdl_equiv(xyz, leg):
    """apply_RBDL's work restated (RBDL-equivalent LM position IK, parity unpinned) -> (converged uint8[n], ms); CPU."""
    xyz = _f32(xyz, (-1, 3))
    mask = np.zeros(len(xyz), np.uint8)
    ms = C.c_double(0)
    check(load().lrm_rbdl_equiv_cpu(_ptr(xyz), len(xyz), _ptr(_f32(leg, (14,))), _ptr(mask), C.addressof(ms)))
    return mask, ms.value


def morton_order(points):
    """Indices that put an (n,3) cloud in Morton order (compact tiles for the pair kernels)."""
    points = _f32(points, (-1, 3))
    out = np.zeros(len(points), np.uint64)
    check(load().lrm_morton_order(_ptr(points), len(points), _ptr(out)))
    return out.astype(np.int64)


class OctreeSettings(C.Structure):
    """LrmOctreeSettings (include/lrm.h): the octree knobs of settings.h:15-46."""
    _fields_ = [("box_center", C.c_float * 3), ("box_size", C.c_float * 3), ("min_box", C.c_float),
                ("enable_rot_below", C.c_float), ("convex_radius", C.c_float), ("angle_sample", C.c_int32 * 3),
                ("angle_minmax", C.c_float * 6), ("leg_count", C.c_int32), ("leg_mount", C.c_float * 8),
                ("leg_number_for_stab", C.c_int32), ("max_depth", C.c_int32)]


def octree_default_settings():
    s = OctreeSettings()
    load().lrm_octree_default_settings(C.addressof(s))
    return s


def apply_oct(footholds, leg, settings=None, capacity=None):
    """apply_oct (several_leg_octree.cu:391-488) -> (centres float32[k,3], kernel ms); GPU."""
    footholds = _f32(footholds, (-1, 3))
    cap = capacity if capacity is not None else 65536  # valid leaves; a larger tree makes the call run twice
    while True:
        out = np.zeros((max(cap, 1), 3), np.float32)
        n_out = C.c_size_t(0)
        ms = C.c_float(0)
        rc = load().lrm_apply_oct(_ptr(footholds), len(footholds), _ptr(_f32(leg, (14,))),
                                  None if settings is None else C.addressof(settings), _ptr(out), cap,
                                  C.addressof(n_out), C.addressof(ms))
        if rc == -1 and n_out.value > cap and capacity is None:
            cap = n_out.value
            continue
        if rc != 0:
            raise LrmError(f"liblrm error {rc}: {load().lrm_octree_last_error().decode()}")
        return out[: n_out.value].copy(), ms.value


OCT_EXCHANGE = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.c_void_p)


def _oct_call(fn, head_args, leg, settings, tail_args, exchange, capacity):
    """One of the lrm_apply_oct* entry points with an optional exchange callback.  `exchange(flags: np.ndarray[uint32])`
    must replace the array, in place, by its element-wise bitwise OR over all ranks.  An exception raised inside it is
    kept, turned into a non-zero return (the library then fails the call instead of going on with flags that were never
    combined) and re-raised here."""
    cap = capacity if capacity is not None else 65536  # valid leaves; a larger tree makes the call run twice
    raised = []

    def _cb(ptr, m, _user):
        try:
            exchange(np.ctypeslib.as_array(ptr, shape=(m,)))
            return 0
        except BaseException as e:  # noqa: BLE001 -- must not propagate through the C frames
            raised.append(e)
            return 1

    cb = OCT_EXCHANGE(_cb) if exchange is not None else None
    cb_arg = [C.cast(cb, C.c_void_p) if cb is not None else None, None]
    while True:
        out = np.zeros((max(cap, 1), 3), np.float32)
        n_out = C.c_size_t(0)
        ms = C.c_float(0)
        rc = fn(*head_args, _ptr(_f32(leg, (14,))), None if settings is None else C.addressof(settings), _ptr(out), cap,
                C.addressof(n_out), C.addressof(ms), *tail_args, *(cb_arg if (exchange is not None or tail_args) else []))
        if raised:
            raise raised[0]
        if rc == -1 and n_out.value > cap and capacity is None:
            cap = n_out.value  # every rank sees the same count and retries together
            continue
        if rc != 0:
            raise LrmError(f"liblrm error {rc}: {load().lrm_octree_last_error().decode()}")
        return out[: n_out.value].copy(), ms.value


def apply_oct_dev(x_ptr, y_ptr, z_ptr, n, leg, settings=None, rank=0, world=1, exchange=None, capacity=None):
    """lrm_apply_oct_dev: the footholds as three device arrays (raw pointers) -> (centres float32[k,3], kernel ms)."""
    return _oct_call(load().lrm_apply_oct_dev, (x_ptr, y_ptr, z_ptr, n), leg, settings, (rank, world), exchange, capacity)


def apply_oct_sharded(footholds, leg, settings, rank, world, exchange, capacity=None):
    """lrm_apply_oct_sharded (every rank holds all footholds, the children of a level are dealt round-robin)
    -> (centres float32[k,3], kernel ms); GPU."""
    footholds = _f32(footholds, (-1, 3))
    return _oct_call(load().lrm_apply_oct_sharded, (_ptr(footholds), len(footholds)), leg, settings, (rank, world), exchange, capacity)


def apply_oct_partitioned(local_footholds, leg, settings, exchange, capacity=None):
    """lrm_apply_oct_partitioned (every rank holds ITS part of the footholds, host array (n, 3)) -> (centres, kernel ms)"""
    f = _f32(local_footholds, (-1, 3))
    return _oct_call(load().lrm_apply_oct_partitioned, (_ptr(f), len(f)), leg, settings, (), exchange, capacity)


def apply_oct_partitioned_dev(x_ptr, y_ptr, z_ptr, n, leg, settings, exchange, capacity=None):
    """lrm_apply_oct_partitioned_dev: this rank's part of the footholds as three device arrays (raw pointers)"""
    return _oct_call(load().lrm_apply_oct_partitioned_dev, (x_ptr, y_ptr, z_ptr, n), leg, settings, (), exchange, capacity)


def dbg_sqrt_check_dev():
    """lrm_sqrtf vs the compiler's IEEE sqrtf on all 2^32 bit patterns, on the device -> (mismatches, first_bad)."""
    bad = C.c_uint64(0)
    first = C.c_uint32(0)
    check(load().lrm_dbg_sqrt_check_dev(C.byref(bad), C.byref(first)))
    return int(bad.value), int(first.value)


def dbg_fast_host(xyz, leg, quat=None):
    """Filtered evaluation on the host without fallback -> dict(mask, mask_unc, dist, valid, dist_unc)."""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    out = dict(mask=np.zeros(n, np.uint8), mask_unc=np.zeros(n, np.uint8), dist=np.zeros_like(xyz),
               valid=np.zeros(n, np.uint8), dist_unc=np.zeros(n, np.uint8))
    check(load().lrm_dbg_fast_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(out["mask"]),
                                   _ptr(out["mask_unc"]), _ptr(out["dist"]), _ptr(out["valid"]),
                                   _ptr(out["dist_unc"])))
    return out


def dbg_tol_host(xyz, leg, quat=None):
    """Contract-tolerance evaluation on the host, no re-evaluation -> (mask, dist, doubt bits uint32)."""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    mask, d, doubt = np.zeros(n, np.uint8), np.zeros_like(xyz), np.zeros(n, np.uint32)
    check(load().lrm_dbg_tol_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask), _ptr(d),
                                  _ptr(doubt)))
    return mask, d, doubt


def dbg_toltab_host(xyz, leg, quat=None):
    """LRM_MODE_TOL on the host with the plane table with deferred decisions -> (mask, vectors, doubt bits, table stats)"""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    mask, d, doubt = np.zeros(n, np.uint8), np.zeros_like(xyz), np.zeros(n, np.uint32)
    stats = np.zeros(5, np.uint32)
    check(load().lrm_dbg_toltab_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask), _ptr(d),
                                     _ptr(doubt), _ptr(stats)))
    return mask, d, doubt, dict(rows=int(stats[0]), vrows=int(stats[1]), refined=int(stats[2]), bytes=int(stats[3]),
                                second_candidates=int(stats[4]))


def dbg_xtab_host(xyz, leg, quat=None):
    """the bit-exact table-guided evaluation (csrc/lrm_point_xtab.h) on the host, no re-evaluation of its doubtful
    points -> (mask, vectors, doubt bits, dict(second_chains, bytes))"""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    mask, d, doubt = np.zeros(n, np.uint8), np.zeros_like(xyz), np.zeros(n, np.uint32)
    stats = np.zeros(2, np.uint32)
    check(load().lrm_dbg_xtab_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask), _ptr(d),
                                   _ptr(doubt), _ptr(stats)))
    return mask, d, doubt, dict(second_chains=int(stats[0]), bytes=int(stats[1]))


def dbg_replay_host(xyz, leg, quat=None):
    """the tolerance evaluation with the plane table + the strict replay of its decisions (LRM_MODE_TOL_REL's short-vector path)
    on the host -> (mask, vectors, doubt bits); vectors of points without doubt are the bit-exact ones"""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    mask, d, doubt = np.zeros(n, np.uint8), np.zeros_like(xyz), np.zeros(n, np.uint32)
    check(load().lrm_dbg_replay_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask), _ptr(d), _ptr(doubt)))
    return mask, d, doubt


def dbg_toltab_build(leg, quat=None, device=False):
    """the plane table of (leg, quat) from the host builder or the device builder -> (bytes as uint8 array, build milliseconds)"""
    size, ms = C.c_size_t(0), C.c_float(0)
    legp, q = _f32(leg, (14,)), _quat(quat)
    buf = np.zeros(24 << 20, np.uint8)
    check(load().lrm_dbg_toltab_build(_ptr(legp), _ptr(q), 1 if device else 0, _ptr(buf), buf.size, C.byref(size), C.byref(ms)))
    assert size.value <= buf.size
    return buf[:size.value].copy(), float(ms.value)


def dbg_toltab_bounds(xz, leg, quat=None):
    """the plane table's lower bound at plane points (abscissa - coxa_length, z) -> (bound, distance of the full plane
    evaluation, its validity, its doubt bits)"""
    xz = _f32(xz, (-1, 2))
    n = len(xz)
    lb, dist = np.zeros(n, np.float32), np.zeros(n, np.float32)
    valid, doubt = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
    check(load().lrm_dbg_toltab_bounds(_ptr(xz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(lb), _ptr(dist),
                                       _ptr(valid), _ptr(doubt)))
    return lb, dist, valid, doubt


def dbg_oct_trace(enable):
    check(load().lrm_dbg_oct_trace(1 if enable else 0))


def dbg_oct_trace_read():
    """records of the traced octree calls: float32[n, 12] = c[3], h[3], parent h[3], flag bits, parent_valid + 2 rot + 4 skip, depth"""
    n = C.c_size_t(0)
    check(load().lrm_dbg_oct_trace_read(None, 0, C.addressof(n)))
    out = np.zeros((int(n.value), 12), np.float32)
    check(load().lrm_dbg_oct_trace_read(_ptr(out), int(n.value), C.addressof(n)))
    return out


def dbg_pair_counts():
    """counting build only: (full evaluations, leg-sphere tests, footholds inside a reach sphere, footholds loaded) since the last call"""
    out = np.zeros(4, np.uint64)
    check(load().lrm_dbg_pair_counts(_ptr(out)))
    return [int(v) for v in out]


def dbg_tol_queue_counts():
    """(points, queued for the bit-exact fix-up, overflowed workgroup segments) of the last tolerance-mode call on device buffers"""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    check(load().lrm_dbg_tol_queue_counts(C.addressof(a), C.addressof(b), C.addressof(c)))
    return int(a.value), int(b.value), int(c.value)


def dbg_tol_ok(leg, quat=None):
    return bool(load().lrm_dbg_tol_ok(_ptr(_f32(leg, (14,))), _ptr(_quat(quat))))


def dbg_pair_sphere(leg, quat=None):
    """Bounding sphere of the pair test of one leg -> (centre[3] relative to the body position, r^2)."""
    out = np.zeros(4, np.float32)
    check(load().lrm_dbg_pair_sphere(_ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(out)))
    return out[:3].copy(), float(out[3])


def dbg_fused_reach_host(xyz, leg, quat=None):
    """The fused kernel's reach-from-distance by-product on the host, no fallback -> (mask, doubt)."""
    xyz = _f32(xyz, (-1, 3))
    n = len(xyz)
    mask, doubt = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    check(load().lrm_dbg_fused_reach_host(_ptr(xyz), n, _ptr(_f32(leg, (14,))), _ptr(_quat(quat)), _ptr(mask),
                                          _ptr(doubt)))
    return mask, doubt


def positionability(bodies, targets, legs, quats, reference_culls=False):
    """robot_full_struct's pipeline as a mask (several_leg.cu:326-877) -> (uint8[nb], ms); GPU."""
    bodies = _f32(bodies, (-1, 3))
    targets = _f32(targets, (-1, 3))
    legs = _f32(legs).reshape(-1, 14)
    quats = _f32(quats).reshape(-1, 4)
    out = np.zeros(len(bodies), np.uint8)
    ms = C.c_float(0)
    check(load().lrm_positionability(_ptr(bodies), len(bodies), _ptr(targets), len(targets), _ptr(legs),
                                     len(legs), _ptr(quats), len(quats), int(reference_culls), _ptr(out),
                                     C.addressof(ms)))
    return out, ms.value

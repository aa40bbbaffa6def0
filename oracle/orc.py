"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import
this module.  `Oracle` wraps oracle/liboracle.so (the plain-C restatement); `Ref` wraps
oracle/_ref/libref.so (the reference's own host path compiled from /root/reference by
oracle/Makefile) when that file exists.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LEG_FIELDS = (
    "body_angle", "body", "coxa_pitch", "coxa_length", "tibia_length", "femur_length",
    "tibia_absolute_pos", "tibia_absolute_neg", "max_angle_coxa", "min_angle_coxa",
    "max_angle_tibia", "min_angle_tibia", "max_angle_femur", "min_angle_femur",
)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def build(quiet=True):
    """(Re)build liboracle.so and, where /root/reference exists, _ref/libref.so."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class _Lib:
    def __init__(self, path):
        self.lib = C.CDLL(path)

    @staticmethod
    def leg_array(legs):
        a = _f32(legs).reshape(-1, 14)
        return a


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        super().__init__(path)
        L = self.lib
        L.orc_reachability_circles.restype = C.c_int
        L.orc_distance_circles.restype = C.c_int
        L.orc_reachable_rotate_leg.restype = C.c_int
        L.orc_in_sphere.restype = C.c_int
        L.orc_in_cylinder.restype = C.c_int

    def get_M2_leg(self, azimut=0.0):
        out = np.zeros(14, np.float32)
        self.lib.orc_get_M2_leg(C.c_float(azimut), _ptr(out))
        return out

    def get_moonbot_leg(self, azimut=0.0):
        out = np.zeros(14, np.float32)
        self.lib.orc_get_moonbot_leg(C.c_float(azimut), _ptr(out))
        return out

    def reach(self, xyz, leg, quat=(1, 0, 0, 0)):
        xyz = _f32(xyz, (-1, 3))
        leg = _f32(leg)
        q = _f32(quat)
        out = np.zeros(len(xyz), np.uint8)
        self.lib.orc_reach(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(leg), _ptr(q), _ptr(out))
        return out

    def dist(self, xyz, leg, quat=(1, 0, 0, 0)):
        xyz = _f32(xyz, (-1, 3))
        leg = _f32(leg)
        q = _f32(quat)
        d = np.zeros_like(xyz)
        v = np.zeros(len(xyz), np.uint8)
        self.lib.orc_dist(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(leg), _ptr(q), _ptr(d), _ptr(v))
        return d, v

    def rotate_leg_data(self, quat, leg):
        out = np.zeros(14, np.float32)
        self.lib.orc_rotate_leg_data(_ptr(_f32(quat)), _ptr(_f32(leg)), _ptr(out))
        return out

    def qt_rotate(self, q, v):
        out = np.zeros(3, np.float32)
        self.lib.orc_qt_rotate(_ptr(_f32(q)), _ptr(_f32(v)), _ptr(out))
        return out

    def qt_multiply(self, a, b):
        out = np.zeros(4, np.float32)
        self.lib.orc_qt_multiply(_ptr(_f32(a)), _ptr(_f32(b)), _ptr(out))
        return out

    def quat_from_vect_angle(self, axis, angle):
        out = np.zeros(4, np.float32)
        self.lib.orc_quat_from_vect_angle(_ptr(_f32(axis)), C.c_float(angle), _ptr(out))
        return out

    def reach_any(self, bodies, targets, legs, quat=(1, 0, 0, 0)):
        """out[leg, body] = any target reachable (reach_mem_kernel semantics)."""
        bodies = _f32(bodies, (-1, 3))
        targets = _f32(targets, (-1, 3))
        legs = self.leg_array(legs)
        out = np.zeros((len(legs), len(bodies)), np.uint8)
        self.lib.orc_reach_any(_ptr(bodies), C.c_size_t(len(bodies)), _ptr(targets),
                               C.c_size_t(len(targets)), _ptr(legs), C.c_size_t(len(legs)),
                               _ptr(_f32(quat)), _ptr(out))
        return out

    def in_sphere(self, radius, c, t):
        return self.lib.orc_in_sphere(C.c_float(radius), _ptr(_f32(c)), _ptr(_f32(t)))

    def in_cylinder(self, radius, plus_z, minus_z, c, t):
        return self.lib.orc_in_cylinder(C.c_float(radius), C.c_float(plus_z), C.c_float(minus_z),
                                        _ptr(_f32(c)), _ptr(_f32(t)))


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref.so"))


class Ref(_Lib):
    """The reference's own host path (oracle/_ref/libref.so)."""

    def __init__(self):
        super().__init__(os.path.join(_HERE, "_ref", "libref.so"))
        self.lib.ref_sizeof_leg.restype = C.c_size_t
        assert self.lib.ref_sizeof_leg() == 56

    def get_M2_leg(self, azimut=0.0):
        out = np.zeros(14, np.float32)
        self.lib.ref_get_M2_leg(C.c_float(azimut), _ptr(out))
        return out

    def get_moonbot_leg(self, azimut=0.0):
        out = np.zeros(14, np.float32)
        self.lib.ref_get_moonbot_leg(C.c_float(azimut), _ptr(out))
        return out

    def reach(self, xyz, leg, quat=(1, 0, 0, 0)):
        xyz = _f32(xyz, (-1, 3))
        out = np.zeros(len(xyz), np.uint8)
        self.lib.ref_reach(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(_f32(leg)), _ptr(_f32(quat)), _ptr(out))
        return out

    def dist(self, xyz, leg, quat=(1, 0, 0, 0)):
        xyz = _f32(xyz, (-1, 3))
        d = np.zeros_like(xyz)
        v = np.zeros(len(xyz), np.uint8)
        self.lib.ref_dist(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(_f32(leg)), _ptr(_f32(quat)), _ptr(d), _ptr(v))
        return d, v

    def reach_kernel_cpu(self, xyz, leg):
        xyz = _f32(xyz, (-1, 3))
        out = np.zeros(len(xyz), np.uint8)
        self.lib.ref_reach_kernel_cpu(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(_f32(leg)), _ptr(out))
        return out

    def dist_kernel_cpu(self, xyz, leg):
        xyz = _f32(xyz, (-1, 3))
        d = np.zeros_like(xyz)
        self.lib.ref_dist_kernel_cpu(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(_f32(leg)), _ptr(d))
        return d

    def reach_circles(self, xyz, leg):
        xyz = _f32(xyz, (-1, 3))
        out = np.zeros(len(xyz), np.uint8)
        self.lib.ref_reach_circles(_ptr(xyz), C.c_size_t(len(xyz)), _ptr(_f32(leg)), _ptr(out))
        return out

    def rotate_leg_data(self, quat, leg):
        out = np.zeros(14, np.float32)
        self.lib.ref_rotate_leg_data(_ptr(_f32(quat)), _ptr(_f32(leg)), _ptr(out))
        return out

    def qt_rotate(self, q, v):
        out = np.zeros(3, np.float32)
        self.lib.ref_qt_rotate(_ptr(_f32(q)), _ptr(_f32(v)), _ptr(out))
        return out

    def qt_multiply(self, a, b):
        out = np.zeros(4, np.float32)
        self.lib.ref_qt_multiply(_ptr(_f32(a)), _ptr(_f32(b)), _ptr(out))
        return out

    def quat_from_vect_angle(self, axis, angle):
        out = np.zeros(4, np.float32)
        self.lib.ref_quat_from_vect_angle(_ptr(_f32(axis)), C.c_float(angle), _ptr(out))
        return out

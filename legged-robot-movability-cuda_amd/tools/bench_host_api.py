#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: lrm_reach / lrm_dist / lrm_reach_dist on HOST buffers (the apply_kernel
path: allocate, copy in, kernel, copy out, free -- as the reference does per call), 1e7 points of the config-2 cloud.
Never the bench headline (bench.py measures device-resident inputs); DESIGN.md section 5 quotes these numbers."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrm_amd as lrm
rng = np.random.default_rng(42)
n = 10_000_000
lo, hi = np.array([-200, -500, -500], np.float32), np.array([700, 500, 300], np.float32)
pts = (rng.random((n, 3), dtype=np.float32) * (hi - lo) + lo)
leg = lrm.get_M2_leg(0.0)
import ctypes as C
from lrm_amd import _capi
L = lrm.lib()
legp, q = np.ascontiguousarray(leg, np.float32), np.array([1, 0, 0, 0], np.float32)
ms = C.c_float(0)


def call(name, mask, field):
    if name == "reach":
        _capi.check(L.lrm_reach(_capi._ptr(pts), n, _capi._ptr(legp), _capi._ptr(q), _capi._ptr(mask), C.addressof(ms)))
    elif name == "dist":
        _capi.check(L.lrm_dist(_capi._ptr(pts), n, _capi._ptr(legp), _capi._ptr(q), _capi._ptr(field), _capi._ptr(mask), C.addressof(ms)))
    else:
        _capi.check(L.lrm_reach_dist(_capi._ptr(pts), n, _capi._ptr(legp), _capi._ptr(q), _capi._ptr(mask), _capi._ptr(field), C.addressof(ms)))
    return ms.value


# The output arrays are the caller's: "fresh" = allocated per call and never touched, as bench.cpp:125-131 does with
# new[] (their pages are faulted in by the copy); "reused" = the same arrays again.  Allocation and release of the arrays
# stay outside the timed region.
for pipe in ("0", "1"):  # LRM_HOST_PIPELINE is read per call: chunked H2D || kernels || D2H through pinned slots when "1"
    os.environ["LRM_HOST_PIPELINE"] = pipe
    for mode_name, mode in (("bit-exact", lrm.MODE_FAST), ("tolerance", lrm.MODE_TOL)):
        lrm.set_mode(mode)
        for name in ("reach", "dist", "reach_dist"):
            if mode == lrm.MODE_TOL and name == "reach":
                continue
            keep_m, keep_f = np.ones(n, np.uint8), np.ones((n, 3), np.float32)
            call(name, keep_m, keep_f)
            res = {}
            for kind in ("fresh", "reused"):
                best, kms = 1e9, 0.0
                for _ in range(4):
                    m, f = (np.empty(n, np.uint8), np.empty((n, 3), np.float32)) if kind == "fresh" else (keep_m, keep_f)
                    t0 = time.perf_counter()
                    kms = call(name, m, f)
                    best = min(best, time.perf_counter() - t0)
                    del m, f
                res[kind] = best
            print("LRM_HOST_PIPELINE=%s %s mode, %s: fresh output arrays %.1f ms (%.2e eval/s incl. PCIe), reused %.1f ms (%.2e), kernel %.3f ms"
                  % (pipe, mode_name, name, res["fresh"] * 1e3, n / res["fresh"], res["reused"] * 1e3, n / res["reused"], kms), flush=True)
lrm.set_mode(lrm.MODE_FAST)

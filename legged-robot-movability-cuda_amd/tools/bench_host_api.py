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
for pipe in ("0", "1"):  # LRM_HOST_PIPELINE is read per call: chunked H2D || kernels || D2H on three streams when "1"
    os.environ["LRM_HOST_PIPELINE"] = pipe
    for mode_name, mode in (("bit-exact", lrm.MODE_FAST), ("tolerance", lrm.MODE_TOL)):
        lrm.set_mode(mode)
        for name, fn in (("reach", lambda: lrm.apply_reach(pts, leg)), ("dist", lambda: lrm.apply_dist(pts, leg)), ("reach_dist", lambda: lrm.apply_reach_dist(pts, leg))):
            if mode == lrm.MODE_TOL and name == "reach":
                continue
            fn()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
            print("LRM_HOST_PIPELINE=%s %s mode, %s: wall %.1f ms (%.2e eval/s incl. PCIe), kernel %.3f ms" % (pipe, mode_name, name, best * 1e3, n / best, r[-1]), flush=True)
lrm.set_mode(lrm.MODE_FAST)

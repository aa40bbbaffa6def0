#!/usr/bin/env python3
"""A distance sweep over the 45 body orientations of robot_full_struct (several_leg.cu:831-857: roll x pitch x yaw samples),
one fused reach+distance call per orientation on the same 1e7-point cloud, in the table-guided modes: every orientation is a
new (leg, orientation) pair, i.e. a new plane table.  Wall time of the 45 calls (cold caches: lrm_release_workspaces first)
with the device table builder and with the host builder (LRM_TOLTAB_HOST=1).  One JSON line.

    python legged-robot-movability-cuda_amd/tools/bench_orientation_sweep.py [--points 10000000] [--mode tol_rel]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def quats():
    out = []
    for r in np.linspace(-0.3, 0.3, 3):
        for p in np.linspace(-0.3, 0.3, 3):
            for y in np.linspace(-0.6, 0.6, 5):
                cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
                out.append(np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                                     cr * cp * sy - sr * sp * cy], np.float32))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--mode", default="tol_rel", choices=["tol", "tol_rel", "fast"])
    args = ap.parse_args()
    import torch
    import lrm_amd as lrm
    n = args.points
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    lo = torch.tensor([-200.0, -500.0, -500.0], device="cuda").view(3, 1)
    hi = torch.tensor([700.0, 500.0, 300.0], device="cuda").view(3, 1)
    cloud = (torch.rand((3, n), device="cuda", generator=g) * (hi - lo) + lo).contiguous()
    mask = torch.empty(n, dtype=torch.uint8, device="cuda")
    field = torch.empty((3, n), dtype=torch.float32, device="cuda")
    leg = lrm.get_M2_leg(0.0)
    qs = quats()
    lrm.set_mode({"tol": lrm.MODE_TOL, "tol_rel": lrm.MODE_TOL_REL, "fast": lrm.MODE_FAST}[args.mode])
    out = {"workload": f"{len(qs)} orientations x fused reach+distance on {n} points, mode {args.mode}, a new plane table per orientation"}

    def sweep():
        t0 = time.perf_counter()
        for q in qs:
            lrm.device.reach_dist(cloud[0], cloud[1], cloud[2], leg, q, mask=mask, out=field)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for name, env in (("device_builder", None), ("host_builder", "1")):
        if env:
            os.environ["LRM_TOLTAB_HOST"] = env
        else:
            os.environ.pop("LRM_TOLTAB_HOST", None)
        lrm.release_workspaces()
        lrm.device.reach_dist(cloud[0], cloud[1], cloud[2], leg, None, mask=mask, out=field)  # queues, builder scratch, code load
        torch.cuda.synchronize()
        cold = sweep()
        warm = sweep()  # every table cached
        out[name] = {"cold_ms": cold, "cached_ms": warm, "per_new_orientation_ms": (cold - warm) / len(qs)}
    os.environ.pop("LRM_TOLTAB_HOST", None)
    lrm.set_mode(lrm.MODE_FAST)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

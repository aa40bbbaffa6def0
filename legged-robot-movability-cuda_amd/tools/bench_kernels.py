#!/usr/bin/env python3
"""SURVEY.md section 8(d) measurement table on one MI355X: reach / distance / fused kernels on
  C1  the bench.cpp planar grid (Pix = 1.0: 106 704 points) and its intended z-range variant,
  C2  1e7 uniform-random points (the headline workload),
  all-unreachable and (nearly) all-reachable clouds, which bracket the divergence,
with HIP events around each launch, 20 warm-up + 100 timed repetitions, median; plus the box's
achievable copy bandwidth (a device-to-device copy of the same SoA arrays) next to the nominal
8 TB/s.  One JSON line.

    python legged-robot-movability-cuda_amd/tools/bench_kernels.py [--points 10000000] [--reps 100]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--warm", type=int, default=20)
    args = ap.parse_args()
    import torch
    import lrm_amd
    from lrm_amd import device, workloads
    leg = lrm_amd.get_M2_leg(0.0)

    def timed(fn):
        for _ in range(args.warm):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    def kernels(name, pts):
        n = len(pts)
        t = torch.from_numpy(np.ascontiguousarray(pts.T)).cuda()
        x, y, z = t[0], t[1], t[2]
        mask = torch.empty(n, dtype=torch.uint8, device="cuda")
        bits = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
        d = torch.empty((3, n), dtype=torch.float32, device="cuda")
        valid = torch.empty(n, dtype=torch.uint8, device="cuda")
        ms_r = timed(lambda: device.reach(x, y, z, leg, out=mask, bits=bits))
        ms_d = timed(lambda: device.dist(x, y, z, leg, out=d, valid=valid))
        ms_f = timed(lambda: device.reach_dist(x, y, z, leg, mask=mask, out=d, bits=bits))
        frac = float(mask.float().mean().item())
        row = {"workload": name, "points": n, "reachable_fraction": frac}
        for key, ms, by in (("reach", ms_r, 13.125), ("dist", ms_d, 25.0), ("fused", ms_f, 25.125)):
            row[key] = {"ms": ms, "ns_per_point": ms * 1e6 / n, "evals_per_s": n / (ms * 1e-3),
                        "algorithmic_GBs": n * by / (ms * 1e-3) / 1e9}
        return row

    rng = np.random.default_rng(42)
    n = args.points
    rows = [kernels("C1 bench.cpp grid, Pix 1.0 (z from XMin as committed)", workloads.bench_grid(1.0)),
            kernels("C1 grid, intended z range [-350, 51]", workloads.bench_grid(1.0, z_from_xmin=False)),
            kernels("C2 uniform random cube", workloads.random_cloud(n, seed=42))]
    far = (rng.random((n, 3), dtype=np.float32) * np.float32(400) + np.float32(900)).astype(np.float32)
    rows.append(kernels("all unreachable (cube at 0.9-1.3 m)", far))
    # a small box inside the workspace of the M2 leg (in front of the coxa, below the femur joint)
    near = np.stack([rng.uniform(330, 370, n), rng.uniform(-30, 30, n), rng.uniform(-230, -190, n)], 1).astype(np.float32)
    rows.append(kernels("(nearly) all reachable (40 mm box inside the workspace)", near))
    # copy bandwidth of the box: 12 B read + 12 B written per point
    src = torch.empty((3, n), dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    ms_c = timed(lambda: dst.copy_(src))
    res = {"device": torch.cuda.get_device_name(0), "reps": args.reps, "warm": args.warm, "rows": rows,
           "copy_kernel": {"ms": ms_c, "GBs": n * 24 / (ms_c * 1e-3) / 1e9, "nominal_GBs": 8000.0}}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel times of the fused reach+distance launch in the three arithmetic modes on one GPU (HIP events, steady
clocks), and the doubt statistics of the tolerance mode.  Usage: python tools/bench_modes.py [--points N]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--modes", default="tol,fast,strict", help="comma-separated: tol, tol_rel, fast, strict")
    ap.add_argument("--leg", default="m2", choices=["m2", "moonbot"])
    ap.add_argument("--azimut", type=float, default=0.0, help="body angle of the leg (rad)")
    ap.add_argument("--cloud", default="cube", choices=["cube", "grid", "reachable", "far"],
                    help="cube: config 2 (uniform in the leg's bounding cube); grid: the reference's planar bench grid (y = 0), random samples of it; reachable / far: only reachable points (resampled from the cube) / only points beyond the workspace: the two ends of the lane divergence")
    args = ap.parse_args()
    import torch
    import lrm_amd
    rng = np.random.default_rng(42)
    lo, hi = np.array([-200, -500, -500], np.float32), np.array([700, 500, 300], np.float32)
    n = args.points
    host = np.empty((3, n), np.float32)
    for s in range(0, n, 2_000_000):
        e = min(n, s + 2_000_000)
        host[:, s:e] = (rng.random((e - s, 3), dtype=np.float32) * (hi - lo) + lo).T
    if args.cloud == "grid":  # bench.cpp:109-120: x in [-100, 601], y = 0, z in [-100, 51]
        host[0] = rng.uniform(-100, 601, n).astype(np.float32)
        host[1] = 0
        host[2] = rng.uniform(-100, 51, n).astype(np.float32)
    if args.cloud == "far":
        host[0] += 900.0
    cloud = torch.from_numpy(host).cuda()
    if args.cloud == "reachable":  # keep the reachable points of a few cubes until n are collected
        lrm_amd.set_mode(lrm_amd.MODE_FAST)
        parts, have, seed = [], 0, 1
        while have < n:
            g = torch.Generator(device="cuda")
            g.manual_seed(seed)
            seed += 1
            c = torch.rand((3, n), device="cuda", generator=g) * torch.tensor(hi - lo, device="cuda").view(3, 1) + torch.tensor(lo, device="cuda").view(3, 1)
            m = lrm_amd.device.reach(c[0].contiguous(), c[1].contiguous(), c[2].contiguous(), lrm_amd.get_M2_leg(0.0))
            keep = c[:, m.bool()]
            parts.append(keep)
            have += keep.shape[1]
        cloud = torch.cat(parts, dim=1)[:, :n].contiguous()
    x, y, z = cloud[0], cloud[1], cloud[2]
    leg = lrm_amd.get_M2_leg(args.azimut) if args.leg == "m2" else lrm_amd.get_moonbot_leg(args.azimut)
    mask = torch.empty(n, dtype=torch.uint8, device="cuda")
    field = torch.empty((3, n), dtype=torch.float32, device="cuda")
    bits = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
    out = {"points": n}
    modes = {"tol": lrm_amd.MODE_TOL, "tol_rel": lrm_amd.MODE_TOL_REL, "fast": lrm_amd.MODE_FAST, "strict": lrm_amd.MODE_STRICT}
    for name in args.modes.split(","):
        lrm_amd.set_mode(modes[name])
        fn = lambda: lrm_amd.device.reach_dist(x, y, z, leg, None, mask=mask, out=field, bits=bits)
        for _ in range(150):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.reps
        if name == "tol":
            try:
                out["tol_queue_counts(points, full, exact)"] = lrm_amd.dbg_tol_queue_counts()
            except Exception as e:  # below the plane-table threshold
                out["tol_queue_counts"] = str(e)
        if name == "tol":  # the variant under test against the bit-exact mode, on the device
            lrm_amd.set_mode(lrm_amd.MODE_FAST)
            m2 = torch.empty_like(mask)
            f2 = torch.empty_like(field)
            b2 = torch.empty_like(bits)
            lrm_amd.device.reach_dist(x, y, z, leg, None, mask=m2, out=f2, bits=b2)
            lrm_amd.set_mode(modes[name])
            err = (field - f2).norm(dim=0) / torch.maximum(f2.norm(dim=0), (cloud.norm(dim=0) + float(leg[1])) / 8)
            out["tol_check"] = {"mask_mismatches": int((mask != m2).sum()), "bit_word_mismatches": int((bits != b2).sum()),
                                "max_err": float(err.max()), "nonfinite": int((~torch.isfinite(field)).sum())}
            del m2, f2, b2, err
        out[name] = {"ms_per_call": ms, "evals_per_s": n / (ms * 1e-3), "hbm_GBs_algorithmic": 25 * n / (ms * 1e-3) / 1e9,
                     "frac_of_8TBs": 25 * n / (ms * 1e-3) / 8e12}
    lrm_amd.set_mode(lrm_amd.MODE_FAST)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

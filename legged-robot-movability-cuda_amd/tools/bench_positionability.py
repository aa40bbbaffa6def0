#!/usr/bin/env python3
"""BASELINE config 3: 6-leg positionability, body poses x terrain cloud, one MI355X.
Times lrm_reach_any_dev (all legs, all targets, one launch) with HIP events and prints one JSON
line; --check compares a random subset of bodies with the brute-force oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=100_000)
    ap.add_argument("--terrain-side", type=int, default=316)  # 316^2 = 99 856 points
    ap.add_argument("--legs", type=int, default=6)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--warm", type=int, default=50, help="untimed launches first: the GPU needs ~50 ms of load to reach its steady clocks")
    ap.add_argument("--check", type=int, default=0, help="verify this many random bodies against the oracle")
    ap.add_argument("--mode", choices=["strict", "fast"], default="fast")
    ap.add_argument("--morton", action="store_true", help="feed both clouds in Morton order (lrm_morton_order)")
    ap.add_argument("--sweep", action="store_true",
                    help="also time the whole robot_full_struct pipeline: 45 orientations + the reference culls")
    args = ap.parse_args()
    import torch
    import lrm_amd
    from lrm_amd import workloads
    lrm_amd.set_mode(lrm_amd.MODE_FAST if args.mode == "fast" else lrm_amd.MODE_STRICT)
    ground = workloads.terrain(args.terrain_side)
    bodies = workloads.body_lattice(ground, args.bodies)
    if args.morton:
        ground = ground[lrm_amd.morton_order(ground)]
        bodies = bodies[lrm_amd.morton_order(bodies)]
    legs = workloads.hexapod(lrm_amd.get_M2_leg, args.legs)
    tb = torch.from_numpy(np.ascontiguousarray(bodies.T)).cuda()
    tt = torch.from_numpy(np.ascontiguousarray(ground.T)).cuda()
    out = torch.empty((len(legs), len(bodies)), dtype=torch.uint8, device="cuda")
    alll = torch.empty(len(bodies), dtype=torch.uint8, device="cuda")
    run = lambda: lrm_amd.device.reach_any(tb[0], tb[1], tb[2], tt[0], tt[1], tt[2], legs, None, out=out, all_legs=alll)
    for _ in range(max(args.warm, 1)):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.reps
    pairs = float(len(bodies)) * len(ground) * len(legs)
    res = {"workload": f"config 3: {len(bodies)} body poses x {len(ground)} terrain points x {len(legs)} legs",
           "mode": args.mode, "morton_order": bool(args.morton), "ms": ms, "leg_target_pairs_answered_per_s": pairs / (ms * 1e-3),
           "positionable_fraction": float(alll.float().mean().item()),
           "per_leg_fraction": out.float().mean(dim=1).cpu().tolist()}
    if args.check:
        from oracle.orc import Oracle
        o = Oracle()
        idx = np.sort(np.random.default_rng(0).choice(len(bodies), args.check, replace=False))
        t0 = time.time()
        want = o.reach_any(bodies[idx], ground, legs)
        res["oracle_check"] = {"bodies": int(args.check), "seconds": time.time() - t0,
                               "identical": bool(np.array_equal(out.cpu().numpy()[:, idx], want))}
    if args.sweep:
        quats = workloads.reference_sweep_quats()
        # scalar-first quaternions (qtRotate's convention) for a sweep in which poses are feasible
        yaw = np.linspace(0, np.pi / 2, 5)
        tilt = [-np.pi / 8, 0.0, np.pi / 8]
        sane = np.array([[np.cos(a / 2) * np.cos(t / 2), 0, np.cos(a / 2) * np.sin(t / 2), np.sin(a / 2) * np.cos(t / 2)]
                         for t in tilt for a in yaw] * 3, np.float32)
        for name, qs in (("reference_quats", quats), ("scalar_first_quats", sane)):
            t0 = time.time()
            accepted, kms = lrm_amd.positionability(bodies, ground, legs[:4], qs, reference_culls=True)
            res["sweep_" + name] = {"orientations": len(qs), "legs": 4, "kernel_ms": kms, "wall_s": time.time() - t0,
                                    "accepted": int(accepted.sum())}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

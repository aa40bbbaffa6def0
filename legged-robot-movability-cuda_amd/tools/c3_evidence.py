#!/usr/bin/env python3
"""BASELINE config 3 on the reference's own terrain and body lattice (tests/golden/terrain_ground.npz, Morton order, six M2
legs, identity orientation): ONE launch of lrm_reach_any_dev per repetition.

    python tools/c3_evidence.py --reps 50                 timing of the shipped library (JSON line)
    LRM_LIB_PATH=.../liblrm_count.so python tools/c3_evidence.py --count
                                                          a -DLRM_PAIR_COUNT build: how many (leg, target) pairs one launch
                                                          really evaluates (the kernel answers nb * nt * nlegs pairs, most of
                                                          them by bounding boxes and spheres)
tools/c3_profile.sh runs both plus the rocprofv3 passes and writes profiles/rNN_c3_evidence.json."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--warm", type=int, default=30)
    ap.add_argument("--count", action="store_true")
    args = ap.parse_args()
    import torch
    import lrm_amd
    from lrm_amd import workloads
    t = np.load(os.path.join(ROOT, "tests", "golden", "terrain_ground.npz"))
    ground, bodies = t["ground"], t["bodies"]
    ground = ground[lrm_amd.morton_order(ground)]
    bodies = bodies[lrm_amd.morton_order(bodies)]
    legs = workloads.hexapod(lrm_amd.get_M2_leg, 6)
    tb = torch.from_numpy(np.ascontiguousarray(bodies.T)).cuda()
    tt = torch.from_numpy(np.ascontiguousarray(ground.T)).cuda()
    out = torch.empty((6, len(bodies)), dtype=torch.uint8, device="cuda")
    alll = torch.empty(len(bodies), dtype=torch.uint8, device="cuda")
    run = lambda: lrm_amd.device.reach_any(tb[0], tb[1], tb[2], tt[0], tt[1], tt[2], legs, None, out=out, all_legs=alll)
    res = {"workload": f"{len(bodies)} body poses x {len(ground)} terrain points x 6 M2 legs, identity orientation, Morton order",
           "pairs_answered": int(len(bodies)) * int(len(ground)) * 6}
    if args.count:
        run()
        torch.cuda.synchronize()
        lrm_amd.dbg_pair_counts()  # reset (the first launch also built the bounding boxes)
        run()
        c = lrm_amd.dbg_pair_counts()
        res.update({"pairs_evaluated": c[0], "leg_sphere_tests": c[1], "footholds_inside_reach_sphere": c[2], "footholds_loaded": c[3],
                    "evaluated_fraction_of_answered": c[0] / res["pairs_answered"]})
    else:
        for _ in range(args.warm):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            run()
        b.record()
        torch.cuda.synchronize()
        res.update({"ms": a.elapsed_time(b) / args.reps, "positionable_fraction": float(alll.float().mean().item())})
    print(json.dumps(res))


if __name__ == "__main__":
    main()

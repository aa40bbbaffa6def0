"""Times lrm_apply_oct (the level-synchronous redesign of apply_oct, several_leg_octree.cu:391-488) on
synthetic footholds.  One JSON line per configuration: leaves found, wall seconds, the library's own
kernel milliseconds.  No reference timing exists for this path (it is dead code upstream).

    python legged-robot-movability-cuda_amd/tools/bench_octree.py [--footholds 2000] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lrm_amd as lrm  # noqa: E402


def footholds(n, seed, spread):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-spread, spread, (n, 2))
    z = 20 * np.sin(xy[:, 0] / 120) + rng.normal(0, 4, n) - 150
    return np.column_stack([xy, z]).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--footholds", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--scale", type=int, default=0,
                    help="config-5 shape instead: this many footholds of a 10 m x 10 m relief, the reference's settings "
                         "(root box +-5000 mm, min box 100 mm, 27 orientations below 50 mm) at --depth")
    ap.add_argument("--depth", type=int, default=6)
    args = ap.parse_args()
    dim = lrm.get_M2_leg(0.0)
    if args.scale:
        rng = np.random.default_rng(5)
        n = args.scale
        xy = rng.uniform(-5000, 5000, (n, 2)).astype(np.float32)
        z = (400 * np.sin(xy[:, 0] / 900) * np.cos(xy[:, 1] / 700) + 60 * np.sin(xy[:, 0] / 130) + rng.normal(0, 5, n) - 200).astype(np.float32)
        f = np.column_stack([xy, z]).astype(np.float32)
        st = lrm.octree_default_settings()
        st.max_depth = args.depth
        st.leg_number_for_stab = 3
        t0 = time.perf_counter()
        out, ms = lrm.apply_oct(f, dim, st)
        wall = time.perf_counter() - t0
        print(json.dumps({"workload": f"apply_oct, config-5 shape: {n} footholds on a 10 m x 10 m relief, depth {args.depth}, 4 legs, stability 3",
                          "leaves": len(out), "wall_s_first_call": wall, "kernel_ms": ms}), flush=True)
        t0 = time.perf_counter()
        out, ms = lrm.apply_oct(f, dim, st)
        print(json.dumps({"second_call_wall_s": time.perf_counter() - t0, "kernel_ms": ms, "leaves": len(out)}), flush=True)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(f.T)).cuda()
        lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, ms = lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
        print(json.dumps({"device_resident_footholds_wall_s": time.perf_counter() - t0, "kernel_ms": ms, "leaves": len(out)}), flush=True)
        return
    cases = [
        ("two legs, 0.8 m root, depth 6", dict(half=400.0, depth=6, stab=2, legs=2, mounts=(0.0, 0.3))),
        ("four legs, stability 3, 0.8 m root, depth 5", dict(half=400.0, depth=5, stab=3, legs=4, mounts=None)),
        ("four legs, stability 1, rotations from the root, depth 4",
         dict(half=300.0, depth=4, stab=1, legs=4, mounts=None, rot_below=400.0)),
    ]
    f = footholds(args.footholds, seed=404, spread=600.0)
    for name, c in cases:
        st = lrm.octree_default_settings()
        for i in range(3):
            st.box_size[i] = c["half"]
        st.max_depth = c["depth"]
        st.leg_number_for_stab = c["stab"]
        st.leg_count = c["legs"]
        if c.get("rot_below") is not None:
            st.enable_rot_below = c["rot_below"]
        if c.get("mounts"):
            for i, m in enumerate(c["mounts"]):
                st.leg_mount[i] = m
        lrm.apply_oct(f, dim, st)  # warm-up (allocations, code load)
        best_wall, best_ms, leaves = 1e30, 1e30, 0
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out, ms = lrm.apply_oct(f, dim, st)
            best_wall = min(best_wall, time.perf_counter() - t0)
            best_ms = min(best_ms, ms)
            leaves = len(out)
        print(json.dumps({"workload": f"apply_oct: {name}, {args.footholds} footholds", "leaves": leaves,
                          "wall_s": best_wall, "kernel_ms": best_ms}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised campaign for LRM_MODE_TOL on one GPU: random leg geometries and joint limits around the two reference
robots, random body orientations, and clouds chosen to be hard -- uniform, far (beyond the inner grid of the plane table), near
and far interleaved, the planar bench grid, a cluster around the coxa axis, and a cloud pushed ONTO the workspace boundary (every point moved along its own distance vector, then jittered
by 1e-4 .. 1e-1 mm).  The bit-exact mode (LRM_MODE_FAST, itself checked bit for bit against the oracle by the test
suite) is the reference, on the device: the reach mask and bit words must be identical, the distance field within the
contract tolerance.  Prints one JSON line; exit code 1 on any violation.

    python legged-robot-movability-cuda_amd/tools/stress_tol.py [--legs 40] [--points 400000] [--seed 1]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", type=int, default=40)
    ap.add_argument("--points", type=int, default=400_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tilt", type=float, default=1.0, help="scale of the random body rotation (1: up to ~40 degrees)")
    ap.add_argument("--rel", action="store_true",
                    help="LRM_MODE_TOL_REL instead of LRM_MODE_TOL: the error is the LITERAL relative one, |d - d_ref| / |d_ref| "
                         "(0 / 0 = 0), against the same 1e-5")
    ap.add_argument("--exact", action="store_true",
                    help="compare LRM_MODE_FAST with LRM_MODE_STRICT instead (the filtered kernels against the plain restatement "
                         "of the reference's arithmetic): every float of the field must be bit-identical")
    args = ap.parse_args()
    import torch
    import lrm_amd as lrm
    rng = np.random.default_rng(args.seed)
    n = args.points
    tol = 1e-5
    out = {"legs": 0, "tol_eligible": 0, "cases": 0, "evaluations": 0, "mask_mismatches": 0, "bit_word_mismatches": 0,
           "nonfinite": 0, "max_err": 0.0, "worst": None, "max_queued_fraction": 0.0, "mean_queued_fraction": 0.0,
           "overflowed_segments": 0}
    queued = []

    def run(mode, x, y, z, leg, q):
        lrm.set_mode(mode)
        m, d, b = lrm.device.reach_dist(x, y, z, leg, q, mask=torch.empty(len(x), dtype=torch.uint8, device="cuda"),
                                        bits=torch.empty((len(x) + 63) // 64, dtype=torch.int64, device="cuda"))
        return m, d, b

    for li in range(args.legs):
        if li == 0:
            leg = lrm.get_M2_leg(0.0)
        elif li == 1:
            leg = lrm.get_moonbot_leg(0.7)
        else:
            leg = lrm.leg_factory(float(rng.uniform(-3.1, 3.1)), float(rng.uniform(60, 260)), float(rng.uniform(-60, 20)),
                                  float(rng.uniform(30, 110)), float(rng.uniform(90, 180)), float(rng.uniform(90, 200)),
                                  float(rng.uniform(25, 85)), float(rng.uniform(50, 100)), float(rng.uniform(80, 140)),
                                  float(rng.uniform(-15, 10)), float(rng.uniform(-15, 10)))
        out["legs"] += 1
        print(f"leg {li + 1} / {args.legs}: {out['evaluations']} evaluations so far, max error {out['max_err']:.3e}", file=sys.stderr, flush=True)  # (a sign of life for the job runner)
        reach = float(leg[1] + leg[3] + leg[4] + leg[5])  # body + coxa + tibia + femur lengths
        for qi in range(3):
            if qi == 0:
                q = np.array([1, 0, 0, 0], np.float32)
            else:
                q = rng.normal(size=4) * np.array([1.0, 0.25 * args.tilt, 0.25 * args.tilt, 0.35 * args.tilt])
                q[0] = abs(q[0]) + 0.8
                q = (q / np.linalg.norm(q)).astype(np.float32)
            out["tol_eligible"] += int(lrm.dbg_tol_ok(leg, q))
            clouds = {
                "uniform": (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(1.15 * reach),
                "grid": np.column_stack([rng.uniform(-100, 601, n), np.zeros(n), rng.uniform(-350, 51, n)]).astype(np.float32),
                "axis": (rng.normal(size=(n, 3)) * np.array([12.0, 12.0, 150.0]) + np.array([leg[1], 0, 0])).astype(np.float32),
                # beyond the inner grid of the plane table (+-1024 mm of the femur joint): the outer grid and its bound
                "far": ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(1.15 * reach) + np.array([1.6 * reach, 0.8 * reach, 0.0])).astype(np.float32),
                # several metres out: the outer grid's own decision band
                "far3": ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(1.5 * reach) + np.array([5.0 * reach, 2.0 * reach, reach])).astype(np.float32),
            }
            # near and far points interleaved: every wave takes the outer grid for all its lanes
            mixed = clouds["uniform"].copy()
            mixed[::7] = clouds["far"][::7]
            mixed[3::64] *= np.float32(9.0)  # and a few beyond the outer grid (+-8192 mm) and the table's band limit
            clouds["mixed"] = mixed
            # around the coxa axis for THIS orientation: points chosen in the coxa frame (radius up to 60 mm from the axis,
            # where the yaw direction amplifies every rounding by target radius / r), mapped back to the body frame:
            # x_coxa = Rp (Rz (Rq^-1 p) - (body, 0, 0)),  Rz = rotation by -body_angle about z, Rp = by -coxa_pitch about y
            rr = rng.uniform(0, 60, n) ** 1.0
            th = rng.uniform(-np.pi, np.pi, n)
            c = np.column_stack([rr * np.cos(th), rr * np.sin(th), rng.uniform(-1.1 * reach, 1.1 * reach, n)])
            pit = float(leg[2])
            cp, sp = np.cos(pit), np.sin(pit)  # Rp^-1: rotation by +coxa_pitch
            v = np.column_stack([c[:, 0] * cp - c[:, 2] * sp, c[:, 1], c[:, 0] * sp + c[:, 2] * cp])
            v[:, 0] += float(leg[1])
            ba = float(leg[0])
            cb, sb = np.cos(ba), np.sin(ba)    # Rz^-1: rotation by +body_angle
            v = np.column_stack([v[:, 0] * cb - v[:, 1] * sb, v[:, 0] * sb + v[:, 1] * cb, v[:, 2]])
            qw, qv = float(q[0]), np.asarray(q[1:], np.float64)
            tq = 2 * np.cross(qv, v)
            clouds["coxa_axis"] = (v + qw * tq + np.cross(qv, tq)).astype(np.float32)  # rotate by q
            # boundary cloud: uniform points moved along their own distance vector, jittered
            u = clouds["uniform"]
            t = torch.from_numpy(np.ascontiguousarray(u.T)).cuda()
            _, d0, _ = run(lrm.MODE_FAST, t[0], t[1], t[2], leg, q)
            jitter = (10.0 ** rng.uniform(-4, -1, (n, 1))) * rng.normal(size=(n, 3))
            clouds["boundary"] = (u - d0.cpu().numpy().T + jitter).astype(np.float32)
            for name, pts in clouds.items():
                t = torch.from_numpy(np.ascontiguousarray(pts.T)).cuda()
                m1, d1, b1 = run(lrm.MODE_STRICT if args.exact else lrm.MODE_FAST, t[0], t[1], t[2], leg, q)
                m2, d2, b2 = run(lrm.MODE_FAST if args.exact else (lrm.MODE_TOL_REL if args.rel else lrm.MODE_TOL), t[0], t[1], t[2], leg, q)
                torch.cuda.synchronize()
                if not args.exact:
                    try:  # how much of the cloud the tolerance kernel handed to the bit-exact fix-up (doubt bands + unanswered table cells)
                        npts, nq, nover = lrm.dbg_tol_queue_counts()
                        if npts == len(pts):
                            queued.append((nq / npts, name))
                            out["overflowed_segments"] += nover
                    except lrm.LrmError:
                        pass
                if args.exact:  # "error" = 1 for every point with a differing bit pattern (nan == nan)
                    same = (d1.view(torch.int32) == d2.view(torch.int32)) | (torch.isnan(d1) & torch.isnan(d2))
                    err = (~same.all(dim=0)).to(torch.float32)
                elif args.rel:
                    err = torch.nan_to_num((d2 - d1).norm(dim=0) / d1.norm(dim=0), nan=0.0, posinf=float("inf"))  # 0 / 0: equal zero vectors
                    err = torch.where(torch.isfinite(d1).all(dim=0), err, torch.zeros_like(err))  # non-finite inputs: compared by `nonfinite`
                else:
                    err = (d2 - d1).norm(dim=0) / torch.maximum(d1.norm(dim=0), (t.norm(dim=0) + float(leg[1])) / 8)
                    err = torch.nan_to_num(err, nan=0.0)  # 0 / 0 at a point on the boundary with a zero vector in both modes
                bad_m = int((m1 != m2).sum())
                bad_b = int((b1 != b2).sum())
                nonfinite = int((torch.isfinite(d1) != torch.isfinite(d2)).sum())
                e = float(err.max())
                out["cases"] += 1
                out["evaluations"] += len(pts)
                out["mask_mismatches"] += bad_m
                out["bit_word_mismatches"] += bad_b
                out["nonfinite"] += nonfinite
                if e > out["max_err"]:
                    out["max_err"] = e
                    i = int(err.argmax())
                    out["worst"] = {"leg": [float(v) for v in leg], "quat": [float(v) for v in q], "cloud": name,
                                    "point": [float(v) for v in pts[i]], "fast": [float(v) for v in d1[:, i].cpu()],
                                    "tol": [float(v) for v in d2[:, i].cpu()]}
                if bad_m or bad_b or nonfinite or e > tol:
                    print(json.dumps({"violation": {"leg": [float(v) for v in leg], "quat": [float(v) for v in q], "cloud": name,
                                                    "mask": bad_m, "bits": bad_b, "nonfinite": nonfinite, "err": e}}), flush=True)
    lrm.set_mode(lrm.MODE_FAST)
    if queued:
        out["max_queued_fraction"] = max(q for q, _ in queued)
        out["mean_queued_fraction"] = float(np.mean([q for q, _ in queued]))
        out["mean_queued_fraction_by_cloud"] = {nm: float(np.mean([q for q, k in queued if k == nm])) for nm in sorted({k for _, k in queued})}
    print(json.dumps(out))
    return 1 if (out["mask_mismatches"] or out["bit_word_mismatches"] or out["nonfinite"] or out["max_err"] > tol) else 0


if __name__ == "__main__":
    sys.exit(main())

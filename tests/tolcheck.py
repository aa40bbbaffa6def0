"""The acceptance metric of LRM_MODE_TOL (include/lrm.h), shared by the CPU and GPU tests.

FROZEN (round 3): the metric below is part of the mode's contract in include/lrm.h and does not move when a test
fails -- a failing point is a bug in the kernel (or goes to the bit-exact fix-up), not a reason to widen the floor.
It is a FLOORED reading of BASELINE.json's "1e-5 relative", i.e. a deviation from the literal text for short vectors;
the literal relative error is asserted where float32 can deliver it (|d_ref| >= 16 mm: tests/test_gpu_tol.py) and
reported everywhere else (bench.py "tolerance_check").  LRM_MODE_FAST meets the literal text (tolerance 0).

BASELINE.json asks for the distance field "within 1e-5 relative".  The field is a difference of positions:
d = p - (nearest boundary point), and the reference computes it in the coxa frame: its first operation on the point is
`x -= body` (place_over_coxa, one_leg.cu:13), so the coordinates it works with have magnitude up to |p| + body in
float32 and |d| carries an absolute uncertainty of a few ulp of THAT (~1e-4 mm) in ANY float32 implementation -- the
reference's own host and CUDA builds differ by that much.  A purely relative bound is therefore meaningless for short
vectors; the metric is

    err(i) = |d_i - dref_i|_2 / max(|dref_i|_2, (|p_i|_2 + body) / 8)          must be <= 1e-5

i.e. 1e-5 relative wherever the vector is longer than 1/8 of the coordinate scale, and an absolute
1e-5 * (|p| + body) / 8 (~ 20 ulp of the coordinates) below that.  The plain relative error (floor 1e-2 mm) is reported
next to it.  (Until late in round 2 the floor was |p| / 8: fine for the bench cloud, but a randomised campaign
(tools/stress_tol.py --tilt 2.5) found 1.3e-5 at a point 22 mm from the BODY origin -- 200 mm from the leg -- where the
floor ignored the 184 mm translation every implementation rounds through; measured over random legs and orientations the
error is at most 10 ulp of |p| + body, and the metric above at most 4.6e-6.)
"""
import numpy as np

TOL = 1.0e-5


def body_of(leg):
    """LegDimensions.body (mm): field 1 of the 14 floats"""
    return float(np.asarray(leg, np.float64).reshape(-1)[1])


def field_error(points, d, dref, leg):
    points = np.asarray(points, np.float64).reshape(-1, 3)
    d = np.asarray(d, np.float64).reshape(-1, 3)
    dref = np.asarray(dref, np.float64).reshape(-1, 3)
    err = np.linalg.norm(d - dref, axis=1)
    nref = np.linalg.norm(dref, axis=1)
    floor = (np.linalg.norm(points, axis=1) + abs(body_of(leg))) / 8.0
    with np.errstate(invalid="ignore", divide="ignore"):
        metric = err / np.maximum(nref, floor)
        plain = err / np.maximum(nref, 1.0e-2)
    bad = ~np.isfinite(d).all(axis=1) & np.isfinite(dref).all(axis=1)
    metric = np.where(bad, np.inf, np.nan_to_num(metric, nan=0.0))
    return dict(metric=metric, abs=err, plain=np.nan_to_num(plain, nan=0.0))


def summary(points, d, dref, leg):
    e = field_error(points, d, dref, leg)
    return dict(max_metric=float(e["metric"].max(initial=0.0)), max_abs_mm=float(np.nan_to_num(e["abs"]).max(initial=0.0)),
                max_plain_rel=float(e["plain"].max(initial=0.0)),
                frac_plain_below_tol=float((e["plain"] <= TOL).mean()) if len(e["plain"]) else 1.0)

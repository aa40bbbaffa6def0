"""Generate the golden fixtures of tests/golden/ from the reference's OWN host path
(oracle/_ref/libref.so = /root/reference sources compiled by oracle/Makefile).

Run in the build container only (the reference does not travel):
    python tests/golden/make_golden.py
Fixtures hold data only: inputs (points, leg parameters, quaternion) and the outputs the
reference computed for them (reachability byte, distance vector, distance validity byte).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.orc import Ref, build  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
QUATS = {
    "id": (1, 0, 0, 0),               # settings.h:51 quatTest
    "y40": (0.924, 0, -0.384, 0),     # settings.h:55 (commented alternative)
    "z40": (0.940, 0, 0, 0.342),      # settings.h:56
    "y20": (0.985, 0, 0.174, 0),      # settings.h:54
    "gen": (0.9, 0.1, 0.2, -0.3),     # non-unit, all components
}


def arange_f32(start, end, step):
    """bench.cpp:21-27 arange: float accumulation, inclusive end."""
    out = []
    v = np.float32(start)
    step = np.float32(step)
    while v <= np.float32(end):
        out.append(v)
        v = np.float32(v + step)
    return np.array(out, np.float32)


def grid(xs, ys, zs):
    """bench.cpp:30-50 generate3DGrid: x-major, z fastest, AoS."""
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    return np.ascontiguousarray(g, np.float32)


def fk(leg, coxa, femur, tibia):
    """Tip position for joint angles, WITH the coxa pitch (the reference's
    forward_kinematics one_leg.cu:377-402 ignores it); float64 then rounded."""
    (body_angle, body, pitch, coxa_l, tibia_l, femur_l) = [float(v) for v in leg[:6]]
    # femur-plane coordinates
    px = coxa_l + femur_l * np.cos(femur) + tibia_l * np.cos(femur + tibia)
    pz = femur_l * np.sin(femur) + tibia_l * np.sin(femur + tibia)
    # undo coxa yaw
    x = px * np.cos(coxa)
    y = px * np.sin(coxa)
    z = pz
    # place_over_coxa rotates (x,z) by -pitch after x -= body: invert it
    c, s = np.cos(-pitch), np.sin(-pitch)
    # forward: x' = x*c - z*s ; z' = x*s + z*c  => inverse is the transpose
    xb = x * c + z * s
    zb = -x * s + z * c
    xb = xb + body
    # body_angle rotation (make_asif_leg0 rotates by -body_angle): invert
    cb, sb = np.cos(body_angle), np.sin(body_angle)
    return np.stack([xb * cb - y * sb, xb * sb + y * cb, zb], -1).astype(np.float32)


def boundary_points(leg, rng, n=4000):
    """Tips with one joint saturated (or the absolute tibia limit active), pushed in/out by
    tiny amounts: these hug every arc of the reachable boundary."""
    pitch = float(leg[2])
    cmax, cmin = float(leg[8]), float(leg[9])
    tmax, tmin = float(leg[10]), float(leg[11])
    fmax, fmin = float(leg[12]), float(leg[13])
    apos, aneg = float(leg[6]), float(leg[7])
    c = rng.uniform(cmin, cmax, n)
    f = rng.uniform(fmin, fmax, n)
    t = rng.uniform(tmin, tmax, n)
    which = rng.integers(0, 7, n)
    f = np.where(which == 0, fmin, f)
    f = np.where(which == 1, fmax, f)
    t = np.where(which == 2, tmin, t)
    t = np.where(which == 3, tmax, t)
    t = np.where(which == 4, apos - f, t)
    t = np.where(which == 5, aneg - f, t)
    c = np.where(which == 6, np.where(rng.random(n) < 0.5, cmin, cmax), c)
    base = fk(leg, c, f, t)
    delta = rng.choice([0.0, 1e-4, 5e-4, 9e-4, 1.1e-3, 2e-3, 1e-2, 0.1, 1.0], n) * rng.choice([-1, 1], n)
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    del pitch
    return (base + (delta[:, None] * direction)).astype(np.float32)


def special_points():
    vals = [0.0, -0.0, 1e-30, -1e-30, 181.0, 181.0 + 65.5, 100.0, -100.0, 246.5, 400.0, 510.5, 600.0]
    pts = [(x, y, z) for x in vals for y in (0.0, -0.0, 50.0, -50.0, 1e-20) for z in (0.0, -0.0, -100.0, 100.0, -250.0)]
    pts += [(300, 0, -100), (400, 0, -200), (200, -120, -250), (250, 50, -150), (500, 0, 0)]  # SURVEY 8c probes
    pts += [(-300, 10, -50), (-100, -200, 0), (0, 0, 0), (181, 0, 0), (181, 0, -65.5), (1e6, 0, 0), (-1e6, 5, 5)]
    return np.array(pts, np.float32)


def case(ref, name, pts, leg, quat):
    pts = np.ascontiguousarray(pts, np.float32)
    mask = ref.reach(pts, leg, quat)
    d, v = ref.dist(pts, leg, quat)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, points=pts, leg=np.asarray(leg, np.float32), quat=np.asarray(quat, np.float32),
                        mask=mask, dist=d, valid=v)
    print(f"{name}: n={len(pts)} reachable={mask.mean():.4f} size={os.path.getsize(path) / 1e6:.2f} MB")


def main():
    build()
    ref = Ref()
    rng = np.random.default_rng(42)
    legs = {"m2": ref.get_M2_leg, "moonbot": ref.get_moonbot_leg}
    # the reference's own leg tables for several azimuths
    np.savez(os.path.join(OUT, "legs.npz"),
             **{f"{k}_{i}": f(np.float32(a)) for k, f in legs.items()
                for i, a in enumerate([0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi / 3, -2.0])},
             azimuths=np.array([0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi / 3, -2.0], np.float32))
    # bench grid, as committed (z from XMin: bench.cpp:114) and as intended (z from ZMin)
    g_bench = grid(arange_f32(-100, 601, 5.12), arange_f32(0, 0, 5.12), arange_f32(-100, 51, 5.12))
    g_full = grid(arange_f32(-100, 601, 4.0), arange_f32(0, 0, 4.0), arange_f32(-350, 51, 4.0))
    lo = np.array([-200, -500, -500], np.float32)
    hi = np.array([700, 500, 300], np.float32)
    cube = (rng.random((20000, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    for lname, lf in legs.items():
        leg0 = lf(np.float32(0.0))
        case(ref, f"grid_bench_{lname}_id", g_bench, leg0, QUATS["id"])
        case(ref, f"grid_full_{lname}_id", g_full, leg0, QUATS["id"])
        for qn, q in QUATS.items():
            case(ref, f"cube_{lname}_az0_{qn}", cube, leg0, q)
        leg1 = lf(np.float32(np.pi / 3))
        case(ref, f"cube_{lname}_az60_y20", cube, leg1, QUATS["y20"])
        case(ref, f"cube_{lname}_az60_gen", cube, leg1, QUATS["gen"])
        bp = boundary_points(leg0, rng)
        case(ref, f"boundary_{lname}_id", bp, leg0, QUATS["id"])
        case(ref, f"special_{lname}_id", special_points(), leg0, QUATS["id"])
        case(ref, f"special_{lname}_y40", special_points(), leg0, QUATS["y40"])
    # rotate_leg_data table
    rl = []
    for lname, lf in legs.items():
        for az in (0.0, np.pi / 3, -2.0):
            leg = lf(np.float32(az))
            for q in QUATS.values():
                rl.append(np.concatenate([leg, np.asarray(q, np.float32), ref.rotate_leg_data(q, leg)]))
    np.save(os.path.join(OUT, "rotate_leg_data.npy"), np.array(rl, np.float32))


if __name__ == "__main__":
    main()

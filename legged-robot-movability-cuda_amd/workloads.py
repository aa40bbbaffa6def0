"""Synthetic workloads of the benchmark configurations (BASELINE.json `configs`).

* config 1/2: the reference's bench grid (bench.cpp:109-120) and the uniform-random cloud.
* config 3: a cratered, fractal-noise terrain cloud and a body-pose lattice above it, in the spirit of
  the reference's maps.py:190-297 / before.py:24-61 (256x256 ground samples over 4 m x 8 m, spherical
  craters and rocks, fractal noise, bodies on a 50 mm lattice from the ground up to +350 mm).  This is
  this repository's own generator (value noise, numpy Generator seeded 42): it reproduces the
  statistics that matter to the kernels (point spacing, relief, body/ground clearances), not the
  reference's exact samples.
"""
import numpy as np


def bench_grid(pix, z_from_xmin=True):
    """bench.cpp:109-120: x in [-100, 601], y = 0, z in [XMin (sic) or ZMin, 51], float accumulation."""
    def arange(start, end, step):
        out, v, step = [], np.float32(start), np.float32(step)
        while v <= np.float32(end):
            out.append(v)
            v = np.float32(v + step)
        return np.array(out, np.float32)
    xs, ys = arange(-100, 601, pix), arange(0, 0, pix)
    zs = arange(-100 if z_from_xmin else -350, 51, pix)
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    return np.ascontiguousarray(g, np.float32)


def random_cloud(n, seed=42):
    """config 2: uniform in the leg's bounding cube [-200,700] x [-500,500] x [-500,300] mm."""
    rng = np.random.default_rng(seed)
    lo = np.array([-200, -500, -500], np.float32)
    hi = np.array([700, 500, 300], np.float32)
    return (rng.random((n, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)


def _value_noise(shape, res, rng):
    """Bilinear-smoothstep value noise on a (res+1) lattice, in [-1, 1]."""
    lat = rng.uniform(-1, 1, (res[0] + 1, res[1] + 1))
    u = np.linspace(0, res[0], shape[0], endpoint=False)
    v = np.linspace(0, res[1], shape[1], endpoint=False)
    iu, iv = u.astype(int), v.astype(int)
    fu, fv = u - iu, v - iv
    su, sv = fu * fu * (3 - 2 * fu), fv * fv * (3 - 2 * fv)
    a = lat[iu][:, iv]
    b = lat[iu + 1][:, iv]
    c = lat[iu][:, iv + 1]
    d = lat[iu + 1][:, iv + 1]
    top = a + (b - a) * su[:, None]
    bot = c + (d - c) * su[:, None]
    return top + (bot - top) * sv[None, :]


def _fractal(shape, res, octaves, persistence, rng):
    out = np.zeros(shape)
    amp, f = 1.0, 1
    for _ in range(octaves):
        out += amp * _value_noise(shape, (res[0] * f, res[1] * f), rng)
        amp *= persistence
        f *= 2
    return out


def terrain(n_side=256, seed=42, extent=((-2000.0, 2000.0), (-6000.0, 2000.0))):
    """(n_side^2, 3) float32 ground cloud: craters/rocks clipped from spheres + fractal relief."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(extent[0][0], extent[0][1], n_side)
    ys = np.linspace(extent[1][0], extent[1][1], n_side)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    Z = np.zeros_like(X)

    def clip(cx, cy, cz, r, down):
        d2 = r * r - (X - cx) ** 2 - (Y - cy) ** 2
        inside = d2 > 0
        cap = np.sqrt(np.where(inside, d2, 0.0))
        if down:  # crater: ground pushed below the lower cap
            np.minimum(Z, np.where(inside, cz - cap, np.inf), out=Z)
        else:     # rock: ground lifted onto the upper cap
            np.maximum(Z, np.where(inside, cz + cap, -np.inf), out=Z)

    for _ in range(50):
        cx, cy = rng.uniform(-2000, 2000, 2)
        clip(cx, cy, rng.uniform(-400, -100), rng.uniform(200, 500), False)
    clip(-2000, -3000, 300, 3000, True)    # crater
    clip(2000, 4000, -800, 4000, False)    # cliff
    clip(1500, 0, -150, 1000, False)       # big rock
    clip(1500, -1000, -150, 700, False)    # small rock
    Z += 300 * _fractal(Z.shape, (4, 8), 5, 0.35, rng)
    np.minimum(Z, 1000, out=Z)
    Z += 30 * _fractal(Z.shape, (16, 32), 3, 0.2, rng)
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1).astype(np.float32)


def body_lattice(ground, n_bodies, voxel=50.0, clearance=(0.0, 350.0), seed=42):
    """Body positions on a `voxel` lattice over the terrain footprint, z from the local ground
    to +350 mm (before.py:24-61 keeps the whole bounding box; only the band that can matter is
    generated here), subsampled to n_bodies."""
    rng = np.random.default_rng(seed)
    side = int(round(np.sqrt(len(ground))))
    gz = ground[:, 2].reshape(side, side)
    x0, x1 = ground[:, 0].min(), ground[:, 0].max()
    y0, y1 = ground[:, 1].min(), ground[:, 1].max()
    xs = np.arange(x0, x1, voxel)
    ys = np.arange(y0, y1, voxel)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    ix = np.clip(((X - x0) / (x1 - x0) * (side - 1)).round().astype(int), 0, side - 1)
    iy = np.clip(((Y - y0) / (y1 - y0) * (side - 1)).round().astype(int), 0, side - 1)
    base = gz[iy, ix]
    layers = np.arange(clearance[0], clearance[1] + 1e-6, voxel)
    pts = np.stack([np.repeat(X.ravel(), len(layers)), np.repeat(Y.ravel(), len(layers)),
                    (base.ravel()[:, None] + layers[None, :]).ravel()], -1).astype(np.float32)
    if n_bodies is not None and n_bodies < len(pts):
        pts = pts[np.sort(rng.choice(len(pts), n_bodies, replace=False))]
    return np.ascontiguousarray(pts)


def hexapod(leg_factory, n_legs=6):
    """n_legs copies of one leg mounted every 2 pi / n_legs (several_leg.cpp:40-47 does 4)."""
    return np.stack([leg_factory(np.float32(2 * np.pi * k / n_legs)) for k in range(n_legs)])


def reference_sweep_quats():
    """The 45 orientations of robot_full_struct (several_leg.cu:814-857): roll, pitch in
    {-pi/8, 0, pi/8}, yaw in {0, pi/8, pi/4, 3pi/8, pi/2}, quat = yaw * pitch * roll * init, built with
    the reference's own quatFromVectAngle / qtMultiply (unified_math_cuda.cu.h:40-57), float32."""
    f = np.float32

    def from_axis(axis, angle):
        s, c = f(np.sin(f(angle) / f(2), dtype=f)), f(np.cos(f(angle) / f(2), dtype=f))
        ax = np.asarray(axis, f)
        mag = f(np.sqrt(f(ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2]))
        return np.array([s, c * ax[0] / mag, c * ax[1] / mag, c * ax[2] / mag], f)

    def mul(a, b):  # (x, y, z, w) with w the scalar part, as qtMultiply
        w = f(f(f(a[3] * b[3] - a[0] * b[0]) - a[1] * b[1]) - a[2] * b[2])
        x = f(f(f(a[3] * b[0] + a[0] * b[3]) + a[1] * b[2]) - a[2] * b[1])
        y = f(f(f(a[3] * b[1] - a[0] * b[2]) + a[1] * b[3]) + a[2] * b[0])
        z = f(f(f(a[3] * b[2] + a[0] * b[1]) - a[1] * b[0]) + a[2] * b[3])
        return np.array([x, y, z, w], f)

    pi = f(3.14159265358979323846)
    init = from_axis((0, 0, 1), 0.0)
    out = []
    for r in range(3):
        roll = f(-pi / 8 + (pi / 8 - -pi / 8) * (f(r) / f(2)))
        q_roll = mul(from_axis((1, 0, 0), roll), init)
        for p in range(3):
            pitch = f(-pi / 8 + (pi / 8 - -pi / 8) * (f(p) / f(2)))
            q_pitch = mul(from_axis((0, 1, 0), pitch), q_roll)
            for y in range(5):
                yaw = f(0 + (pi / 2 - 0) * (f(y) / f(4)))
                out.append(mul(from_axis((0, 0, 1), yaw), q_pitch))
    return np.array(out, f)

"""Workload generators (lrm_amd/workloads.py): the bench grid follows bench.cpp's float
accumulation exactly; the terrain / body lattice reproduce the statistics of the reference's
maps.py / before.py inputs (recorded below from a run of the reference's own scripts in the build
container: maps.ground is (65536, 3) float32, x in [-2000, 2000], y in [-6000, 2000],
z in [-2805.0, 1025.0], std(z) = 1137.9), not its exact samples."""
import numpy as np


def test_bench_grid_sizes_match_the_reference_sweep():
    from lrm_amd import workloads
    g = workloads.bench_grid(1.0)
    assert g.shape == (702 * 152, 3) and g.dtype == np.float32  # Pix = 1.0 -> 106 704 points
    assert g[:, 1].max() == 0 and g[:, 0].min() == -100 and g[:, 2].min() == -100
    # x-major, z fastest (bench.cpp:41-47)
    assert g[1, 0] == g[0, 0] and g[1, 2] > g[0, 2]
    gi = workloads.bench_grid(5.12, z_from_xmin=False)
    assert gi[:, 2].min() == -350
    fix = np.load("tests/golden/grid_bench_m2_id.npz")["points"]
    assert np.array_equal(workloads.bench_grid(5.12), fix)


def test_terrain_statistics_and_determinism():
    from lrm_amd import workloads
    g = workloads.terrain(256)
    assert g.shape == (65536, 3) and g.dtype == np.float32
    assert (g[:, 0].min(), g[:, 0].max(), g[:, 1].min(), g[:, 1].max()) == (-2000, 2000, -6000, 2000)
    assert -3200 < g[:, 2].min() < -2400 and 900 < g[:, 2].max() < 1100
    assert 1000 < g[:, 2].std() < 1300                      # reference: 1137.9
    assert np.array_equal(g, workloads.terrain(256))        # seeded
    b = workloads.body_lattice(g, 5000)
    assert b.shape == (5000, 3)
    # bodies sit between the local ground and +350 mm
    side = 256
    ix = np.clip(((b[:, 0] + 2000) / 4000 * (side - 1)).round().astype(int), 0, side - 1)
    iy = np.clip(((b[:, 1] + 6000) / 8000 * (side - 1)).round().astype(int), 0, side - 1)
    clearance = b[:, 2] - g[:, 2].reshape(side, side)[iy, ix]
    assert clearance.min() > -1e-3 and clearance.max() < 350 + 1e-3


def test_reference_terrain_fixture_pins():
    """tests/golden/terrain_ground.npz holds the OUTPUT of the reference's own maps.py (made by
    tests/golden/make_terrain.py in the build container): the pins SURVEY.md section 8(c) lists."""
    import zlib
    from conftest import reference_terrain
    t = reference_terrain()
    g = t["ground"]
    assert g.shape == (65536, 3) and g.dtype == np.float32
    assert (g[:, 0].min(), g[:, 0].max(), g[:, 1].min(), g[:, 1].max()) == (-2000, 2000, -6000, 2000)
    assert abs(g[:, 2].min() - -2805.0) < 0.5 and abs(g[:, 2].max() - 1025.0) < 0.5
    assert abs(float(g[1000, 2]) - -25.3596) < 1e-4
    assert zlib.crc32(g.tobytes()) == 1234698559
    b = t["bodies"]
    assert b.shape == (89600, 3) and b.dtype == np.float32 and zlib.crc32(b.tobytes()) == 2829093085
    # before.py:24-37: 50 mm lattice over the bounding box, z up to max + 350
    assert (len(t["lattice_x"]), len(t["lattice_y"]), len(t["lattice_z"])) == (80, 160, 84)
    assert np.allclose(np.diff(t["lattice_x"]), 50) and np.allclose(np.diff(t["lattice_z"]), 50, atol=1e-3)
    # the bodies are lattice nodes standing 0..350 mm above the nearest ground sample
    side = 256
    ix = np.clip(np.rint((b[:, 0] + 2000) / 4000 * (side - 1)).astype(int), 0, side - 1)
    iy = np.clip(np.rint((b[:, 1] + 6000) / 8000 * (side - 1)).astype(int), 0, side - 1)
    clearance = b[:, 2] - g[:, 2].reshape(side, side)[iy, ix]
    assert clearance.min() >= -1e-3 and clearance.max() <= 350 + 1e-3
    # this repository's own generator (scale-out sizes) keeps the same footprint and relief statistics
    from lrm_amd import workloads
    own = workloads.terrain(256)
    assert abs(own[:, 2].std() - g[:, 2].std()) < 0.15 * g[:, 2].std()


def test_reference_sweep_quaternions():
    from lrm_amd import workloads
    q = workloads.reference_sweep_quats()
    assert q.shape == (45, 4) and q.dtype == np.float32
    assert np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-6)
    # first orientation: roll = pitch = -pi/8, yaw = 0 ; last: +pi/8, +pi/8, pi/2
    assert not np.array_equal(q[0], q[-1])


def test_morton_order_is_a_permutation_with_locality(lrm):
    from lrm_amd import workloads
    g = workloads.terrain(64)
    o = lrm.morton_order(g)
    assert np.array_equal(np.sort(o), np.arange(len(g)))
    s = g[o]
    # consecutive points are close: 64-point chunks are compact compared with raster rows
    ext = lambda a: np.mean([np.ptp(a[i:i + 64, :2], axis=0).max() for i in range(0, len(a), 64)])
    assert ext(s) < 0.5 * ext(g)

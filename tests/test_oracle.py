"""The oracle against (a) the fixtures generated from the reference's own host path,
(b) that host path itself where it is built, (c) the behavioural properties the
reference's stale Catch2 file one_leg.cpp describes."""
import numpy as np
import pytest

from conftest import bits_equal, golden_cases, load_case, random_cloud


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference_fixture(oracle, name):
    c = load_case(name)
    m = oracle.reach(c["points"], c["leg"], c["quat"])
    d, v = oracle.dist(c["points"], c["leg"], c["quat"])
    assert np.array_equal(m, c["mask"])
    assert np.array_equal(v, c["valid"])
    assert bits_equal(d, c["dist"]).all()


def test_oracle_leg_tables(oracle):
    legs = dict(np.load("tests/golden/legs.npz"))
    az = legs.pop("azimuths")
    for i, a in enumerate(az):
        assert oracle.get_M2_leg(a).tobytes() == legs[f"m2_{i}"].tobytes()
        assert oracle.get_moonbot_leg(a).tobytes() == legs[f"moonbot_{i}"].tobytes()
    m2 = oracle.get_M2_leg(0.0)
    assert m2.nbytes == 56  # HeaderCPP.h:19-52
    # static_variables.cpp:69-93: body 181, coxa 65.5, femur 129, tibia 135
    assert (m2[1], m2[3], m2[5], m2[4]) == (181.0, 65.5, 129.0, 135.0)


def test_oracle_rotate_leg_data(oracle):
    tab = np.load("tests/golden/rotate_leg_data.npy")
    for row in tab:
        leg, q, want = row[:14], row[14:18], row[18:]
        assert oracle.rotate_leg_data(q, leg).tobytes() == want.tobytes()


def test_oracle_equals_reference_build_on_random_points(oracle, ref):
    pts = random_cloud(100000, seed=7)
    for legf_o, legf_r in ((oracle.get_M2_leg, ref.get_M2_leg), (oracle.get_moonbot_leg, ref.get_moonbot_leg)):
        for az in (0.0, 2.5):
            leg = legf_r(az)
            assert leg.tobytes() == legf_o(az).tobytes()
            for q in ((1, 0, 0, 0), (0.996, 0, -0.087, 0), (0.7, -0.2, 0.1, 0.6)):
                assert np.array_equal(oracle.reach(pts, leg, q), ref.reach(pts, leg, q))
                d0, v0 = oracle.dist(pts, leg, q)
                d1, v1 = ref.dist(pts, leg, q)
                assert np.array_equal(v0, v1) and bits_equal(d0, d1).all()
    # the reference's own CPU loops (compiled-in quatTest = identity)
    leg = ref.get_M2_leg(0.0)
    assert np.array_equal(ref.reach_kernel_cpu(pts, leg), oracle.reach(pts, leg))
    assert bits_equal(ref.dist_kernel_cpu(pts, leg), oracle.dist(pts, leg)[0]).all()


# ---- properties (one_leg.cpp ideas, re-expressed for the absolute-tibia-limit model) ----
def _fk(leg, c, f, t):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("mk", os.path.join("tests", "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk.fk(leg, c, f, t)


@pytest.mark.parametrize("legname", ["get_M2_leg", "get_moonbot_leg"])
def test_fk_samples_inside_limits_are_reachable(oracle, legname):
    """one_leg.cpp:141-202: joint samples strictly inside every limit must be reachable."""
    leg = getattr(oracle, legname)(0.0)
    n = 24
    c = np.linspace(leg[9], leg[8], n + 2)[1:-1]
    f = np.linspace(leg[13], leg[12], n + 2)[1:-1]
    t = np.linspace(leg[11], leg[10], n + 2)[1:-1]
    C, F, T = [a.ravel() for a in np.meshgrid(c, f, t, indexing="ij")]
    ok = (F + T < leg[6] - 1e-3) & (F + T > leg[7] + 1e-3)
    pts = _fk(leg, C[ok], F[ok], T[ok])
    m = oracle.reach(pts, leg)
    assert ok.sum() > 5000 and m.all()


@pytest.mark.parametrize("legname", ["get_M2_leg", "get_moonbot_leg"])
def test_overlong_tibia_is_unreachable_and_distance_matches(oracle, legname):
    """one_leg.cpp:204-402, :590-828: a tip pushed out of the outer circle by delta is
    unreachable and |distance| ~= delta."""
    leg = getattr(oracle, legname)(0.0)
    rng = np.random.default_rng(3)
    n = 2000
    c = rng.uniform(leg[9] * 0.9, leg[8] * 0.9, n)
    # fully stretched leg (tibia = 0) inside the femur / absolute-tibia window
    lo = max(leg[13], leg[7]) + 0.05
    hi = min(leg[12], leg[6]) - 0.05
    f = rng.uniform(lo, hi, n)
    for delta in (1.0, 0.1):
        longer = leg.copy()
        longer[4] += delta  # tibia_length
        pts = _fk(longer, c, f, np.zeros(n))
        assert not oracle.reach(pts, leg).any()
        d, v = oracle.dist(pts, leg)
        assert not v.any()
        norm = np.linalg.norm(d.astype(np.float64), axis=1)
        assert np.allclose(norm, delta, rtol=2e-3, atol=2e-4)


def test_distance_bool_equals_reachability(oracle):
    pts = random_cloud(200000, seed=11)
    for legf in (oracle.get_M2_leg, oracle.get_moonbot_leg):
        leg = legf(0.0)
        assert np.array_equal(oracle.dist(pts, leg)[1], oracle.reach(pts, leg))


def test_mirror_symmetry_in_y(oracle):
    """The leg is symmetric about its own xz plane (coxa limits are +-60 deg)."""
    pts = random_cloud(100000, seed=5)
    mir = pts * np.array([1, -1, 1], np.float32)
    leg = oracle.get_M2_leg(0.0)
    a, b = oracle.reach(pts, leg), oracle.reach(mir, leg)
    # decisions can only differ on points within float rounding of a boundary
    assert (a != b).mean() < 1e-4


def test_known_answers_in_the_leg_plane(oracle):
    """one_leg.cpp:100-139, :498-588 (manual points) for the pitch-free moonbot leg, along a
    stretched direction 30 degrees below the horizon (inside the tibia window; the horizontal
    direction itself violates tibia_absolute_pos = -5 deg): just outside the outer radius is
    unreachable, just inside is reachable, and 1 mm beyond gives a 1 mm vector along it."""
    leg = oracle.get_moonbot_leg(0.0)
    L = float(leg[4] + leg[5])
    a = np.deg2rad(-30.0)
    u = np.array([np.cos(a), 0.0, np.sin(a)])
    base = np.array([leg[1] + leg[3], 0.0, 0.0])
    pts = np.array([base + (L + 0.01) * u, base + (L + 1.0) * u, base + (L - 0.01) * u, base + (L - 1.0) * u],
                   np.float32)
    assert list(oracle.reach(pts, leg)) == [0, 0, 1, 1]
    d, _ = oracle.dist(pts[1:2], leg)
    assert np.allclose(d[0], u, atol=2e-3)
    far = np.array([[1e4, 0, 0], [-500, 0, 0], [0, 0, 1e4]], np.float32)
    assert not oracle.reach(far, leg).any()

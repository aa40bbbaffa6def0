"""The bit-exact table-guided evaluation (csrc/lrm_point_xtab.h: DECISIONS from the plane table of the tolerance mode,
VALUES in the reference's operation order) compiled for the CPU, WITHOUT the re-evaluation of its doubtful points,
against the oracle: every point the evaluation does not flag must carry the oracle's reach mask, validity byte and all
three floats of the distance vector BIT FOR BIT (tolerance 0); the flagged fraction must stay small (those points go
through the filtered code of LRM_MODE_FAST in a second launch on the GPU)."""
import numpy as np
import pytest

from conftest import bits_equal, golden_cases, load_case, random_cloud

QUATS = [(1, 0, 0, 0), (0.9848, 0, 0.1736, 0), (0.9397, 0, 0, 0.342), (0.9, 0.1, 0.2, -0.3)]


def check(lrm, pts, leg, quat, want_mask, want_valid, want_dist, max_doubt):
    m, d, doubt, stats = lrm.dbg_xtab_host(pts, leg, quat)
    sure = (doubt & 0xffff) == 0
    assert np.array_equal(m[sure], want_mask[sure]), "reach mask differs on points the table path calls certain"
    assert np.array_equal(m[sure], want_valid[sure]), "validity byte differs on points the table path calls certain"
    same = bits_equal(d[sure], want_dist[sure]).all(axis=1)
    bad = np.flatnonzero(~same)
    assert len(bad) == 0, (f"{len(bad)} of {int(sure.sum())} certain vectors are not bit-identical; first: point "
                           f"{pts[sure][bad[0]]}, got {d[sure][bad[0]]}, want {want_dist[sure][bad[0]]}")
    assert 1.0 - sure.mean() <= max_doubt, f"{1.0 - sure.mean():.4f} of the points are in doubt"
    return 1.0 - sure.mean(), stats


@pytest.mark.parametrize("name", golden_cases("cube") + golden_cases("grid"))
def test_xtab_matches_reference_fixture(lrm, name):
    c = load_case(name)
    if not lrm.dbg_tol_ok(c["leg"], c["quat"]):
        pytest.skip("leg not eligible for the table-guided modes (the library then uses the filtered kernels)")
    # the planar bench grids lie on the symmetry plane of the symmetric legs and contain the coxa axis: more doubt than a cloud
    check(lrm, c["points"], c["leg"], c["quat"], c["mask"], c["valid"], c["dist"], 0.09)


@pytest.mark.parametrize("name", golden_cases("boundary") + golden_cases("special"))
def test_xtab_on_boundary_hugging_points(lrm, name):
    """Points constructed ON the decision boundaries (and signed zeros, on-axis points): most are in doubt by design; the others must be right."""
    c = load_case(name)
    if not lrm.dbg_tol_ok(c["leg"], c["quat"]):
        pytest.skip("leg not eligible")
    check(lrm, c["points"], c["leg"], c["quat"], c["mask"], c["valid"], c["dist"], 1.0)


@pytest.mark.parametrize("legname", ["m2", "moonbot"])
@pytest.mark.parametrize("az", [0.0, np.pi / 3, -2.0])
def test_xtab_random_cloud_vs_oracle(lrm, oracle, legname, az):
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    pts = random_cloud(200_000, seed=7)
    for q in QUATS:
        want_d, want_v = oracle.dist(pts, leg, q)
        frac, stats = check(lrm, pts, leg, q, oracle.reach(pts, leg, q), want_v, want_d, 0.02)
        # the twin of an invalid direct candidate in front of the coxa + the rare real second candidates
        assert stats["second_chains"] <= 0.35 * len(pts)  # the twin of an invalid direct candidate in front of the coxa


@pytest.mark.parametrize("shift", [900.0, 4000.0])
def test_xtab_far_clouds_use_the_outer_grid(lrm, oracle, shift):
    leg = lrm.get_M2_leg(0.0)
    pts = random_cloud(100_000, seed=17)
    pts[:, 0] += np.float32(shift)
    want_d, want_v = oracle.dist(pts, leg)
    check(lrm, pts, leg, None, oracle.reach(pts, leg), want_v, want_d, 0.05)


def test_xtab_near_the_coxa_axis(lrm, oracle):
    """No conditioning guard around the coxa axis (the tolerance mode sends r < 16 mm to the fix-up): the values are the
    reference's own operations, only the yaw-sector decisions can be in doubt there."""
    leg = lrm.get_M2_leg(0.0)
    rng = np.random.default_rng(5)
    n = 100_000
    pts = np.empty((n, 3), np.float32)
    # the coxa axis in the body frame: x = body (181 mm) along the pitched z axis
    t = rng.uniform(-300, 300, n)
    pitch = np.deg2rad(-45.0)
    pts[:, 0] = 181.0 + t * np.sin(pitch) + rng.normal(0, 4.0, n)
    pts[:, 1] = rng.normal(0, 4.0, n)
    pts[:, 2] = t * np.cos(pitch) + rng.normal(0, 4.0, n)
    want_d, want_v = oracle.dist(pts, leg)
    check(lrm, pts, leg, None, oracle.reach(pts, leg), want_v, want_d, 0.10)


def test_xtab_nonfinite_inputs_are_in_doubt(lrm):
    pts = np.array([[np.nan, 0, 0], [np.inf, 1, 2], [1e30, 1e30, -1e30], [0, 0, 0], [181.0, 0.0, 0.0]], np.float32)
    _, _, doubt, _ = lrm.dbg_xtab_host(pts, lrm.get_M2_leg(0.0))
    assert (doubt[:3] != 0).all()


@pytest.mark.parametrize("legname,az", [("m2", 0.0), ("moonbot", np.pi / 3), ("m2", -2.0)])
def test_replay_of_the_tolerance_decisions_is_bit_exact(lrm, oracle, legname, az):
    """LRM_MODE_TOL_REL's short-vector path: the tolerance evaluation takes the decisions (clamp target, candidate, yaw-limit
    alternative), the fix-up replays the winner's value chain with the reference's own operations (lrm_xtab_replay) -- no table, no
    bands.  For EVERY point the tolerance evaluation does not doubt (not only the short vectors) the replay must give the oracle's
    vector bit for bit."""
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    pts = random_cloud(200_000, seed=9)
    for q in QUATS:
        m, d, doubt = lrm.dbg_replay_host(pts, leg, q)
        want_d, want_v = oracle.dist(pts, leg, q)
        sure = (doubt & 0xffff) == 0
        assert np.array_equal(m[sure], oracle.reach(pts, leg, q)[sure])
        same = bits_equal(d[sure], want_d[sure]).all(axis=1)
        bad = np.flatnonzero(~same)
        assert len(bad) == 0, (f"{len(bad)} of {int(sure.sum())} replayed vectors are not bit-identical; first: point {pts[sure][bad[0]]}, "
                               f"got {d[sure][bad[0]]}, want {want_d[sure][bad[0]]}")
        assert sure.mean() > 0.98


@pytest.mark.parametrize("name", golden_cases("cube") + golden_cases("grid") + golden_cases("boundary") + golden_cases("special"))
def test_replay_on_the_reference_fixtures(lrm, name):
    c = load_case(name)
    if not lrm.dbg_tol_ok(c["leg"], c["quat"]):
        pytest.skip("leg not eligible")
    m, d, doubt = lrm.dbg_replay_host(c["points"], c["leg"], c["quat"])
    sure = (doubt & 0xffff) == 0
    assert np.array_equal(m[sure], c["mask"][sure])
    assert bits_equal(d[sure], c["dist"][sure]).all()

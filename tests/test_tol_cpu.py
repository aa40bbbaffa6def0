"""LRM_MODE_TOL on the host (lrm_dbg_tol_host = csrc/lrm_point_tol.h compiled for the CPU, WITHOUT the bit-exact
re-evaluation of the doubtful points) against the oracle: every point the evaluation does not flag must have the
oracle's mask bit for bit and a distance vector inside the tolerance (tests/tolcheck.py); the flagged fraction
must stay small (those points cost a second pass on the GPU)."""
import numpy as np
import pytest

from conftest import golden_cases, load_case, random_cloud
from tolcheck import TOL, field_error

QUATS = [(1, 0, 0, 0), (0.9848, 0, 0.1736, 0), (0.9397, 0, 0, 0.342), (0.9, 0.1, 0.2, -0.3)]


def check(lrm, pts, leg, quat, want_mask, want_valid, want_dist, max_doubt):
    m, d, doubt = lrm.dbg_tol_host(pts, leg, quat)
    sure = (doubt & 0xffff) == 0
    assert np.array_equal(m[sure], want_mask[sure]), "reach mask differs on points the filter calls certain"
    assert np.array_equal(m[sure], want_valid[sure]), "validity byte differs on points the filter calls certain"
    e = field_error(pts[sure], d[sure], want_dist[sure], leg)
    assert e["metric"].max(initial=0.0) <= TOL, f"distance error {e['metric'].max():.3e} (abs {e['abs'].max():.3e} mm)"
    assert 1.0 - sure.mean() <= max_doubt, f"{1.0 - sure.mean():.4f} of the points are in doubt"
    return 1.0 - sure.mean()


@pytest.mark.parametrize("name", golden_cases("cube") + golden_cases("grid"))
def test_tol_matches_reference_fixture(lrm, name):
    c = load_case(name)
    if not lrm.dbg_tol_ok(c["leg"], c["quat"]):
        pytest.skip("leg not eligible for the tolerance mode (the library then uses LRM_MODE_FAST)")
    # the planar bench grids (y = 0) contain the coxa axis (4-5 % of their points are within LRM_TOL_RMIN = 16 mm of it) and
    # lie on the symmetry plane of the symmetric legs, where the two yaw-limit planes tie exactly (3 % for the
    # moonbot leg): more doubt than a cloud
    check(lrm, c["points"], c["leg"], c["quat"], c["mask"], c["valid"], c["dist"], 0.09)


@pytest.mark.parametrize("name", golden_cases("boundary") + golden_cases("special"))
def test_tol_on_boundary_hugging_points(lrm, name):
    """Points constructed ON the decision boundaries: most are in doubt by design; the others must be right."""
    c = load_case(name)
    if not lrm.dbg_tol_ok(c["leg"], c["quat"]):
        pytest.skip("leg not eligible")
    check(lrm, c["points"], c["leg"], c["quat"], c["mask"], c["valid"], c["dist"], 1.0)


@pytest.mark.parametrize("legname", ["m2", "moonbot"])
@pytest.mark.parametrize("az", [0.0, np.pi / 3, -2.0])
def test_tol_random_cloud_vs_oracle(lrm, oracle, legname, az):
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    pts = random_cloud(200_000, seed=7)
    for q in QUATS:
        assert lrm.dbg_tol_ok(leg, q), "the two robots of the reference must be eligible in every orientation tested"
        want_d, want_v = oracle.dist(pts, leg, q)
        check(lrm, pts, leg, q, oracle.reach(pts, leg, q), want_v, want_d, 0.012)


def test_tol_eligibility_of_odd_legs(lrm):
    """Legs the filters cannot take (yaw limits at +-90 deg) are refused, never mis-evaluated."""
    odd = lrm.leg_factory(0.0, 181, -45, 65.5, 129, 135, 90.0, 90.0, 120.0, -5, -5)
    assert not lrm.dbg_tol_ok(odd)
    with pytest.raises(lrm.LrmError):
        lrm.dbg_tol_host(random_cloud(10), odd)


def test_tol_nonfinite_inputs_are_in_doubt(lrm):
    pts = np.array([[np.nan, 0, 0], [np.inf, 1, 2], [1e30, 1e30, -1e30], [0, 0, 0], [181.0, 0.0, 0.0]], np.float32)
    _, _, doubt = lrm.dbg_tol_host(pts, lrm.get_M2_leg(0.0))
    assert (doubt[:3] != 0).all()


@pytest.mark.parametrize("legname,az,q", [("m2", 0.0, QUATS[0]), ("moonbot", np.pi / 3, QUATS[1]), ("m2", -2.0, QUATS[3])])
@pytest.mark.parametrize("shift", [0.0, 900.0, 4000.0])
def test_tol_plane_table_vs_oracle(lrm, oracle, legname, az, q, shift):
    """The plane table with deferred decisions (csrc/lrm_toltab.cpp) replaces the full plane evaluation: a cell names at
    most two clamp targets and one open validity, everything else is decided per cell on the host.  Every point the
    table path resolves (no doubt bit, no 0x100 "unanswered cell") must carry the oracle's mask and a vector inside the
    tolerance; unanswered points (sent to the bit-exact fix-up on the GPU) must stay a small fraction -- on the inner
    grid (the config-2 cube) and on the outer one (the cube shifted beyond +-1024 mm of the femur joint, and 4 m out: the outer grid is
    built for the larger decision bands of far points)."""
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    pts = random_cloud(200_000, seed=17)
    pts[:, 0] += np.float32(shift)
    m, d, doubt, stats = lrm.dbg_toltab_host(pts, leg, q)
    want_d, want_v = oracle.dist(pts, leg, q)
    sure = (doubt & 0xffff) == 0
    assert np.array_equal(m[sure], oracle.reach(pts, leg, q)[sure]) and np.array_equal(m[sure], want_v[sure])
    e = field_error(pts[sure], d[sure], want_d[sure], leg)
    assert e["metric"].max(initial=0.0) <= TOL
    assert 4 <= stats["rows"] <= 31 and 2 <= stats["vrows"] <= 31 and stats["refined"] > 0 and stats["bytes"] < 2_000_000
    assert ((doubt & 0x100) != 0).mean() < (0.02 if shift < 2000 else 0.05)
    assert sure.mean() > (0.97 if shift < 2000 else 0.95)
    # same decisions as the full evaluation wherever both are certain (the arithmetic differs in the last bits only)
    m2, d2, doubt2 = lrm.dbg_tol_host(pts, leg, q)
    both = sure & ((doubt2 & 0xffff) == 0)
    assert np.array_equal(m[both], m2[both])
    assert (np.abs(d[both] - d2[both]) <= 1e-3 + 1e-6 * np.linalg.norm(d2[both], axis=1, keepdims=True)).all()  # (a few ulp of the vector's LENGTH in every component where it is metres long)


@pytest.mark.parametrize("legname,az,q", [("m2", 0.0, QUATS[0]), ("moonbot", np.pi / 3, QUATS[1]), ("m2", -2.0, QUATS[3])])
def test_plane_table_lower_bounds_hold(lrm, legname, az, q):
    """Every coarse cell of the table carries a lower bound of the in-plane distance sqrt(du^2 + dz^2) of its plane points
    (0 where a point may be valid).  The per-point code orders the two yaw candidates by w^2 + bound^2, takes the reach flag
    from the first one and skips the second when the first one's distance is below the other's bound: the bound must never
    exceed what the full plane evaluation finds (points whose full evaluation is itself in doubt go to the bit-exact
    code whatever the table said about them -- unless they were skipped, so they are held to the bound as well, with
    the width of a doubt band as allowance)."""
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    rng = np.random.default_rng(23)
    for half in (1000.0, 8000.0):  # inner and outer grid
        xz = rng.uniform(-half, half, (400_000, 2)).astype(np.float32)
        lb, dist, valid, doubt = lrm.dbg_toltab_bounds(xz, leg, q)
        assert np.isfinite(lb).all() and (lb >= 0).all()
        assert (lb[valid != 0] == 0).all()
        slack = np.where(doubt == 0, 1e-5 * dist + 1e-4, 0.05)
        assert (lb <= dist + slack).all(), float((lb - dist).max())
        # and it is worth having: within 1.5 cell diagonals of the distance for nearly all invalid points of the inner grid
        if half == 1000.0:
            inv = (valid == 0) & (doubt == 0)
            assert ((dist - lb)[inv] < 35.0).mean() > 0.95


def test_second_candidate_is_rarely_evaluated(lrm):
    """with the cells' lower bounds the second yaw candidate of a config-2 point is evaluated for well under 1 % of the points
    (36 % with the bound of the round-2 kernels: every wave of a random cloud then ran it)"""
    pts = random_cloud(200_000, seed=29)
    _, _, _, stats = lrm.dbg_toltab_host(pts, lrm.get_M2_leg(0.0), QUATS[0])
    assert stats["second_candidates"] < 0.004 * len(pts), stats


def test_plane_table_does_not_depend_on_the_builder_threads(lrm, monkeypatch):
    """the table's rows are classified on several host threads and numbered afterwards, serially: one thread or eight, the
    same table (same statistics, same answers, same doubt bits).  (The builder is also clean under -fsanitize=thread and
    -fsanitize=address,undefined on the CPU build.)"""
    pts = random_cloud(50_000, seed=5)
    leg = lrm.get_M2_leg(0.9)
    q = QUATS[2]
    monkeypatch.setenv("LRM_TOLTAB_THREADS", "1")
    m1, d1, b1, s1 = lrm.dbg_toltab_host(pts, leg, q)
    monkeypatch.setenv("LRM_TOLTAB_THREADS", "7")
    m2, d2, b2, s2 = lrm.dbg_toltab_host(pts, leg, q)
    assert s1 == s2 and np.array_equal(m1, m2) and np.array_equal(b1, b2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))

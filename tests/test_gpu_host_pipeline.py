"""LRM_HOST_PIPELINE=1 (run with -m gpu): the host-buffer entry points of the apply_kernel boundary (lrm_reach, lrm_dist,
lrm_reach_dist; cross_compiled.cu:33-79) with chunked H2D || kernels || D2H on three streams.  The same kernels run on
sub-ranges, so every output byte must equal the unpipelined call's -- on the reference fixtures, on ragged sizes that
leave a short last chunk, in the bit-exact and in the tolerance mode."""
import numpy as np
import pytest

from conftest import bits_equal, golden_cases, load_case, random_cloud

pytestmark = pytest.mark.gpu


def both_ways(monkeypatch, fn, chunk):
    monkeypatch.setenv("LRM_HOST_PIPELINE", "0")
    plain = fn()
    monkeypatch.setenv("LRM_HOST_PIPELINE", "1")
    monkeypatch.setenv("LRM_HOST_PIPELINE_CHUNK", str(chunk))
    piped = fn()
    return plain, piped


@pytest.mark.parametrize("name", golden_cases("cube")[:4] + golden_cases("grid")[:2])
def test_pipeline_equals_plain_on_reference_fixtures(lrm, name, monkeypatch):
    c = load_case(name)
    pts, leg, q = c["points"], c["leg"], c["quat"]
    (m0, _), (m1, ms1) = both_ways(monkeypatch, lambda: lrm.apply_reach(pts, leg, q), 4096)
    assert np.array_equal(m0, m1) and np.array_equal(m1, c["mask"]) and ms1 > 0
    (d0, v0, _), (d1, v1, _) = both_ways(monkeypatch, lambda: lrm.apply_dist(pts, leg, q), 4096)
    assert bits_equal(d0, d1).all() and np.array_equal(v0, v1) and bits_equal(d1, c["dist"]).all()
    (m0, d0, _), (m1, d1, _) = both_ways(monkeypatch, lambda: lrm.apply_reach_dist(pts, leg, q), 8192)
    assert np.array_equal(m0, m1) and bits_equal(d0, d1).all()


@pytest.mark.parametrize("n,chunk", [(8192, 4096), (8193, 4096), (100_003, 4096), (100_003, 65536), (3_000_001, 1 << 20)])
@pytest.mark.parametrize("mode", ["fast", "tol"])
def test_pipeline_ragged_sizes(lrm, n, chunk, mode, monkeypatch):
    lrm.set_mode(lrm.MODE_TOL if mode == "tol" else lrm.MODE_FAST)
    try:
        pts = random_cloud(n, seed=n % 1000)
        leg = lrm.get_moonbot_leg(0.7)
        q = (0.98, 0.1, -0.1, 0.05)
        (m0, d0, _), (m1, d1, ms) = both_ways(monkeypatch, lambda: lrm.apply_reach_dist(pts, leg, q), chunk)
        assert np.array_equal(m0, m1)
        if mode == "fast":
            assert bits_equal(d0, d1).all()
        else:  # the tolerance mode's main kernel is chosen per launch size (table from 2e5 points on): same contract, maybe other last bits
            from tolcheck import TOL, field_error
            assert field_error(pts, d1, d0, leg)["metric"].max() <= 2 * TOL
        assert ms > 0
    finally:
        lrm.set_mode(lrm.MODE_FAST)

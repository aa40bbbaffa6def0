"""The plane table built on the device (csrc/lrm_toltab_dev.hip) against the host builder (csrc/lrm_toltab.cpp): the same bytes.
Both run the per-cell arithmetic of csrc/lrm_toltab_build.h (double: +, -, *, /, sqrt only, fixed summation orders); the host
builder's tables are what tests/test_tol_cpu.py and tests/test_xtab_cpu.py check against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QUATS = [None, (0.9848, 0, 0.1736, 0), (0.9397, 0, 0, 0.342), (0.9, 0.1, 0.2, -0.3)]


@pytest.mark.parametrize("legname,az", [("m2", 0.0), ("m2", -2.0), ("moonbot", 0.0), ("moonbot", np.pi / 3)])
def test_device_built_table_equals_host_built_table(lrm, legname, az):
    leg = lrm.get_M2_leg(az) if legname == "m2" else lrm.get_moonbot_leg(az)
    for q in QUATS:
        host, ms_host = lrm.dbg_toltab_build(leg, q, device=False)
        dev, ms_dev = lrm.dbg_toltab_build(leg, q, device=True)
        dev2, ms_dev2 = lrm.dbg_toltab_build(leg, q, device=True)  # (the first call allocates the builder's scratch)
        assert host.size == dev.size, (host.size, dev.size)
        diff = np.flatnonzero(host != dev)
        assert diff.size == 0, f"{diff.size} bytes differ, first at {diff[0]} (header is 1600 bytes)"
        assert np.array_equal(dev, dev2)
        print(f"{legname} {az:.2f} {q}: {host.size} bytes, host {ms_host:.1f} ms, device {ms_dev2:.3f} ms")


def test_device_builder_on_random_legs(lrm):
    """random leg geometries and joint limits (as tools/stress_tol.py draws them), random orientations"""
    rng = np.random.default_rng(11)
    done = 0
    for _ in range(24):
        leg = lrm.leg_factory(float(rng.uniform(-3.1, 3.1)), float(rng.uniform(60, 260)), float(rng.uniform(-60, 20)),
                              float(rng.uniform(30, 110)), float(rng.uniform(90, 180)), float(rng.uniform(90, 200)),
                              float(rng.uniform(25, 85)), float(rng.uniform(50, 100)), float(rng.uniform(80, 140)),
                              float(rng.uniform(-15, 10)), float(rng.uniform(-15, 10)))
        q = rng.normal(size=4) * np.array([1.0, 0.25, 0.25, 0.35])
        q[0] = abs(q[0]) + 0.8
        q = (q / np.linalg.norm(q)).astype(np.float32)
        if not lrm.dbg_tol_ok(leg, q):
            continue
        try:
            host, _ = lrm.dbg_toltab_build(leg, q, device=False)
        except lrm.LrmError:
            continue  # more rows than a cell code can name: no table for this leg
        dev, _ = lrm.dbg_toltab_build(leg, q, device=True)
        assert np.array_equal(host, dev)
        done += 1
    assert done >= 12


def test_first_call_builds_its_table_on_the_device(lrm, oracle):
    """the product path: a tolerance-mode call on a new (leg, orientation) builds its table with the device builder; the results
    are those of the host-built table (LRM_TOLTAB_HOST=1) bit for bit, and the build is reported in milliseconds"""
    import os
    import torch
    from conftest import random_cloud
    pts = random_cloud(400_000, seed=31)
    t = torch.from_numpy(np.ascontiguousarray(pts.T)).cuda()
    leg = lrm.get_M2_leg(0.77)
    q = (0.97, 0.1, -0.2, 0.05)
    lrm.set_mode(lrm.MODE_TOL)
    try:
        lrm.release_workspaces()
        m1, d1 = lrm.device.reach_dist(t[0], t[1], t[2], leg, q)
        torch.cuda.synchronize()
        ms_dev = lrm.last_table_build_ms()
        lrm.release_workspaces()
        os.environ["LRM_TOLTAB_HOST"] = "1"
        try:
            m2, d2 = lrm.device.reach_dist(t[0], t[1], t[2], leg, q)
            torch.cuda.synchronize()
            ms_host = lrm.last_table_build_ms()
        finally:
            del os.environ["LRM_TOLTAB_HOST"]
            lrm.release_workspaces()
    finally:
        lrm.set_mode(lrm.MODE_FAST)
    assert torch.equal(m1, m2) and torch.equal(d1.view(torch.int32), d2.view(torch.int32))
    assert np.array_equal(m1.cpu().numpy(), oracle.reach(pts, leg, q))
    print(f"table build: device {ms_dev:.3f} ms, host {ms_host:.1f} ms")
    assert 0 < ms_dev < ms_host

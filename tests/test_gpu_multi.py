"""lrm_reach_dist_multi (run with -m gpu): the single-process multi-device entry of the C ABI.  The build box has one
GPU: ndev = 1 goes through the very same code path (shards, streams, pack kernel, gather, copies back), once with the
plain device copy and once with the RCCL calls forced (LRM_MULTI_FORCE_RCCL=1: dlopen, ncclCommInitAll, grouped
ncclAllGather with one rank).  ndev > 1 is unmeasured on hardware."""
import os

import numpy as np
import pytest

from conftest import bits_equal, random_cloud
from tolcheck import TOL, field_error

pytestmark = pytest.mark.gpu


def packed(mask):
    n = len(mask)
    return np.packbits(np.pad(mask, (0, (-n) % 64)), bitorder="little").view(np.uint64)


@pytest.mark.parametrize("force_rccl", ["0", "1"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4099, 300_003])
def test_multi_one_device_equals_reach_dist(lrm, n, force_rccl, monkeypatch):
    monkeypatch.setenv("LRM_MULTI_FORCE_RCCL", force_rccl)
    pts = random_cloud(n, seed=n + 3)
    leg = lrm.get_M2_leg(0.4)
    q = (0.97, 0.05, -0.2, 0.1)
    want_m, want_d, _ = lrm.apply_reach_dist(pts, leg, q)
    m, d, bits, ms = lrm.apply_reach_dist_multi(pts, leg, q, ndev=1)
    assert np.array_equal(m, want_m)
    assert bits_equal(d, want_d).all()
    assert np.array_equal(bits, packed(want_m))
    assert ms.shape == (1,) and ms[0] > 0
    lrm.lib().lrm_multi_release()


def test_multi_tolerance_mode_and_oracle(lrm, oracle, monkeypatch):
    """the mode applies to the shards' kernels: mask exact, field inside the tolerance of the oracle"""
    monkeypatch.setenv("LRM_MULTI_FORCE_RCCL", "1")
    pts = random_cloud(400_000, seed=8)
    leg = lrm.get_moonbot_leg(0.0)
    lrm.set_mode(lrm.MODE_TOL)
    try:
        m, d, bits, _ = lrm.apply_reach_dist_multi(pts, leg, None, ndev=1, devices=[0])
    finally:
        lrm.set_mode(lrm.MODE_FAST)
        lrm.lib().lrm_multi_release()
    want_m = oracle.reach(pts, leg)
    want_d, _ = oracle.dist(pts, leg)
    assert np.array_equal(m, want_m) and np.array_equal(bits, packed(want_m))
    assert field_error(pts, d, want_d, leg)["metric"].max() <= TOL


def test_multi_argument_errors(lrm):
    pts = random_cloud(100, seed=1)
    leg = lrm.get_M2_leg(0.0)
    ndev = lrm.device_count()
    with pytest.raises(lrm.LrmError):
        lrm.apply_reach_dist_multi(pts, leg, None, ndev=0)
    with pytest.raises(lrm.LrmError):
        lrm.apply_reach_dist_multi(pts, leg, None, ndev=1, devices=[ndev])      # ordinal out of range
    with pytest.raises(lrm.LrmError):
        lrm.apply_reach_dist_multi(pts, leg, None, ndev=ndev + 1)               # more devices than the box has


@pytest.mark.parametrize("mode", ["tol", "tol_rel", "fast"])
def test_multi_calls_do_not_leak_queue_workspaces(lrm, mode):
    """lrm_reach_dist_multi creates and destroys a stream per device and call; the table-guided modes key their queue workspace by
    (device, stream): the entry must go with the stream (round 3 kept one 60 MB workspace per call for ever)."""
    import torch
    pts = random_cloud(2_000_000, seed=12)
    leg = lrm.get_M2_leg(0.0)
    lrm.set_mode({"tol": lrm.MODE_TOL, "tol_rel": lrm.MODE_TOL_REL, "fast": lrm.MODE_FAST}[mode])
    try:
        for _ in range(2):
            lrm.apply_reach_dist_multi(pts, leg, None, ndev=1)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(12):
            lrm.apply_reach_dist_multi(pts, leg, None, ndev=1)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
    finally:
        lrm.set_mode(lrm.MODE_FAST)
        lrm.lib().lrm_multi_release()
    assert free0 - free1 < (16 << 20), f"{(free0 - free1) >> 20} MiB of device memory went missing over 12 calls"

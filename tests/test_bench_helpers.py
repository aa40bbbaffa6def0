"""CPU checks of bench.py's host-side helpers (the timed path itself needs a GPU)."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_chunk_seeded_shards_are_slices_of_one_cloud():
    """N > 1: every rank generates only its shard; the shards of any world size must be slices of the SAME cloud."""
    from lrm_amd import shard
    b = _bench()
    n = 2_500_000  # 2.5 chunks
    whole = b.make_shard(0, n)
    assert whole.shape == (3, n) and whole.dtype == np.float32
    assert (whole.min(axis=1) >= b.LO).all() and (whole.max(axis=1) <= b.HI).all()
    for world in (2, 3, 8):
        parts = [b.make_shard(*shard.shard_bounds(n, world, r)) for r in range(world)]
        assert np.array_equal(np.concatenate(parts, axis=1), whole)
    # config 2 (N = 1) keeps the cloud of the tests and of the committed profiles
    from conftest import random_cloud
    assert np.array_equal(b.make_cloud(100_000, 42).T, random_cloud(100_000, seed=42))


def test_committed_profile_lookup_is_tied_to_the_kernel_sources(tmp_path):
    """bench.py prints the PMC figures of profiles/ only when they were taken with the tree's kernel sources
    (kernel_src_sha); anything else is reported as stale, never mixed in."""
    from lrm_amd.srchash import kernel_src_sha
    b = _bench()
    sha = kernel_src_sha()
    assert len(sha) == 64 and sha == kernel_src_sha()
    json.dump({"points_per_launch": 10_000_000, "mode": "tol", "step_hbm_bytes": 2.7e8, "kernel_src_sha": sha}, open(tmp_path / "r09_hbm_traffic.json", "w"))
    json.dump({"mode": "tol", "valu_insts_per_eval": 400.0, "kernel_src_sha": sha}, open(tmp_path / "r09_valu.json", "w"))
    json.dump({"points_per_launch": 10_000_000, "mode": "fast", "step_hbm_bytes": 2.5e8, "kernel_src_sha": "0" * 64}, open(tmp_path / "r09fast_hbm_traffic.json", "w"))
    p = b.committed_profile(10_000_000, "tol", str(tmp_path))
    assert p["traffic"] == 2.7e8 and p["traffic_source"] == "r09_hbm_traffic.json" and p["valu_insts_per_eval"] == 400.0 and p["kernel_src_sha"] == sha
    f = b.committed_profile(10_000_000, "fast", str(tmp_path))
    assert f["traffic"] is None and f["traffic_source"].startswith("stale") and f["valu_insts_per_eval"] is None
    assert b.committed_profile(12345, "tol", str(tmp_path))["traffic"] is None
    # the repository's own profiles: whatever matches the tree must be plausible
    p = b.committed_profile(10_000_000, "tol")
    if p["traffic"] is not None:
        assert 250e6 < p["traffic"] < 320e6          # algorithmic 251.25 MB + the fix-up's scattered accesses
    if p["valu_insts_per_eval"] is not None:
        assert 300 < p["valu_insts_per_eval"] < 800

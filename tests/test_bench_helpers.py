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


def test_committed_profile_lookup():
    b = _bench()
    p = b.committed_profile(10_000_000, "tol")
    tr = json.load(open(os.path.join(ROOT, "profiles", p["traffic_source"])))
    assert tr["mode"] == "tol" and tr["points_per_launch"] == 10_000_000
    assert 250e6 < p["traffic"] < 320e6          # algorithmic 251.25 MB + the fix-up's scattered accesses
    assert 300 < p["valu_insts_per_eval"] < 800
    f = b.committed_profile(10_000_000, "fast")
    assert f["traffic"] is not None and f["traffic"] < 253e6   # the bit-exact fused kernel has no wasted traffic
    assert b.committed_profile(12345, "tol")["traffic"] is None

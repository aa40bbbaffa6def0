"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the sharding / gather logic.  Each
rank evaluates its contiguous slice with the library's explicit CPU entry point
(lrm_reach_cpu: the GPU kernels cannot run here) and the all-gathered bit mask must equal the
single-process mask of the whole cloud."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import lrm_amd
    from conftest import random_cloud
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts = random_cloud(n, seed=5)
        leg = lrm_amd.get_M2_leg(0.2)
        quat = (0.97, 0.0, 0.2, 0.1)

        def local_bits(lo, hi):
            mask, _ = lrm_amd.apply_reach_cpu(pts[lo:hi], leg, quat)
            return torch.from_numpy(lrm_amd.shard.pack_bits(mask).copy())

        words = lrm_amd.shard.reach_bits_sharded(local_bits, n)
        full, _ = lrm_amd.apply_reach_cpu(pts, leg, quat)
        got = lrm_amd.shard.unpack_bits(words.numpy(), n)
        ok = bool(np.array_equal(got, full)) and words.numel() == (n + 63) // 64
        # per-body bytes (aggregation results) gather
        lo, hi = lrm_amd.shard.shard_bounds(n, world, rank)
        by = lrm_amd.shard.all_gather_bytes(torch.from_numpy(full[lo:hi].copy()), n)
        ok = ok and bool(np.array_equal(by.numpy(), full))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 100000), (2, 64), (3, 1000), (2, 65), (4, 100)])
def test_sharded_reach_matches_single_process(world, n):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() + world * 7 + n) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    assert len(ret) == world and all(ret.values())


def test_shard_bounds_cover_and_align():
    from lrm_amd import shard
    for n in (0, 1, 63, 64, 65, 1000, 10**7 + 3):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard.shard_bounds(n, world, r)
                assert lo == prev or lo == n
                assert lo % 64 == 0 or lo == n
                prev = hi
            assert prev == n
    m = (np.arange(200) % 3 == 0).astype(np.uint8)
    assert np.array_equal(shard.unpack_bits(shard.pack_bits(m), 200), m)

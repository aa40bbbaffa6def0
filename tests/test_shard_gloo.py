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


# ---- the bench step loop (BitsGatherLoop), body-sharded positionability, target-sharded any-flags ------------
class _CpuBackend:
    """CPU stand-ins for the three GPU operations the sharded drivers compose: float32 numpy spheres, and the
    oracle's brute-force reach_any (test infrastructure) for the per-body answers.  The drivers' job -- slicing,
    the MAX all-reduce of the far-target cull, the gathers -- is what these tests check: world 2 / 3 must equal
    the same composition run in a single process."""

    def __init__(self):
        from oracle.orc import Oracle
        self.o = Oracle()

    def any_in_sphere(self, centres, targets, radius):
        c = np.asarray(centres, np.float32).reshape(-1, 3)
        t = np.asarray(targets, np.float32).reshape(-1, 3)
        out = np.zeros(len(c), np.uint8)
        for i in range(0, len(c), 256):
            d = c[i:i + 256, None, :] - t[None, :, :]
            out[i:i + 256] = ((d * d).sum(-1, dtype=np.float32) < np.float32(radius) ** 2).any(1)
        return out

    def positionability(self, bodies, targets, legs, quats, culls):
        acc = np.zeros(len(bodies), np.uint8)
        if len(targets) == 0:
            return acc
        for q in np.asarray(quats, np.float32).reshape(-1, 4):
            acc |= self.o.reach_any(bodies, targets, legs, q).min(axis=0)
        return acc

    def reach_any(self, bodies, targets, legs, quat):
        if len(targets) == 0:
            return np.zeros((len(legs), len(bodies)), np.uint8)
        return self.o.reach_any(bodies, targets, legs, (1, 0, 0, 0) if quat is None else quat)


def _scene(nb, nt, seed):
    rng = np.random.default_rng(seed)
    txy = rng.uniform(-900, 900, (nt, 2))
    targets = np.column_stack([txy, 40 * np.sin(txy[:, 0] / 150) + rng.normal(0, 5, nt)]).astype(np.float32)
    bodies = np.column_stack([rng.uniform(-1500, 1500, (nb, 2)), rng.uniform(40, 330, nb)]).astype(np.float32)
    return bodies, targets


def _worker2(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import lrm_amd
    from lrm_amd import shard
    from conftest import random_cloud
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # 1. the step loop: ONE n-point cloud, rank r owns shard_bounds(n, world, r); every step's gathered words
        #    must equal the single-process words (a different leg per step: the two buffers must not mix)
        pts = random_cloud(n, seed=9)
        loop = shard.BitsGatherLoop(n, device="cpu")
        assert (loop.lo, loop.hi) == shard.shard_bounds(n, world, rank)
        for k in range(5):
            leg = lrm_amd.get_M2_leg(0.1 * k)

            def compute(words, lo, hi, leg=leg):
                mask, _ = lrm_amd.apply_reach_cpu(pts[lo:hi], leg)
                words.copy_(torch.from_numpy(shard.pack_bits(mask).copy()))

            b = loop.step(k, compute)
            full, _ = lrm_amd.apply_reach_cpu(pts, leg)
            got = loop.result(b)
            ok = ok and got.numel() == (n + 63) // 64 and bool(np.array_equal(shard.unpack_bits(got.numpy(), n), full))
        # 2. body-sharded positionability, with and without the reference's culls
        be = _CpuBackend()
        bodies, targets = _scene(max(n // 200, 3), 400, seed=3)
        legs = np.stack([lrm_amd.get_M2_leg(2 * np.pi * l / 4) for l in range(4)])
        quats = np.array([[1, 0, 0, 0], [0.98, 0.0, 0.17, 0.0]], np.float32)
        for culls in (False, True):
            got = shard.positionability_sharded(bodies, targets, legs, quats, culls, backend=be)
            ret[(rank, "pos", culls)] = got.tobytes()
        # 3. target-sharded any-flags: each rank holds a slice of the cloud
        tlo, thi = shard.shard_bounds(len(targets), world, rank, align=1)
        out, all_legs = shard.reach_any_target_sharded(bodies, targets[tlo:thi], legs, None, backend=be)
        want = be.reach_any(bodies, targets, legs, None)
        ok = ok and bool(np.array_equal(out, want)) and bool(np.array_equal(all_legs, want.min(axis=0)))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 20000), (3, 1000), (2, 65), (3, 130)])
def test_step_loop_and_sharded_positionability(world, n):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import lrm_amd
    from lrm_amd import shard
    port = 31500 + (os.getpid() + world * 11 + n) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker2, args=(world, port, n, ret), nprocs=world, join=True)
    assert all(ret[r] is True for r in range(world))
    # every rank returns the mask of ALL bodies, equal to the single-process composition of the same steps
    be = _CpuBackend()
    bodies, targets = _scene(max(n // 200, 3), 400, seed=3)
    legs = np.stack([lrm_amd.get_M2_leg(2 * np.pi * l / 4) for l in range(4)])
    quats = np.array([[1, 0, 0, 0], [0.98, 0.0, 0.17, 0.0]], np.float32)
    for culls in (False, True):
        want = shard.positionability_sharded(bodies, targets, legs, quats, culls, backend=be)  # no process group here
        for r in range(world):
            assert ret[(r, "pos", culls)] == want.tobytes()
    assert 0 < shard.positionability_sharded(bodies, targets, legs, quats, False, backend=be).sum() or n < 1000

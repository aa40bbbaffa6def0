"""GPU parity of the level-synchronous octree (csrc/lrm_octree.hip) against the brute-force
restatement tests/octree_oracle.py.  Parity is pinned by composition only (see that file)."""
import numpy as np
import pytest

from octree_oracle import apply_oct as oracle_apply_oct

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["strict", "fast"])
def mode(request, lrm):
    lrm.set_mode(lrm.MODE_FAST if request.param == "fast" else lrm.MODE_STRICT)
    yield request.param
    lrm.set_mode(lrm.MODE_FAST)


def footholds(n, seed, spread=600.0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-spread, spread, (n, 2))
    z = 20 * np.sin(xy[:, 0] / 120) + rng.normal(0, 4, n) - 150
    return np.column_stack([xy, z]).astype(np.float32)


def settings(lrm, half, depth, stab=4, legs=4, mounts=None, rot_below=50.0):
    st = lrm.octree_default_settings()
    for i in range(3):
        st.box_size[i] = half
    st.max_depth = depth
    st.leg_number_for_stab = stab
    st.leg_count = legs
    st.enable_rot_below = rot_below
    if mounts is not None:
        for i, m in enumerate(mounts):
            st.leg_mount[i] = m
    return st


@pytest.mark.parametrize("half,depth,stab,rot_below,deep", [
    (5000.0, 1, 4, 50.0, False),   # the committed configuration (settings.h): one level, rotations never active
    (400.0, 4, 3, 50.0, True),     # down to boxes below MINBOXSIZE: dead quadrants and unsplittable leaves
    (400.0, 4, 4, 50.0, True),     # LegNumberForStab = LegCount as committed: deep tree, nothing valid
    (800.0, 3, 2, 50.0, True),
    (300.0, 3, 1, 400.0, False),   # small root with the 27 orientation samples active from the first level
])
def test_apply_oct_matches_bruteforce(lrm, oracle, half, depth, stab, rot_below, deep):
    f = footholds(160, seed=404)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, half, depth, stab=stab, rot_below=rot_below)
    got, ms = lrm.apply_oct(f, dim, st)
    want, n_nodes = oracle_apply_oct(oracle, f, dim, st)
    assert ms >= 0
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if deep:
        assert n_nodes > 200  # the refinement actually happened
    if stab == 3:
        assert len(want) > 10


def test_apply_oct_two_legs_many_valid_leaves(lrm, oracle):
    f = footholds(160, seed=404)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, 400.0, 5, stab=2, legs=2, mounts=(0.0, 0.3))
    got, _ = lrm.apply_oct(f, dim, st)
    want, n_nodes = oracle_apply_oct(oracle, f, dim, st)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and len(want) > 100 and n_nodes > 500


def test_apply_oct_defaults_and_capacity(lrm, oracle):
    f = footholds(64, seed=5)
    dim = lrm.get_moonbot_leg(0.0)
    got, _ = lrm.apply_oct(f, dim)  # settings.h as committed
    want, _ = oracle_apply_oct(oracle, f, dim, lrm.octree_default_settings())
    assert np.array_equal(got, want)
    # capacity too small -> error code and required size
    import ctypes as C
    st = settings(lrm, 800.0, 3, stab=1)
    full, _ = lrm.apply_oct(f, dim, st)
    if len(full) > 1:
        out = np.zeros((1, 3), np.float32)
        n_out = C.c_size_t(0)
        rc = lrm.lib().lrm_apply_oct(f.ctypes.data_as(C.c_void_p), len(f), dim.ctypes.data_as(C.c_void_p), C.addressof(st),
                                     out.ctypes.data_as(C.c_void_p), 1, C.addressof(n_out), None)
        assert rc == -1 and n_out.value == len(full)
    # no footholds: nothing is valid
    empty, _ = lrm.apply_oct(np.zeros((0, 3), np.float32), dim, st)
    assert len(empty) == 0


def test_apply_oct_at_scale_chunked_equals_every_foothold_kernel(lrm, oracle, mode):
    """1e6 footholds (a 4 m x 4 m relief), depth 5: levels with >= 65 children run the chunk-culled kernel (one
    workgroup per child, only the footholds of nearby 64-foothold chunks).  Its leaves must equal, bit for bit, the
    every-foothold kernel's (LRM_OCT_BRUTE=1: an independent traversal of the same flags) and must not depend on the
    order of the footholds.  (Equality with the brute-force oracle is checked on trees of up to 3 500 nodes above,
    where the deeper levels also run the chunk-culled kernel; the oracle is O(children x footholds) in Python.)"""
    import os
    if mode == "strict":
        pytest.skip("one arithmetic mode is enough at this size")
    rng = np.random.default_rng(77)
    n = 1_000_000
    xy = rng.uniform(-2000, 2000, (n, 2))
    z = 60 * np.sin(xy[:, 0] / 300) * np.cos(xy[:, 1] / 250) + rng.normal(0, 3, n) - 150
    f = np.column_stack([xy, z]).astype(np.float32)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, 2000.0, 5, stab=3)
    got, ms = lrm.apply_oct(f, dim, st)
    os.environ["LRM_OCT_BRUTE"] = "1"
    try:
        brute, ms_brute = lrm.apply_oct(f, dim, st)
    finally:
        del os.environ["LRM_OCT_BRUTE"]
    assert len(got) > 50 and np.array_equal(got.view(np.uint32), brute.view(np.uint32))
    # the bounding-sphere cull in front of the distance evaluations (oct_item_flags) is exact: same leaves without it
    os.environ["LRM_OCT_NOCULL"] = "1"
    try:
        nocull, ms_nocull = lrm.apply_oct(f, dim, st)
    finally:
        del os.environ["LRM_OCT_NOCULL"]
    assert np.array_equal(got.view(np.uint32), nocull.view(np.uint32))
    # footholds already on the device (lrm_apply_oct_dev): the same tree
    import torch
    t = torch.from_numpy(np.ascontiguousarray(f.T)).cuda()
    dev, ms_dev = lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
    assert np.array_equal(got.view(np.uint32), dev.view(np.uint32))
    shuffled, _ = lrm.apply_oct(f[rng.permutation(n)], dim, st)
    assert np.array_equal(got.view(np.uint32), shuffled.view(np.uint32))
    print(f"apply_oct, 1e6 footholds, depth 5: {len(got)} valid leaves; chunk-culled {ms:.2f} ms of kernels, every-foothold {ms_brute:.2f} ms, without the sphere cull {ms_nocull:.2f} ms")


def _oct_worker(rank, world, port, ret):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import lrm_amd
    from test_gpu_octree import footholds, settings
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f = footholds(3000, seed=9, spread=900.0)
        dim = lrm_amd.get_M2_leg(0.0)
        st = settings(lrm_amd, 800.0, 5, stab=3)
        got, _ = lrm_amd.shard.apply_oct_sharded(f, dim, st)
        # the same with the footholds already on the device (lrm_apply_oct_dev through the exchange callback)
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(f.T)).cuda()
        dev, _ = lrm_amd.shard.apply_oct_sharded((t[0], t[1], t[2]), dim, st)
        assert got.tobytes() == dev.tobytes()
        ret[rank] = got.tobytes()
    finally:
        dist.destroy_process_group()


def test_apply_oct_sharded_over_two_ranks_equals_single_process(lrm):
    """The level-sharded octree (lrm_apply_oct_sharded: children dealt round-robin, flags combined with all_reduce MAX
    per level) with two ranks sharing this box's GPU over gloo: both ranks return the single-process leaves."""
    import os
    import torch.multiprocessing as mp
    lrm.set_mode(lrm.MODE_FAST)
    f = footholds(3000, seed=9, spread=900.0)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, 800.0, 5, stab=3)
    want, _ = lrm.apply_oct(f, dim, st)
    assert len(want) > 20
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_oct_worker, args=(2, 30500 + os.getpid() % 1000, ret), nprocs=2, join=True)
    assert ret[0] == want.tobytes() and ret[1] == want.tobytes()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_apply_oct_random_settings_three_traversals_agree(lrm, seed):
    """Random terrains, tree sizes, leg counts, stability numbers, orientation thresholds and leg geometries: the default
    traversal (chunk-culled kernel with several workgroups per large child + the exact bounding-sphere cull), the
    every-foothold kernel (LRM_OCT_BRUTE=1) and the traversal without the sphere cull (LRM_OCT_NOCULL=1) must return
    the same leaves, bit for bit, in the same order; the device-resident entry too."""
    import os
    import torch
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(20_000, 200_000))
    half = float(rng.choice([600.0, 1200.0, 2500.0]))
    xy = rng.uniform(-half, half, (n, 2))
    z = (rng.uniform(20, 150) * np.sin(xy[:, 0] / rng.uniform(100, 600)) * np.cos(xy[:, 1] / rng.uniform(100, 600))
         + rng.normal(0, 4, n) - rng.uniform(80, 260))
    f = np.column_stack([xy, z]).astype(np.float32)
    dim = lrm.get_M2_leg(0.0) if seed % 2 else lrm.get_moonbot_leg(0.0)
    st = lrm.octree_default_settings()
    for i in range(3):
        st.box_size[i] = half
    st.max_depth = int(rng.integers(3, 6))
    st.leg_count = int(rng.integers(2, 5))
    st.leg_number_for_stab = int(rng.integers(1, st.leg_count + 1))
    for i in range(st.leg_count):
        st.leg_mount[i] = float(rng.uniform(-np.pi, np.pi))
    st.enable_rot_below = float(rng.choice([50.0, 200.0, 700.0]))
    st.min_box = float(rng.choice([60.0, 100.0, 150.0]))
    got, _ = lrm.apply_oct(f, dim, st)
    results = {}
    for var in ("LRM_OCT_BRUTE", "LRM_OCT_NOCULL"):
        os.environ[var] = "1"
        try:
            results[var], _ = lrm.apply_oct(f, dim, st)
        finally:
            del os.environ[var]
    t = torch.from_numpy(np.ascontiguousarray(f.T)).cuda()
    results["device"], _ = lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
    for name, r in results.items():
        assert np.array_equal(got.view(np.uint32), r.view(np.uint32)), (name, len(got), len(r))
    print(f"seed {seed}: {n} footholds, half {half}, depth {st.max_depth}, {st.leg_count} legs, stability {st.leg_number_for_stab}: {len(got)} leaves")

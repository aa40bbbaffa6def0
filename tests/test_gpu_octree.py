"""GPU parity of the level-synchronous octree (csrc/lrm_octree.hip) against the brute-force
restatement tests/octree_oracle.py.  Parity is pinned by composition only (see that file)."""
import numpy as np
import pytest

from octree_oracle import apply_oct as oracle_apply_oct

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["strict", "fast", "fast_tol", "fast_tab", "fast_tab_inline"])
def mode(request, lrm, monkeypatch):
    """strict: lrm_point.h verbatim; fast: the filtered code for every work item (LRM_OCT_TOL=0); fast_tol: the contract-tolerance
    evaluation first, the filtered code for its doubts and for vectors that end near a face of the child box (LRM_OCT_TOL=1: whatever
    the cloud's size; the library's default takes it from 3e5 footholds on), without plane tables (LRM_OCT_TAB=0); fast_tab: the same
    through the plane tables of the (orientation, leg) pairs, the doubtful (item, orientation) pairs of levels with 256 children or
    more queued for a second launch (the default); fast_tab_inline: the tables with the doubts redone in place (LRM_OCT_DEFER=0).
    All five must give the same leaves."""
    lrm.set_mode(lrm.MODE_STRICT if request.param == "strict" else lrm.MODE_FAST)
    monkeypatch.setenv("LRM_OCT_TOL", "1" if request.param in ("fast_tol", "fast_tab", "fast_tab_inline") else "0")
    monkeypatch.setenv("LRM_OCT_TAB", "1" if request.param in ("fast_tab", "fast_tab_inline") else "0")
    monkeypatch.setenv("LRM_OCT_DEFER", "0" if request.param == "fast_tab_inline" else "1")
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
        # foothold-partitioned: no rank holds the whole cloud.  A spatial split (the root box's octants) and an arbitrary
        # one (every world-th foothold) must give the same tree: the flags are ORs over footholds.
        own = lrm_amd.shard.octant_owner(f, st.box_center, world)
        part, _ = lrm_amd.shard.apply_oct_partitioned(f[own == rank], dim, st)
        mine = torch.from_numpy(np.ascontiguousarray(f[rank::world].T)).cuda()
        part_dev, _ = lrm_amd.shard.apply_oct_partitioned((mine[0], mine[1], mine[2]), dim, st)
        assert got.tobytes() == part.tobytes() == part_dev.tobytes(), (len(got), len(part), len(part_dev))
        ret[rank] = got.tobytes()
    finally:
        dist.destroy_process_group()


def test_apply_oct_sharded_over_two_ranks_equals_single_process(lrm):
    """The level-sharded octree (lrm_apply_oct_sharded: children dealt round-robin) and the foothold-partitioned one
    (lrm_apply_oct_partitioned: every rank holds a part of the cloud; octant split and an arbitrary split), flags OR-ed
    over the ranks per level, with two ranks sharing this box's GPU over gloo: both ranks return the single-process leaves."""
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


def test_octree_exchange_failures_fail_the_call(lrm):
    """an exception inside the exchange callback must fail the call and come back to the caller (a ctypes callback that
    returned nothing used to swallow it: the library went on with flags that were never combined); a rank that reads
    the failure marker of a peer fails too"""
    from lrm_amd import _capi
    f = footholds(500, seed=3)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, 400.0, 3, stab=3)
    calls = []

    def boom(flags):
        calls.append(len(flags))
        if len(calls) == 2:
            raise RuntimeError("all_reduce failed")

    with pytest.raises(RuntimeError, match="all_reduce failed"):
        _capi.apply_oct_partitioned(f, dim, st, boom)
    assert calls == [1, 8]  # the one-word handshake, then the first level: nothing after the failure

    def poisoned(flags):  # a peer that failed contributes 0xffffffff to every word
        if len(flags) > 1:
            flags[:] = 0xffffffff

    with pytest.raises(lrm.LrmError, match="peer rank failed"):
        _capi.apply_oct_partitioned(f, dim, st, poisoned)
    # an identity exchange (a single rank's OR) gives the plain tree
    want, _ = lrm.apply_oct(f, dim, st)
    got, _ = _capi.apply_oct_partitioned(f, dim, st, lambda flags: None)
    assert got.tobytes() == want.tobytes()


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


def test_deferred_queue_that_overflows_runs_the_level_again_inline(lrm, mode, monkeypatch):
    """The queue of deferred (item, orientation) pairs is sized from the cloud; when it overflows the kernel raises a flag and the
    host evaluates the level again with the doubts in place.  Forced here (64 records, every level deferred): the same leaves."""
    if mode != "fast_tab":
        pytest.skip("the deferred form only")
    f = footholds(6000, seed=31)
    dim = lrm.get_M2_leg(0.0)
    st = settings(lrm, 400.0, 5, stab=3)
    want, _ = lrm.apply_oct(f, dim, st)
    monkeypatch.setenv("LRM_OCT_DEFER_FROM", "1")
    every_level, _ = lrm.apply_oct(f, dim, st)
    monkeypatch.setenv("LRM_OCT_DEFER_CAP", "64")
    overflowed, _ = lrm.apply_oct(f, dim, st)
    assert len(want) > 0
    assert np.array_equal(want.view(np.uint32), every_level.view(np.uint32)) and np.array_equal(want.view(np.uint32), overflowed.view(np.uint32))


def test_config5_share_of_one_gpu_with_an_oracle_sample(lrm, oracle, mode):
    """BASELINE config 5 at ONE GPU's share of the 1e8-point cloud: 1.25e7 footholds on a 10 m x 10 m relief, the
    reference's settings (root box +-5000 mm, min box 100 mm, rotations below 50 mm), 4 legs, stability 3, depth 6.
    The whole tree cannot be restated on the CPU (2e4 children x 1e5-1e7 footholds x 4 legs), so: invariants of the whole
    tree, and for a sample of the children of every level the three flag bits the kernel returned against
    tests/octree_oracle.child_flags over exactly the footholds the reference itself would test for that child (its
    elongated-box cull, several_leg_octree.cu:76-82).  Deep levels (<= 4e5 footholds in the box): the flags must be
    EQUAL.  The first levels (millions of footholds in the box): the oracle runs on a random 2e5 of them, and because
    every flag is an OR over footholds its flags must be a SUBSET of the kernel's."""
    if mode == "strict":
        pytest.skip("one arithmetic mode is enough at this size (the modes are bit-identical)")
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from octree_oracle import child_flags, octree_legs, quat_from_angle_index
    n = 12_500_000
    rng = np.random.default_rng(5)
    xy = rng.uniform(-5000, 5000, (n, 2)).astype(np.float32)
    z = (400 * np.sin(xy[:, 0] / 900) * np.cos(xy[:, 1] / 700) + 60 * np.sin(xy[:, 0] / 130) + rng.normal(0, 5, n).astype(np.float32) - 200).astype(np.float32)
    f = np.column_stack([xy, z]).astype(np.float32)
    del xy, z
    dim = lrm.get_M2_leg(0.0)
    st = lrm.octree_default_settings()
    st.max_depth = 6
    st.leg_number_for_stab = 3
    t = torch.from_numpy(np.ascontiguousarray(f.T)).cuda()
    lrm.dbg_oct_trace(True)
    try:
        leaves, ms = lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
        rec = lrm.dbg_oct_trace_read()
    finally:
        lrm.dbg_oct_trace(False)
    leaves2, _ = lrm.device.apply_oct(t[0], t[1], t[2], dim, st)
    del t
    # ---- the whole tree ----
    assert np.array_equal(leaves.view(np.uint32), leaves2.view(np.uint32)), "two runs, two trees"
    depth = rec[:, 11].astype(int)
    sizes = [int((depth == d).sum()) for d in range(6)]
    assert sizes[0] == 8 and all(s % 8 == 0 and s > 0 for s in sizes), sizes
    assert len(rec) > 15_000 and len(leaves) > 3_000
    assert (np.abs(leaves) <= 5000).all()
    flags = rec[:, 9].astype(int)
    meta = rec[:, 10].astype(int)
    assert not ((meta & 2) != 0).any()  # rotations only below 50 mm boxes: never at 78 mm
    # a child is refined only when it is on an edge: every level's size is 8 x the on-edge (and not leaf) children of the one before
    for d in range(5):
        lv = depth == d
        on_edge = ((flags[lv] & 4) != 0) & ((flags[lv] & 2) == 0) & ((meta[lv] & 4) == 0)
        assert sizes[d + 1] == 8 * int(on_edge.sum()), (d, sizes)
    # ---- a sample of children of every level against the oracle ----
    order = np.argsort(f[:, 0], kind="stable")
    fs = f[order]
    del f, order
    legs = octree_legs(dim, st)
    quats = [quat_from_angle_index(oracle, a, st) for a in range(27)]
    reach_len = np.float32(np.float32(np.float32(dim[1] + dim[3]) + dim[5]) + dim[4])
    pick = np.random.default_rng(11)

    def check(k):
        c, h, ph = rec[k, 0:3], rec[k, 3:6], rec[k, 6:9]
        ext = ph + reach_len
        a, b = np.searchsorted(fs[:, 0], [c[0] - ext[0] - 1, c[0] + ext[0] + 1])
        near = fs[a:b]
        near = near[(np.abs(near[:, 1] - c[1]) <= ext[1] + 1) & (np.abs(near[:, 2] - c[2]) <= ext[2] + 1)]
        exact = len(near) <= 400_000
        if not exact:
            near = near[np.random.default_rng(k).choice(len(near), 200_000, replace=False)]
        want = child_flags(oracle, near, c, h, ph, bool(meta[k] & 1), bool(meta[k] & 2), st, legs, quats, reach_len)
        want_bits = int(want[0]) | (int(want[1]) << 1) | (int(want[2]) << 2)
        return k, exact, want_bits

    todo = []
    for d in range(6):
        cand = np.flatnonzero((depth == d) & ((meta & 4) == 0))
        todo += list(pick.choice(cand, min(16, len(cand)), replace=False))
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(check, todo))
    n_exact = 0
    for k, exact, want_bits in res:
        if exact:
            n_exact += 1
            assert flags[k] == want_bits, (k, depth[k], flags[k], want_bits)
        else:
            assert (want_bits & ~flags[k]) == 0, (k, depth[k], flags[k], want_bits)
    assert n_exact >= 30
    print(f"config-5 share: {n} footholds, depth 6: {len(rec)} children in levels of {sizes}, {len(leaves)} valid leaves, {ms:.0f} ms of kernels; "
          f"{n_exact} sampled children equal to the oracle, {len(res) - n_exact} (first levels) consistent with a foothold sample")

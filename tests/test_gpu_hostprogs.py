"""The C++ host programs on top of the C ABI, run as subprocesses on the GPU box:
lrm_cuda  = the reference's file-to-file driver (several_leg.cpp:124-224, raw float32 SoA files);
lrm_bench = the reference's bench.cpp sweep (CSV rows "N;ns_per_point")."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits_equal, random_cloud

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "legged-robot-movability-cuda_amd", "host")


def _built(name):
    path = os.path.join(HOST, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", HOST], check=True)
    return path


@pytest.mark.parametrize("robot,legname", [(1, "get_M2_leg"), (0, "get_moonbot_leg")])
def test_file_to_file_driver(tmp_path, oracle, robot, legname):
    # before.py:60-99 pattern: a 3-D grid written as one float32 file per component
    pts = random_cloud(50001, seed=8)
    for k, c in enumerate("xyz"):
        pts[:, k].astype(np.float32).tofile(tmp_path / f"dist_input_t{c}.bin")
    out = subprocess.run([_built("lrm_cuda"), str(tmp_path), str(robot)], check=True, capture_output=True, text=True)
    assert "Cuda reachability took" in out.stdout and "Cuda distance took" in out.stdout
    leg = getattr(oracle, legname)(0.0)
    reach = np.fromfile(tmp_path / "out_reachability.bin", np.uint8)
    d = np.stack([np.fromfile(tmp_path / f"out_dist_x{c}.bin", np.float32) for c in "xyz"], -1)
    assert np.array_equal(reach, oracle.reach(pts, leg))
    assert bits_equal(d, oracle.dist(pts, leg)[0]).all()


def test_soa_host_entry_points(lrm, oracle):
    import ctypes as C
    pts = random_cloud(12347, seed=77)
    leg = lrm.get_M2_leg(0.5)
    q = np.array([0.99, 0.0, 0.1, 0.05], np.float32)
    x, y, z = [np.ascontiguousarray(pts[:, k]) for k in range(3)]
    mask = np.zeros(len(pts), np.uint8)
    valid = np.zeros(len(pts), np.uint8)
    d = [np.zeros(len(pts), np.float32) for _ in range(3)]
    ms = C.c_float()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    L = lrm.lib()
    assert L.lrm_reach_soa(P(x), P(y), P(z), len(pts), P(leg), P(q), P(mask), C.addressof(ms)) == 0 and ms.value > 0
    assert L.lrm_dist_soa(P(x), P(y), P(z), len(pts), P(leg), P(q), P(d[0]), P(d[1]), P(d[2]), P(valid), C.addressof(ms)) == 0
    want_d, want_v = oracle.dist(pts, leg, q)
    assert np.array_equal(mask, oracle.reach(pts, leg, q))
    assert np.array_equal(valid, want_v) and bits_equal(np.stack(d, -1), want_d).all()


def test_bench_harness_writes_reference_csv_format(tmp_path):
    """bench.cpp:164-171 rows `N;ns_per_point`, four files, grid sizes of the committed sweep
    (x in [-100,601], y = 0, z in [-100,51] at pitch min_pix * 2^k <= 50)."""
    subprocess.run([_built("lrm_bench"), str(tmp_path), "3.2", "3", "1", "2", "6.4"], check=True, capture_output=True)
    # compute index 4 (bench.cpp:87-91): the RBDL-equivalent IK, own minimum pitch (MinPixRBDL), rbdl.csv
    for name, reps, grids in (("rgpu.csv", 3, 4), ("rcpu.csv", 1, 4), ("dgpu.csv", 3, 4), ("dcpu.csv", 1, 4), ("rbdl.csv", 2, 3)):
        rows = [l.split(";") for l in open(tmp_path / name).read().split()]
        sizes = [int(r[0]) for r in rows]
        assert all(float(r[1]) > 0 for r in rows)
        # pitches 3.2, 6.4, 12.8, 25.6 -> 4 grids (3 from 6.4)
        assert len(rows) == grids * reps
        if name == "rbdl.csv":
            continue
        nx = lambda p: len(np.arange(-100, 601 + 1e-6, p))
        # float accumulation in bench.cpp's arange may differ by one sample from numpy's: allow it
        assert abs(sizes[0] - nx(3.2) * len(np.arange(-100, 51 + 1e-6, 3.2))) <= nx(3.2) + 50


def test_cpp_mirror_robot_full_struct_and_apply_oct(tmp_path, lrm):
    """host/sweep_main.cpp calls robot_full_struct(...) and apply_oct(...) of include/lrm_compat.hpp (the
    reference's signatures, several_leg.cu.h:12-14, several_leg_octree.cu.h:4) on .bin files; the ctypes path
    (lrm_positionability with the reference's 45 orientations and culls, lrm_apply_oct with default settings)
    must give the same accepted bodies / leaf centres."""
    from lrm_amd import workloads
    ground = workloads.terrain(n_side=64, seed=5)
    bodies = workloads.body_lattice(ground, 1500, seed=6)
    for stem, arr in (("body", bodies), ("target", ground)):
        for k, c in enumerate("xyz"):
            np.ascontiguousarray(arr[:, k]).tofile(tmp_path / f"{stem}_{c}.bin")
    out = subprocess.run([_built("lrm_sweep"), str(tmp_path), "1", "4"], check=True, capture_output=True, text=True)
    assert "robot_full_struct:" in out.stdout and "apply_oct:" in out.stdout
    legs = workloads.hexapod(lrm.get_M2_leg, 4)
    want_mask, _ = lrm.positionability(bodies, ground, legs, workloads.reference_sweep_quats(), reference_culls=True)
    got = np.stack([np.fromfile(tmp_path / f"accepted_{c}.bin", np.float32) for c in "xyz"], -1)
    counts = np.fromfile(tmp_path / "accepted_count.bin", np.int32)
    assert np.array_equal(got.view(np.uint32), bodies[want_mask != 0].view(np.uint32))
    assert len(counts) == len(got) and (counts == 3).all()  # several_leg.cu:868
    want_oct, _ = lrm.apply_oct(ground, legs[0])
    got_oct = np.stack([np.fromfile(tmp_path / f"oct_{c}.bin", np.float32) for c in "xyz"], -1)
    assert np.array_equal(got_oct.view(np.uint32), want_oct.view(np.uint32))

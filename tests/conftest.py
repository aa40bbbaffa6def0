import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so): the checker, never the thing under test."""
    from oracle.orc import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own host path (oracle/_ref/libref.so); present where it was built."""
    from oracle import orc
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return orc.Ref()


@pytest.fixture(scope="session")
def lrm():
    import lrm_amd
    if not os.path.exists(lrm_amd.LIB_PATH):
        lrm_amd.build()
    lrm_amd.load()
    return lrm_amd


def golden_cases(prefix=""):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    return [n for n in names if n != "legs" and not n.startswith("terrain") and n.startswith(prefix)]


def reference_terrain():
    """The reference's own terrain cloud and near-ground body lattice (tests/golden/make_terrain.py)."""
    return dict(np.load(os.path.join(GOLDEN, "terrain_ground.npz")))


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def bits_equal(a, b):
    """Bitwise equality of float arrays, treating any-NaN == any-NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    same = a.view(np.uint32) == b.view(np.uint32)
    return same | (np.isnan(a) & np.isnan(b))


def random_cloud(n, seed=42):
    """BASELINE config 2 cloud: uniform in the leg's bounding cube, seed 42."""
    rng = np.random.default_rng(seed)
    lo = np.array([-200, -500, -500], np.float32)
    hi = np.array([700, 500, 300], np.float32)
    return (rng.random((n, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)

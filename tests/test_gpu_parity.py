"""GPU parity (run with -m gpu on an MI355X): the HIP kernels, called through the C ABI,
against the reference fixtures, the CPU oracle, and size-independent properties at
BASELINE.json's full sizes.  Strict mode must be BIT-IDENTICAL: mask, validity byte and every
float of the distance field."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal, golden_cases, load_case, random_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["strict", "fast", "fast_table", "fast_filtered"])
def mode(request, lrm, monkeypatch):
    """Every GPU test runs in the bit-exact arithmetic modes; all must be bit-identical to the oracle.
      strict         lrm_point.h verbatim
      fast           the library default: the table-guided kernel (dist_xtab_kernel, lrm_point_xtab.h) from 2e5 points on,
                     the filtered kernel (dist_soa_kernel<., true>, lrm_point_fast.h) below and for legs without a table
      fast_table     LRM_TOL_TABLE=2: the table-guided kernel whatever the size (every fixture, every ragged size)
      fast_filtered  LRM_XTAB=0: the filtered kernel whatever the size (what rounds 1-3 shipped; still the fallback)"""
    lrm.set_mode(lrm.MODE_STRICT if request.param == "strict" else lrm.MODE_FAST)
    if request.param == "fast_table":
        monkeypatch.setenv("LRM_TOL_TABLE", "2")
    if request.param == "fast_filtered":
        monkeypatch.setenv("LRM_XTAB", "0")
    yield request.param
    lrm.set_mode(lrm.MODE_FAST)  # the library default


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a GPU"
    return torch


def soa(torch, pts):
    t = torch.from_numpy(np.ascontiguousarray(pts.T)).cuda()
    return t[0], t[1], t[2]


@pytest.mark.parametrize("name", golden_cases())
def test_host_buffer_api_matches_reference_fixture(lrm, name):
    """lrm_reach / lrm_dist / lrm_reach_dist = apply_kernel drop-ins (AoS host buffers)."""
    c = load_case(name)
    m, ms = lrm.apply_reach(c["points"], c["leg"], c["quat"])
    d, v, ms2 = lrm.apply_dist(c["points"], c["leg"], c["quat"])
    m2, d2, ms3 = lrm.apply_reach_dist(c["points"], c["leg"], c["quat"])
    assert ms > 0 and ms2 > 0 and ms3 > 0
    assert np.array_equal(m, c["mask"]) and np.array_equal(m2, c["mask"])
    assert np.array_equal(v, c["valid"])
    assert bits_equal(d, c["dist"]).all() and bits_equal(d2, c["dist"]).all()


@pytest.mark.parametrize("name", golden_cases("cube") + golden_cases("boundary") + golden_cases("special"))
def test_device_api_matches_reference_fixture(lrm, torch_cuda, name):
    c = load_case(name)
    x, y, z = soa(torch_cuda, c["points"])
    m, bits = lrm.device.reach(x, y, z, c["leg"], c["quat"], want_bits=True)
    d, v = lrm.device.dist(x, y, z, c["leg"], c["quat"])
    m2, d2 = lrm.device.reach_dist(x, y, z, c["leg"], c["quat"])
    torch_cuda.cuda.synchronize()
    assert np.array_equal(m.cpu().numpy(), c["mask"])
    assert np.array_equal(m2.cpu().numpy(), c["mask"])
    assert np.array_equal(v.cpu().numpy(), c["valid"])
    assert bits_equal(d.cpu().numpy().T, c["dist"]).all()
    assert bits_equal(d2.cpu().numpy().T, c["dist"]).all()
    # ballot bit mask == byte mask
    n = len(c["mask"])
    packed = np.packbits(np.pad(c["mask"], (0, (-n) % 64)), bitorder="little").view(np.uint64)
    assert np.array_equal(bits.cpu().numpy().view(np.uint64), packed)


# sizes around 2^20: several grid-stride iterations, quads and tails of the single-launch reach kernel
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 4099, 100003,
                               1048576, 1048577, 1048579, 1048639, 1048641])
def test_ragged_sizes(lrm, oracle, torch_cuda, n):
    pts = random_cloud(max(n, 1), seed=n + 1)[:n]
    leg = lrm.get_M2_leg(0.3)
    q = (0.98, 0.0, 0.15, 0.05)
    want_m = oracle.reach(pts, leg, q)
    want_d, want_v = oracle.dist(pts, leg, q)
    if n == 0:
        m, _ = lrm.apply_reach(pts, leg, q)
        assert m.shape == (0,)
        return
    # three separately allocated (16-byte aligned) component arrays: for n % 4 != 0 the rows of ONE (3, n)
    # tensor are misaligned and would always take the scalar kernel -- this way the vector kernel's n % 4 tail
    # and partial-word code run too
    x, y, z = (torch_cuda.from_numpy(np.ascontiguousarray(pts[:, k])).cuda() for k in range(3))
    assert all(t.data_ptr() % 16 == 0 for t in (x, y, z))
    # guard bytes after every output: kernels must not write past n
    mask = torch_cuda.full((n + 64,), 7, dtype=torch_cuda.uint8, device="cuda")
    out = torch_cuda.full((3, ((n + 16 + 3) // 4) * 4), -777.0, dtype=torch_cuda.float32, device="cuda")  # rows stay 16-byte aligned
    bits = torch_cuda.full(((n + 63) // 64 + 2,), -1, dtype=torch_cuda.int64, device="cuda")
    lrm.device.reach(x, y, z, leg, q, out=mask[:n], bits=bits[:(n + 63) // 64])
    ox = [out[i, :n] for i in range(3)]
    valid = torch_cuda.full((n + 64,), 7, dtype=torch_cuda.uint8, device="cuda")
    L = lrm.lib()
    from lrm_amd import _capi
    legp = np.ascontiguousarray(leg, np.float32)
    qp = np.ascontiguousarray(q, np.float32)
    _capi.check(L.lrm_dist_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), n, _capi._ptr(legp), _capi._ptr(qp),
                               ox[0].data_ptr(), ox[1].data_ptr(), ox[2].data_ptr(), valid.data_ptr(),
                               torch_cuda.cuda.current_stream().cuda_stream))
    torch_cuda.cuda.synchronize()
    assert np.array_equal(mask[:n].cpu().numpy(), want_m)
    assert (mask[n:] == 7).all() and (valid[n:] == 7).all() and (out[:, n:] == -777.0).all()
    assert (bits[(n + 63) // 64:] == -1).all()
    assert np.array_equal(valid[:n].cpu().numpy(), want_v)
    assert bits_equal(out[:, :n].cpu().numpy().T, want_d).all()
    packed = np.packbits(np.pad(want_m, (0, (-n) % 64)), bitorder="little").view(np.uint64)
    assert np.array_equal(bits[:(n + 63) // 64].cpu().numpy().view(np.uint64), packed)


def test_random_cloud_1e6_all_legs_and_orientations(lrm, oracle, torch_cuda):
    pts = random_cloud(1_000_000, seed=42)
    x, y, z = soa(torch_cuda, pts)
    for leg in (lrm.get_M2_leg(0.0), lrm.get_moonbot_leg(np.pi / 3)):
        for q in (None, (0.924, 0, -0.384, 0), (0.9, 0.1, 0.2, -0.3)):
            qq = (1, 0, 0, 0) if q is None else q
            m = lrm.device.reach(x, y, z, leg, q)
            d, v = lrm.device.dist(x, y, z, leg, q)
            torch_cuda.cuda.synchronize()
            assert np.array_equal(m.cpu().numpy(), oracle.reach(pts, leg, qq))
            want_d, want_v = oracle.dist(pts, leg, qq)
            assert np.array_equal(v.cpu().numpy(), want_v)
            assert bits_equal(d.cpu().numpy().T, want_d).all()


def test_full_size_config2_1e7_points(lrm, oracle, torch_cuda, mode):
    """BASELINE config 2 at full size: 1e7 random targets, M2 leg, identity orientation.
    The oracle is run on the whole cloud for the mask (~1 s) and on a 2e6 slice for the
    distance field; the rest of the field is covered by properties."""
    if mode == "fast_table":
        pytest.skip("at this size the default dispatch already runs the table-guided kernel")
    n = 10_000_000
    pts = random_cloud(n, seed=42)
    leg = lrm.get_M2_leg(0.0)
    x, y, z = soa(torch_cuda, pts)
    m, bits = lrm.device.reach(x, y, z, leg, want_bits=True)
    m2, d = lrm.device.reach_dist(x, y, z, leg)
    torch_cuda.cuda.synchronize()
    mask = m.cpu().numpy()
    assert np.array_equal(mask, oracle.reach(pts, leg))
    assert np.array_equal(m2.cpu().numpy(), mask)
    assert 0.02 < mask.mean() < 0.08
    # bit mask: popcount == sum of bytes, and unpacks to the bytes
    b = bits.cpu().numpy().view(np.uint8)
    assert np.array_equal(np.unpackbits(b, bitorder="little")[:n], mask)
    dn = d.cpu().numpy().T
    want_d, _ = oracle.dist(pts[:2_000_000], leg)
    assert bits_equal(dn[:2_000_000], want_d).all()
    # properties at full size: the field is finite; moving a point by minus its distance vector
    # lands on the boundary of the reachable set: the residual distance there is ~0
    assert np.isfinite(dn).all()
    sel = np.random.default_rng(0).integers(0, n, 200000)
    on_boundary = (pts[sel] - dn[sel]).astype(np.float32)
    resid, _ = oracle.dist(on_boundary, leg)
    rn = np.linalg.norm(resid, axis=1)
    assert np.quantile(rn, 0.999) < 2e-2, np.quantile(rn, [0.5, 0.99, 0.999, 1.0])


def test_device_sqrt_is_correctly_rounded_everywhere(lrm, torch_cuda):
    """lrm_sqrtf (Goldschmidt/Markstein on v_rsq_f32 inside [2^-96, 2^96), the compiler's IEEE
    expansion elsewhere) == IEEE sqrtf on every one of the 2^32 float bit patterns, on the device."""
    from lrm_amd import _capi
    bad, first = _capi.dbg_sqrt_check_dev()
    assert bad == 0, f"{bad} patterns differ, first 0x{first:08x}"


def test_exact_math_device_matches_host(lrm, torch_cuda):
    """The device build of lrm_exact_math.h == its host build (itself checked against glibc
    in the CPU suite) on angles, coordinates and raw bit patterns."""
    rng = np.random.default_rng(1)
    n = 2_000_000
    a = np.concatenate([rng.uniform(-7, 7, n), rng.standard_normal(n) * 300,
                        rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    b = np.concatenate([rng.standard_normal(n) * 300, rng.standard_normal(n) * 300,
                        rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)]).astype(np.float32)
    keep = np.isfinite(a) & (np.abs(a) < 100)  # sincosf's emulated range
    a, b = a[keep], b[keep]
    outs = [np.empty_like(a) for _ in range(3)]
    P = lambda v: v.ctypes.data_as(C.c_void_p)
    assert lrm.lib().lrm_dbg_exact_math_host(P(a), P(b), len(a), *[P(o) for o in outs]) == 0
    ta, tb = torch_cuda.from_numpy(a).cuda(), torch_cuda.from_numpy(b).cuda()
    douts = [torch_cuda.empty_like(ta) for _ in range(3)]
    from lrm_amd import _capi
    _capi.check(lrm.lib().lrm_dbg_exact_math_dev(ta.data_ptr(), tb.data_ptr(), len(a), *[o.data_ptr() for o in douts],
                                                 torch_cuda.cuda.current_stream().cuda_stream))
    torch_cuda.cuda.synchronize()
    for h, d_ in zip(outs, douts):
        assert bits_equal(h, d_.cpu().numpy()).all()


def test_device_sqrt_and_div_are_correctly_rounded(torch_cuda, lrm, oracle):
    """The strict kernels rely on IEEE sqrtf and division on the device; a leg whose circle
    tests sit on awkward values exercises both (bit equality with the oracle is the check)."""
    pts = random_cloud(300000, seed=99) * np.float32(1.0000001)
    leg = lrm.leg_factory(0.37, 150.3, -33.3, 61.7, 117.9, 142.2, 55.0, 85.0, 115.0, -7.5, -3.5)
    x, y, z = soa(torch_cuda, pts)
    d, v = lrm.device.dist(x, y, z, leg, (0.99, 0.02, -0.1, 0.03))
    torch_cuda.cuda.synchronize()
    want_d, want_v = oracle.dist(pts, leg, (0.99, 0.02, -0.1, 0.03))
    assert np.array_equal(v.cpu().numpy(), want_v)
    assert bits_equal(d.cpu().numpy().T, want_d).all()


def test_random_legs_including_filter_ineligible_ones(lrm, oracle, torch_cuda):
    """Random leg geometries and orientations; some legs (yaw limits beyond 90 degrees) are outside
    the filter's eligibility and must silently take the strict kernels: results identical anyway."""
    rng = np.random.default_rng(7)
    pts = random_cloud(150000, seed=3)
    x, y, z = soa(torch_cuda, pts)
    for trial in range(10):
        coxa_deg = rng.uniform(30, 80) if trial % 3 else rng.uniform(95, 150)  # every third: ineligible
        leg = lrm.leg_factory(rng.uniform(-3, 3), rng.uniform(80, 250), rng.uniform(-60, 30), rng.uniform(30, 90),
                              rng.uniform(90, 160), rng.uniform(90, 170), coxa_deg, rng.uniform(60, 100),
                              rng.uniform(90, 140), rng.uniform(-20, 10), rng.uniform(-20, 10))
        q = rng.normal(size=4).astype(np.float32)
        q[0] += 3.0
        m = lrm.device.reach(x, y, z, leg, q)
        m2, d = lrm.device.reach_dist(x, y, z, leg, q)
        torch_cuda.cuda.synchronize()
        want_m = oracle.reach(pts, leg, q)
        want_d, _ = oracle.dist(pts, leg, q)
        assert np.array_equal(m.cpu().numpy(), want_m) and np.array_equal(m2.cpu().numpy(), want_m)
        assert bits_equal(d.cpu().numpy().T, want_d).all()


def test_non_finite_and_extreme_inputs(lrm, oracle, torch_cuda):
    """nan / inf / huge / denormal coordinates: same bytes and same floats (nan == nan) as the oracle."""
    vals = np.array([0.0, -0.0, 1e-42, -1e-42, 1e-30, 300.0, -300.0, 1e7, -1e7, 1e30, 3e38, np.inf, -np.inf, np.nan],
                    np.float32)
    g = np.stack(np.meshgrid(vals, vals, vals, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    x, y, z = soa(torch_cuda, g)
    for leg in (lrm.get_M2_leg(0.3), lrm.get_moonbot_leg(0.0)):
        for q in (None, (0.95, 0.1, -0.2, 0.15)):
            qq = (1, 0, 0, 0) if q is None else q
            m = lrm.device.reach(x, y, z, leg, q)
            d, v = lrm.device.dist(x, y, z, leg, q)
            torch_cuda.cuda.synchronize()
            want_d, want_v = oracle.dist(g, leg, qq)
            assert np.array_equal(m.cpu().numpy(), oracle.reach(g, leg, qq))
            assert np.array_equal(v.cpu().numpy(), want_v)
            assert bits_equal(d.cpu().numpy().T, want_d).all()
            # the fused kernel derives its mask from the distance evaluation (lrm_reach_from_dist)
            mf, df = lrm.device.reach_dist(x, y, z, leg, q)
            torch_cuda.cuda.synchronize()
            assert np.array_equal(mf.cpu().numpy(), oracle.reach(g, leg, qq))
            assert bits_equal(df.cpu().numpy().T, want_d).all()


def test_fused_mask_on_points_behind_the_coxa_axis_and_on_yaw_limits(lrm, oracle, torch_cuda):
    """The fused kernel takes the reach mask from the distance's direct candidate (coxa-frame x >= 0,
    an identity) or filters it through the flipped candidate (x < 0): dense clouds around the coxa
    axis, where points are mirrored, and on the yaw-limit planes, where the flipped candidate's yaw
    is within rounding of the limit."""
    rng = np.random.default_rng(11)
    for leg in (lrm.get_M2_leg(0.0), lrm.get_M2_leg(2.0943952), lrm.get_moonbot_leg(0.7)):
        body, pitch, coxa = float(leg[1]), float(leg[2]), float(leg[3])
        n = 400_000
        # (a) a slab around the coxa axis (coxa-frame x ~ 0), in the leg's own azimuth frame
        a = np.stack([body + rng.normal(0, 30, n), rng.uniform(-400, 400, n), rng.uniform(-400, 200, n)], 1)
        # (b) points on the two yaw-limit planes, on both sides of the axis, with ulp-scale scatter
        r = rng.uniform(-400, 400, n)
        lim = np.where(rng.random(n) < 0.5, float(leg[8]), float(leg[9])) + rng.normal(0, 2e-7, n)
        zc = rng.uniform(-300, 200, n)
        xc, yc = r * np.cos(lim), r * np.sin(lim)
        # undo place_over_coxa (x -= body, then (x, z) rotated by -pitch): rotate by +pitch, add the body offset
        b = np.stack([xc * np.cos(pitch) - zc * np.sin(pitch) + body, yc, xc * np.sin(pitch) + zc * np.cos(pitch)], 1)
        ang = float(leg[0])
        pts = np.concatenate([a, b]).astype(np.float64)
        rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        pts = (pts @ rot.T).astype(np.float32)  # leg azimuth frame -> body frame
        x, y, z = soa(torch_cuda, pts)
        for q in (None, (0.97, 0.05, -0.1, 0.2)):
            qq = (1, 0, 0, 0) if q is None else q
            mf, df = lrm.device.reach_dist(x, y, z, leg, q)
            torch_cuda.cuda.synchronize()
            want_m = oracle.reach(pts, leg, qq)
            want_d, _ = oracle.dist(pts, leg, qq)
            assert np.array_equal(mf.cpu().numpy(), want_m)
            assert bits_equal(df.cpu().numpy().T, want_d).all()
            if q is None:
                assert 0.001 < want_m.mean() < 0.9


def test_config4_cloud_1e8_points_on_one_gpu(lrm, oracle, torch_cuda, mode):
    """BASELINE config 4's whole 1e8-point cloud on ONE GPU (the multi-GPU job gives each rank an
    eighth of it): reach mask + bit words, checked exactly against the oracle on three 1e6-point
    windows and through size-independent properties everywhere (bytes are 0/1, bit words unpack to
    the bytes, popcount = byte sum, a shifted window of the same points gives the same answers)."""
    if mode != "fast":
        pytest.skip("the 1e8-point run is done once, in the default mode")
    n = 100_000_000
    rng = np.random.default_rng(4)
    host = np.empty((3, n), np.float32)
    lo = np.array([-200, -500, -500], np.float32)
    hi = np.array([700, 500, 300], np.float32)
    for s0 in range(0, n, 10_000_000):
        host[:, s0:s0 + 10_000_000] = (rng.random((10_000_000, 3), dtype=np.float32) * (hi - lo) + lo).T
    dev = torch_cuda.from_numpy(host).cuda()
    leg = lrm.get_M2_leg(0.0)
    m, bits = lrm.device.reach(dev[0], dev[1], dev[2], leg, want_bits=True)
    torch_cuda.cuda.synchronize()
    mask = m.cpu().numpy()
    assert mask.max() <= 1
    b = bits.cpu().numpy().view(np.uint8)
    assert np.array_equal(np.unpackbits(b, bitorder="little")[:n], mask)
    for w0 in (0, 49_999_937, n - 1_000_000):
        pts = np.ascontiguousarray(host[:, w0:w0 + 1_000_000].T)
        assert np.array_equal(mask[w0:w0 + 1_000_000], oracle.reach(pts, leg))
    # an unaligned window of the same buffer (scalar kernel) must reproduce the same bytes
    w0, wn = 12_345_677, 3_000_001
    m2 = lrm.device.reach(dev[0][w0:w0 + wn], dev[1][w0:w0 + wn], dev[2][w0:w0 + wn], leg)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(m2.cpu().numpy(), mask[w0:w0 + wn])


def test_device_api_rejects_bad_output_tensors(lrm, torch_cuda):
    """Outputs travel as raw pointers: short, mistyped, strided or host tensors must be refused, not written through."""
    pts = random_cloud(1000, seed=1)
    x, y, z = soa(torch_cuda, pts)
    leg = lrm.get_M2_leg(0.0)
    t = torch_cuda
    with pytest.raises(ValueError):
        lrm.device.reach(x, y, z, leg, out=t.empty(999, dtype=t.uint8, device="cuda"))
    with pytest.raises(ValueError):
        lrm.device.reach(x, y, z, leg, out=t.empty(1000, dtype=t.int32, device="cuda"))
    with pytest.raises(ValueError):
        lrm.device.reach(x, y, z, leg, bits=t.empty(15, dtype=t.int64, device="cuda"))
    with pytest.raises(ValueError):
        lrm.device.dist(x, y, z, leg, out=t.empty((1000, 3), dtype=t.float32, device="cuda").T)  # rows not contiguous
    with pytest.raises(ValueError):
        lrm.device.dist(x, y, z, leg, out=t.empty((3, 999), dtype=t.float32, device="cuda"))
    with pytest.raises(ValueError):
        lrm.device.reach_dist(x, y, z, leg, mask=t.empty(1000, dtype=t.uint8))  # host tensor
    # a view of a wider buffer is fine (contiguous rows)
    wide = t.empty((3, 1024), dtype=t.float32, device="cuda")
    d, v = lrm.device.dist(x, y, z, leg, out=wide[:, :1000])
    t.cuda.synchronize()
    assert d.shape == (3, 1000)

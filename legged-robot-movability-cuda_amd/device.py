"""Device-resident entry points on torch CUDA(ROCm) tensors.  torch only owns the memory and
the stream; every computation is a liblrm.so kernel launched on torch's current stream."""
import numpy as np

from . import _capi


def _torch():
    import torch
    return torch


def _dp(t):
    return None if t is None else t.data_ptr()


def _stream(ref=None):
    """torch's current stream ON THE DEVICE OF THE INPUTS (not on whatever device is current)"""
    torch = _torch()
    return torch.cuda.current_stream(ref.device if ref is not None else None).cuda_stream


def _check_f32(*ts):
    torch = _torch()
    n = ts[0].numel()
    dev = ts[0].device
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n and t.device == dev):
            raise ValueError("expected contiguous float32 CUDA tensors of equal length on one device")
    return n


def _check_out(t, ref, dtype, numel, what):
    """Caller-supplied outputs go to the kernels as raw pointers: a short, strided, mistyped or other-device tensor
    would be an out-of-bounds device write."""
    if t is None:
        return
    if not (t.is_cuda and t.device == ref.device and t.dtype == dtype and t.is_contiguous() and t.numel() >= numel):
        raise ValueError(f"{what}: expected a contiguous {dtype} tensor of >= {numel} elements on {ref.device}")


def _check_field(out, ref, n):
    """(3, n) distance field whose three rows are each contiguous (row stride may exceed n: a view of a wider buffer)"""
    torch = _torch()
    if not (out.is_cuda and out.device == ref.device and out.dtype == torch.float32 and out.dim() == 2 and out.shape[0] == 3
            and out.shape[1] >= n and (out.shape[1] <= 1 or out.stride(1) == 1)):
        raise ValueError(f"distance field: expected a float32 (3, >= {n}) tensor with contiguous rows on {ref.device}")


def _leg(leg):
    return np.ascontiguousarray(leg, dtype=np.float32).reshape(-1)


def _q(quat):
    return None if quat is None else np.ascontiguousarray(quat, dtype=np.float32).reshape(4)


def reach(x, y, z, leg, quat=None, out=None, bits=None, want_bits=False):
    """mask[i] = reachability_global(point i) (one byte per point); optionally also the
    ballot bit mask (int64 words, bit i&63 of word i>>6)."""
    torch = _torch()
    n = _check_f32(x, y, z)
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=x.device)
    if want_bits and bits is None:
        bits = torch.empty((n + 63) // 64, dtype=torch.int64, device=x.device)
    _check_out(out, x, torch.uint8, n, "mask")
    _check_out(bits, x, torch.int64, (n + 63) // 64, "bit words")
    leg = _leg(leg)
    q = _q(quat)
    L = _capi.load()
    with torch.cuda.device(x.device):
        if bits is not None:
            _capi.check(L.lrm_reach_bits_dev(_dp(x), _dp(y), _dp(z), n, _capi._ptr(leg), _capi._ptr(q), _dp(out),
                                             _dp(bits), _stream(x)))
            return out, bits
        _capi.check(L.lrm_reach_dev(_dp(x), _dp(y), _dp(z), n, _capi._ptr(leg), _capi._ptr(q), _dp(out), _stream(x)))
    return out


def dist(x, y, z, leg, quat=None, out=None, valid=None, want_valid=True):
    torch = _torch()
    n = _check_f32(x, y, z)
    if out is None:
        out = torch.empty((3, n), dtype=torch.float32, device=x.device)
    if valid is None and want_valid:
        valid = torch.empty(n, dtype=torch.uint8, device=x.device)
    _check_field(out, x, n)
    _check_out(valid, x, torch.uint8, n, "validity bytes")
    leg = _leg(leg)
    q = _q(quat)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().lrm_dist_dev(_dp(x), _dp(y), _dp(z), n, _capi._ptr(leg), _capi._ptr(q), _dp(out[0]),
                                              _dp(out[1]), _dp(out[2]), _dp(valid), _stream(x)))
    return out, valid


def reach_dist(x, y, z, leg, quat=None, mask=None, out=None, bits=None):
    """One launch: reach mask (bytes and/or ballot bit words) + distance field (3, n)."""
    torch = _torch()
    n = _check_f32(x, y, z)
    if out is None:
        out = torch.empty((3, n), dtype=torch.float32, device=x.device)
    if mask is None and bits is None:
        mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    _check_field(out, x, n)
    _check_out(mask, x, torch.uint8, n, "mask")
    _check_out(bits, x, torch.int64, (n + 63) // 64, "bit words")
    leg = _leg(leg)
    q = _q(quat)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().lrm_reach_dist_bits_dev(_dp(x), _dp(y), _dp(z), n, _capi._ptr(leg), _capi._ptr(q),
                                                         _dp(mask), _dp(bits), _dp(out[0]), _dp(out[1]), _dp(out[2]),
                                                         _stream(x)))
    if bits is not None:
        return mask, out, bits
    return mask, out


def reach_any(bx, by, bz, tx, ty, tz, legs, quat=None, out=None, all_legs=None):
    """out[l, b] = any target reachable by leg l from body b (legs used as given);
    all_legs[b] = AND over legs."""
    torch = _torch()
    nb = _check_f32(bx, by, bz)
    nt = _check_f32(tx, ty, tz)
    legs = np.ascontiguousarray(legs, dtype=np.float32).reshape(-1, 14)
    if out is None:
        out = torch.empty((len(legs), nb), dtype=torch.uint8, device=bx.device)
    if all_legs is None:
        all_legs = torch.empty(nb, dtype=torch.uint8, device=bx.device)
    q = _q(quat)
    _check_out(out, bx, torch.uint8, len(legs) * nb, "per-leg results")
    _check_out(all_legs, bx, torch.uint8, nb, "per-body results")
    if nt and tx.device != bx.device:
        raise ValueError("bodies and targets must live on one device")
    with torch.cuda.device(bx.device):
        _capi.check(_capi.load().lrm_reach_any_dev(_dp(bx), _dp(by), _dp(bz), nb, _dp(tx), _dp(ty), _dp(tz), nt,
                                                   _capi._ptr(legs), len(legs), _capi._ptr(q), _dp(out),
                                                   _dp(all_legs), _stream(bx)))
    return out, all_legs


def positionability(bx, by, bz, tx, ty, tz, legs, quats, reference_culls=0, active=None, out=None):
    """lrm_positionability_dev: the orientation sweep of robot_full_struct on device-resident clouds and masks.
    reference_culls: 0 none, 2 the per-orientation cylinder culls.  -> (accepted uint8[nb] on the device, kernel ms)"""
    import ctypes as C
    torch = _torch()
    nb = _check_f32(bx, by, bz)
    nt = _check_f32(tx, ty, tz) if tx.numel() else 0
    legs = np.ascontiguousarray(legs, dtype=np.float32).reshape(-1, 14)
    quats = np.ascontiguousarray(quats, dtype=np.float32).reshape(-1, 4)
    if out is None:
        out = torch.empty(nb, dtype=torch.uint8, device=bx.device)
    _check_out(out, bx, torch.uint8, nb, "accepted bytes")
    _check_out(active, bx, torch.uint8, nb, "active bytes")
    ms = C.c_float(0)
    with torch.cuda.device(bx.device):
        torch.cuda.synchronize(bx.device)  # the library works on the null stream
        _capi.check(_capi.load().lrm_positionability_dev(_dp(bx), _dp(by), _dp(bz), nb, _dp(tx) if nt else None, _dp(ty) if nt else None,
                                                         _dp(tz) if nt else None, nt, _capi._ptr(legs), len(legs), _capi._ptr(quats),
                                                         len(quats), int(reference_culls), _dp(active), _dp(out), C.addressof(ms)))
    return out, ms.value


def any_in_sphere(cx, cy, cz, tx, ty, tz, radius, out=None):
    torch = _torch()
    nc = _check_f32(cx, cy, cz)
    nt = _check_f32(tx, ty, tz)
    if out is None:
        out = torch.empty(nc, dtype=torch.uint8, device=cx.device)
    _check_out(out, cx, torch.uint8, nc, "results")
    with torch.cuda.device(cx.device):
        _capi.check(_capi.load().lrm_any_in_sphere_dev(_dp(cx), _dp(cy), _dp(cz), nc, _dp(tx), _dp(ty), _dp(tz), nt,
                                                       radius, _dp(out), _stream(cx)))
    return out


def any_in_cylinder(cx, cy, cz, tx, ty, tz, radius, plus_z, minus_z, out=None):
    torch = _torch()
    nc = _check_f32(cx, cy, cz)
    nt = _check_f32(tx, ty, tz)
    if out is None:
        out = torch.empty(nc, dtype=torch.uint8, device=cx.device)
    _check_out(out, cx, torch.uint8, nc, "results")
    with torch.cuda.device(cx.device):
        _capi.check(_capi.load().lrm_any_in_cylinder_dev(_dp(cx), _dp(cy), _dp(cz), nc, _dp(tx), _dp(ty), _dp(tz), nt,
                                                         radius, plus_z, minus_z, _dp(out), _stream(cx)))
    return out


def apply_oct_partitioned(x, y, z, leg, settings, exchange):
    """apply_oct with THIS rank's part of the footholds on the device (lrm_apply_oct_partitioned_dev); `exchange` ORs the
    flag words over the ranks.  -> (centres float32[k, 3] on the host, kernel milliseconds)"""
    torch = _torch()
    n = _check_f32(x, y, z) if x.numel() else 0
    with torch.cuda.device(x.device):
        torch.cuda.synchronize(x.device)  # the library works on the null stream
        return _capi.apply_oct_partitioned_dev(_dp(x) if n else None, _dp(y) if n else None, _dp(z) if n else None, n, leg, settings, exchange)


def apply_oct(x, y, z, leg, settings=None, rank=0, world=1, exchange=None):
    """apply_oct on footholds that already live on the device (three float32 tensors); the level loop synchronises the
    device, so this is not a stream-ordered call.  -> (centres float32[k, 3] on the host, kernel milliseconds)"""
    torch = _torch()
    n = _check_f32(x, y, z)
    with torch.cuda.device(x.device):
        torch.cuda.synchronize(x.device)  # the library works on the null stream
        return _capi.apply_oct_dev(_dp(x), _dp(y), _dp(z), n, leg, settings, rank, world, exchange)

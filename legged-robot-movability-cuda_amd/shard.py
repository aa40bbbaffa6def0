"""Multi-GPU sharding of the path: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-device (several_leg.cu:800 uses device 0); this is new capability
(SURVEY.md section 8e).  Targets are independent, so rank r owns the contiguous slice
[lo, hi) of the cloud, computes its reach bits with no communication, and the only exchange
is one all-gather of the bit-packed mask (n/8 bytes in total: 12.5 MB for 1e8 points).
Shard boundaries are multiples of 64 points so that no 64-bit mask word straddles two ranks.
For the body x target aggregation the bodies are sharded instead and the per-body bytes are
gathered the same way.
"""
import numpy as np


def shard_size(n, world, align=64):
    """Points per rank: ceil(n / world) rounded up to `align`."""
    per = -(-n // world)
    return -(-per // align) * align


def shard_bounds(n, world, rank, align=64):
    per = shard_size(n, world, align)
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def pack_bits(mask):
    """uint8 0/1 mask -> int64 words (bit i&63 of word i>>6), the layout of lrm_reach_bits_dev."""
    mask = np.asarray(mask, np.uint8)
    pad = (-len(mask)) % 64
    return np.packbits(np.pad(mask, (0, pad)), bitorder="little").view(np.int64)


def unpack_bits(words, n):
    return np.unpackbits(np.asarray(words).view(np.uint8), bitorder="little")[:n]


def all_gather_bits(local_words, n, group=None):
    """local_words: this rank's int64 words (torch tensor, CPU for gloo / CUDA for nccl) for its
    shard of an n-point cloud.  Returns the ceil(n/64) words of the whole cloud on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per_words = shard_size(n, world) // 64
    buf = torch.zeros(per_words, dtype=torch.int64, device=local_words.device)
    buf[: local_words.numel()] = local_words
    out = torch.empty(per_words * world, dtype=torch.int64, device=local_words.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out[: (n + 63) // 64]


def all_gather_bytes(local, n, group=None, align=64):
    """Same for one-byte-per-item results (per-body masks of the aggregation)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = shard_size(n, world, align)
    buf = torch.zeros(per, dtype=torch.uint8, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty(per * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out[:n]


def reach_bits_sharded(compute_local_bits, n, group=None):
    """Run `compute_local_bits(lo, hi) -> int64 word tensor` on this rank's slice and gather."""
    import torch.distributed as dist
    lo, hi = shard_bounds(n, dist.get_world_size(group), dist.get_rank(group))
    return all_gather_bits(compute_local_bits(lo, hi), n, group)


# ---------------------------------------------------------------------------------------------------
# The step loop of `bench.py --gpus N` (and of any caller that streams clouds through the fused kernel):
# rank r owns shard_bounds(n, world, r) of ONE n-point cloud, a step = local kernel + all-gather of the
# bit-packed reach mask.  Two word buffers alternate so that the gather of step k (side stream) overlaps the
# kernel of step k + 1.  The local computation is injected, so the control flow runs under gloo on the CPU.
# ---------------------------------------------------------------------------------------------------
class BitsGatherLoop:
    def __init__(self, n, device="cuda", group=None, host_staging=False, time_gather=False):
        """n: points of the whole cloud.  device: where the word buffers live ("cuda" for RCCL, "cpu" for gloo).
        host_staging: compute on `device` but gather through host memory (gloo rehearsals of a GPU run).
        time_gather: keep the duration of every gather (events on the side stream / host clock): gather_ms()."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n = n
        self.lo, self.hi = shard_bounds(n, self.world, self.rank)
        self.per_words = shard_size(n, self.world) // 64
        self.local_words = (self.hi - self.lo + 63) // 64
        self.cuda = str(device).startswith("cuda")
        self.host_staging = host_staging
        # padded to the common shard size: the tail (ragged last shards) stays 0
        self.words = [torch.zeros(max(self.per_words, 1), dtype=torch.int64, device=device) for _ in range(2)]
        gdev = "cpu" if host_staging else device
        self.gathered = [torch.zeros(max(self.per_words, 1) * self.world, dtype=torch.int64, device=gdev) for _ in range(2)]
        self.comm_stream = torch.cuda.Stream() if (self.cuda and self.world > 1) else None
        self.free = [None, None]  # event after which words[b] may be rewritten
        self.time_gather = time_gather and self.world > 1
        self._gather_events, self._gather_host_ms = [], []

    def step(self, k, compute):
        """compute(words_view, lo, hi) launches the local evaluation of points [lo, hi) writing words_view
        (ceil((hi - lo) / 64) int64 words) on the current stream.  Returns the buffer index used."""
        torch = self.torch
        b = k & 1
        if self.cuda and self.free[b] is not None:
            torch.cuda.current_stream().wait_event(self.free[b])  # the gather of step k - 2 has read words[b]
        compute(self.words[b][: self.local_words], self.lo, self.hi)
        if self.world == 1:
            return b
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()
            self.comm_stream.wait_event(ready)
            with torch.cuda.stream(self.comm_stream):
                if self.time_gather and not self.host_staging:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    self._gather(b)
                    t1.record()
                    self._gather_events.append((t0, t1))
                else:
                    self._gather(b)
                self.free[b] = torch.cuda.Event()
                self.free[b].record()
        else:
            self._gather(b)
        return b

    def reset_gather_timing(self):
        self._gather_events, self._gather_host_ms = [], []

    def gather_ms(self):
        """mean duration of the gathers since the last reset (None without time_gather or with one rank); call after a
        device synchronisation"""
        if not self.time_gather:
            return None
        ms = [a.elapsed_time(b) for a, b in self._gather_events] + self._gather_host_ms
        return float(sum(ms) / len(ms)) if ms else None

    def _gather(self, b):
        import time
        host_timed = self.time_gather and (self.host_staging or not self.cuda)
        t0 = time.perf_counter() if host_timed else 0.0
        src = self.words[b].cpu() if self.host_staging else self.words[b]
        self.dist.all_gather_into_tensor(self.gathered[b], src, group=self.group)
        if host_timed:
            self._gather_host_ms.append((time.perf_counter() - t0) * 1e3)

    def result(self, b):
        """The ceil(n / 64) words of the whole cloud from buffer b (synchronises the side stream)."""
        if self.cuda:
            self.torch.cuda.synchronize()
        if self.world == 1:
            return self.words[b][: (self.n + 63) // 64]
        return self.gathered[b][: (self.n + 63) // 64]


# ---------------------------------------------------------------------------------------------------
# Positionability on several GPUs.  The reference has none of this (single device).
# ---------------------------------------------------------------------------------------------------
class DeviceBackend:
    """The GPU implementation of the three operations the sharded drivers compose (host arrays in and out)."""

    def any_in_sphere(self, centres, targets, radius):
        import torch
        from . import device
        if len(centres) == 0:
            return np.zeros(0, np.uint8)
        c = torch.from_numpy(np.ascontiguousarray(np.asarray(centres, np.float32).T)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(targets, np.float32).T)).cuda()
        out = device.any_in_sphere(c[0], c[1], c[2], t[0], t[1], t[2], float(radius))
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def positionability(self, bodies, targets, legs, quats, culls):
        from . import _capi
        if len(bodies) == 0:
            return np.zeros(0, np.uint8)
        return _capi.positionability(bodies, targets, legs, quats, reference_culls=culls)[0]

    def reach_any(self, bodies, targets, legs, quat):
        import torch
        from . import device
        b = torch.from_numpy(np.ascontiguousarray(np.asarray(bodies, np.float32).T)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(targets, np.float32).T)).cuda()
        out, _ = device.reach_any(b[0], b[1], b[2], t[0], t[1], t[2], legs, quat)
        torch.cuda.synchronize()
        return out.cpu().numpy()


def _dist_info(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size(group), dist.get_rank(group)
    return None, 1, 0


def _comm_device(dist, group):
    return "cuda" if (dist is not None and dist.get_backend(group) == "nccl") else "cpu"


def _is_cuda(t):
    return hasattr(t, "is_cuda") and t.is_cuda


def _rows(t):
    """(3, n) CUDA tensor -> its three component rows, each contiguous"""
    return tuple(t[k].contiguous() for k in range(3))


def _reduce_max_bytes(t, dist, group):
    """all_reduce(MAX) of a uint8 CUDA tensor: in place over RCCL; through host memory for a CPU backend (gloo rehearsals)"""
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
    return h.to(t.device)


def _positionability_sharded_dev(bodies, targets, legs, quats, reference_culls, group):
    """positionability_sharded on device-resident clouds: (3, nb) and (3, nt) float32 CUDA tensors in, a uint8 CUDA tensor
    [nb] out; the culls' masks, the all-reduce of the far-target cull, the compactions and the sweep all stay on the
    device (RCCL reduces and gathers device memory; only a CPU backend stages the exchanged bytes through the host)."""
    import torch
    from . import device
    dist, world, rank = _dist_info(group)
    nb, nt = bodies.shape[1], targets.shape[1]
    lo, hi = shard_bounds(nb, world, rank)
    mine = bodies[:, lo:hi]
    local = torch.zeros(hi - lo, dtype=torch.uint8, device=bodies.device)
    legs = np.ascontiguousarray(np.asarray(legs, np.float32).reshape(-1, 14))
    if hi > lo or (dist is not None and world > 1):
        mx, my, mz = _rows(mine)
        tx, ty, tz = _rows(targets)
        if reference_culls and nt:
            keep = torch.zeros(nt, dtype=torch.uint8, device=bodies.device)
            alive = torch.zeros(hi - lo, dtype=torch.bool, device=bodies.device)
            if hi > lo:
                collide = device.any_in_sphere(mx, my, mz, tx, ty, tz, 60.0)   # eliminateAlwaysColliding
                near = device.any_in_sphere(mx, my, mz, tx, ty, tz, 400.0)     # eliminateFarBody
                alive = (collide == 0) & (near != 0)
                if bool(alive.any()):
                    ax, ay, az = (c[alive].contiguous() for c in (mx, my, mz))
                    keep = device.any_in_sphere(tx, ty, tz, ax, ay, az, 400.0)
            if dist is not None and world > 1:                                  # eliminateFarTarget over ALL ranks' survivors
                keep = _reduce_max_bytes(keep, dist, group)
            if bool(alive.any()):
                sel = keep != 0
                kx, ky, kz = (c[sel].contiguous() for c in (tx, ty, tz))
                acc, _ = device.positionability(ax, ay, az, kx, ky, kz, legs, quats, 2)
                local[alive] = acc
        elif not reference_culls and hi > lo:
            local, _ = device.positionability(mx, my, mz, tx, ty, tz, legs, quats, 0)
    if dist is None or world == 1:
        return local
    if dist.get_backend(group) == "nccl":
        return all_gather_bytes(local, nb, group)
    return all_gather_bytes(local.cpu(), nb, group).to(bodies.device)


def positionability_sharded(bodies, targets, legs, quats, reference_culls=False, backend=None, group=None):
    """robot_full_struct's result (lrm_positionability) with the BODIES split over the ranks: every rank holds all
    targets (1.2 MB for 1e5 points) and all legs, evaluates its contiguous slice of bodies, and the per-body bytes
    are all-gathered; every rank returns the mask of ALL bodies.

    With reference_culls the one-time culls of multi_rot_estimator (several_leg.cu:413-502) are evaluated here,
    because eliminateFarTarget keeps a target when ANY surviving body -- of any rank -- is within 400 mm: the
    per-rank keep masks are combined with all_reduce(MAX) (RCCL has no bitwise OR; max on 0/1 bytes is the same).
    Without a process group this is the single-process composition of the same steps.

    bodies / targets: host arrays (n, 3) -> a host uint8 array; or float32 CUDA tensors (3, n) (SoA, as the *_dev entry
    points take them) -> a uint8 CUDA tensor, with every intermediate mask kept on the device."""
    import torch
    if _is_cuda(bodies) and _is_cuda(targets) and backend is None:
        return _positionability_sharded_dev(bodies, targets, legs, quats, reference_culls, group)
    backend = backend or DeviceBackend()
    dist, world, rank = _dist_info(group)
    bodies = np.ascontiguousarray(np.asarray(bodies, np.float32).reshape(-1, 3))
    targets = np.ascontiguousarray(np.asarray(targets, np.float32).reshape(-1, 3))
    nb = len(bodies)
    lo, hi = shard_bounds(nb, world, rank)
    mine = bodies[lo:hi]
    local = np.zeros(hi - lo, np.uint8)
    if reference_culls and len(targets):
        collide = backend.any_in_sphere(mine, targets, 60.0)   # eliminateAlwaysColliding
        near = backend.any_in_sphere(mine, targets, 400.0)     # eliminateFarBody
        alive = (collide == 0) & (near != 0)
        keep = backend.any_in_sphere(targets, mine[alive], 400.0) if alive.any() else np.zeros(len(targets), np.uint8)
        if dist is not None and world > 1:                      # eliminateFarTarget over ALL ranks' survivors
            t = torch.from_numpy(np.ascontiguousarray(keep)).to(_comm_device(dist, group))
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            keep = t.cpu().numpy()
        kept = targets[keep != 0]
        if alive.any():
            local[alive] = backend.positionability(mine[alive], kept, legs, quats, 2)
    elif not reference_culls:
        local = backend.positionability(mine, targets, legs, quats, 0)
    if dist is None or world == 1:
        return local
    out = all_gather_bytes(torch.from_numpy(np.ascontiguousarray(local)).to(_comm_device(dist, group)), nb, group)
    return out.cpu().numpy()


def reach_any_target_sharded(bodies, local_targets, legs, quat=None, backend=None, group=None):
    """The any-flags of reach_mem_kernel (lrm_reach_any_dev) when the CLOUD is the big side (1e8 targets): every rank
    holds all bodies and its own slice of the targets; the per-(leg, body) bytes are combined with
    all_reduce(MAX) and the AND over legs is taken locally.  Returns (out[leg, body], all_legs[body]) on every rank.

    bodies / local_targets: host arrays (n, 3), or float32 CUDA tensors (3, n): then the flags never leave the device (the
    target shard -- 150 MB for a 1.25e7-point share -- is used where it lies) and CUDA tensors come back."""
    import torch
    dist, world, rank = _dist_info(group)
    if _is_cuda(bodies) and _is_cuda(local_targets) and backend is None:
        from . import device
        legs_a = np.ascontiguousarray(np.asarray(legs, np.float32).reshape(-1, 14))
        bx, by, bz = _rows(bodies)
        tx, ty, tz = _rows(local_targets)
        out, _ = device.reach_any(bx, by, bz, tx, ty, tz, legs_a, quat)
        if dist is not None and world > 1:
            out = _reduce_max_bytes(out.view(-1), dist, group).view(out.shape)
        return out, out.min(dim=0).values
    backend = backend or DeviceBackend()
    legs = np.ascontiguousarray(np.asarray(legs, np.float32).reshape(-1, 14))
    out = np.ascontiguousarray(backend.reach_any(bodies, local_targets, legs, quat).astype(np.uint8))
    if dist is not None and world > 1:
        t = torch.from_numpy(out).to(_comm_device(dist, group))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        out = t.cpu().numpy()
    return out, out.min(axis=0)


def _or_exchange(dist, group):
    """the octree's exchange callback: flags[:] = bitwise OR over the ranks of a uint32 array of 3-bit flag words (or of
    0xffffffff: a rank that failed).  RCCL has no bitwise-OR reduction: the three bits and the failure marker travel as
    four 0/1 bytes per word and are combined with MAX."""
    import torch
    dev = _comm_device(dist, group)

    def exchange(flags):
        f = torch.from_numpy(flags.astype(np.int64))
        planes = torch.stack([(f >> 0) & 1, (f >> 1) & 1, (f >> 2) & 1, (f == 0xffffffff).to(torch.int64)]).to(torch.uint8).to(dev)
        dist.all_reduce(planes, op=dist.ReduceOp.MAX, group=group)
        p = planes.cpu().numpy().astype(np.uint32)
        out = p[0] | (p[1] << 1) | (p[2] << 2)
        out[p[3] != 0] = 0xffffffff
        flags[:] = out

    return exchange


def octant_owner(footholds, box_center, world):
    """A spatial split for apply_oct_partitioned: the rank that owns each foothold = (index of the root box's octant the
    foothold lies in) mod world.  footholds: host array (n, 3) -> int array [n].  Any disjoint split is CORRECT (a
    child's flags are ORs over footholds); this one keeps a rank's footholds, and so its work, in its own subtrees."""
    f = np.asarray(footholds, np.float32).reshape(-1, 3)
    c = np.asarray(box_center, np.float32).reshape(3)
    octant = (f[:, 0] >= c[0]).astype(np.int64) | ((f[:, 1] >= c[1]).astype(np.int64) << 1) | ((f[:, 2] >= c[2]).astype(np.int64) << 2)
    return octant % max(int(world), 1)


def apply_oct_partitioned(local_footholds, leg, settings=None, group=None):
    """apply_oct (lrm_apply_oct) of a cloud that NO rank holds as a whole (BASELINE config 5 at scale: 1e8 footholds are
    1.2 GB -- a replica per GPU is what this avoids): every rank passes its own part of the footholds -- a host array
    (n, 3) or three CUDA tensors (x, y, z) -- evaluates every child of every level against it, and the flag words are OR-ed
    over the ranks once per level (exact: a child's three flags are ORs over footholds).  Every rank returns all valid
    leaves.  -> (centres float32[k, 3], this rank's kernel milliseconds)"""
    from . import _capi, device
    dist, world, rank = _dist_info(group)
    on_device = isinstance(local_footholds, (tuple, list)) and len(local_footholds) == 3 and all(hasattr(t, "is_cuda") and t.is_cuda for t in local_footholds)
    if dist is None or world == 1:
        if on_device:
            return device.apply_oct(local_footholds[0], local_footholds[1], local_footholds[2], leg, settings)
        return _capi.apply_oct(local_footholds, leg, settings)
    exchange = _or_exchange(dist, group)
    if on_device:
        return device.apply_oct_partitioned(local_footholds[0], local_footholds[1], local_footholds[2], leg, settings, exchange)
    return _capi.apply_oct_partitioned(local_footholds, leg, settings, exchange)


def apply_oct_sharded(footholds, leg, settings=None, group=None):
    """apply_oct (lrm_apply_oct) with the children of every octree level dealt round-robin to the ranks: each rank
    evaluates its share on its GPU, the per-child flag words are OR-ed over the ranks once per level (every
    child has one owner, the others contribute 0), and every rank returns all valid leaves.  Each rank holds all
    footholds, as a host array (n, 3) or as three CUDA tensors (x, y, z).  -> (centres float32[k, 3], this rank's kernel milliseconds)"""
    import torch
    from . import _capi, device
    dist, world, rank = _dist_info(group)
    on_device = isinstance(footholds, (tuple, list)) and len(footholds) == 3 and all(hasattr(t, "is_cuda") and t.is_cuda for t in footholds)
    if dist is None or world == 1:
        if on_device:
            return device.apply_oct(footholds[0], footholds[1], footholds[2], leg, settings)
        return _capi.apply_oct(footholds, leg, settings)
    exchange = _or_exchange(dist, group)

    if on_device:  # (x, y, z) CUDA tensors: no host copy of the cloud
        return device.apply_oct(footholds[0], footholds[1], footholds[2], leg, settings, rank, world, exchange)
    return _capi.apply_oct_sharded(footholds, leg, settings, rank, world, exchange)

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

"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") all-reduce over the
flat fp32 gradient buffer; gloo on CPU for tests.

The reference only has single-process nn.DataParallel (var_updown/scripts/train.py:123-124: replicate parameters,
scatter the batch on dim 0, reduce-add gradients to device 0 every iteration) and it crashes for training
(train.py:157,160 dereference the wrapper).  Here every rank holds a full replica and a row shard of the
minibatch; captions are independent units, so the only exchange is sum(grads): the mean over the global batch is
(1/world) * sum_ranks(local-mean grads) when every rank has the same number of rows.  The 1/world factor is folded
into the clip+SGD kernel (`gscale`), so the buffer is reduced in place exactly once.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Row range [lo, hi) of the global minibatch owned by `rank` (equal shards required for exact mean gradients)."""
    if n_rows % world != 0:
        raise ValueError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return rank * per, (rank + 1) * per


def bucket_bounds(numel: int, n_buckets: int, align: int = 64) -> List[Tuple[int, int]]:
    """Contiguous, `align`-float aligned buckets covering [0, numel).  Sized for xGMI: a few large messages
    (>= tens of MB each) rather than per-parameter ones."""
    n_buckets = max(1, min(n_buckets, max(1, numel // align)))
    per = (numel + n_buckets - 1) // n_buckets
    per = (per + align - 1) // align * align
    out, lo = [], 0
    while lo < numel:
        hi = min(numel, lo + per)
        out.append((lo, hi))
        lo = hi
    return out


def allreduce_flat(flat: torch.Tensor, group=None, n_buckets: int = 1, async_op: bool = False):
    """In-place sum all-reduce of a flat buffer, optionally as `n_buckets` independent collectives (so that a later
    bucket's reduction can overlap whatever still produces an earlier one).  Returns the world size, plus the work
    handles when async_op."""
    w = world_size(group)
    if w == 1:
        return (1, []) if async_op else 1
    works = []
    for lo, hi in bucket_bounds(flat.numel(), n_buckets):
        works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    if async_op:
        return w, works
    return w


def gscale(world: int) -> float:
    """Factor the optimiser applies to the summed gradients (mean over ranks)."""
    return 1.0 / float(world)

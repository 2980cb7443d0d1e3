"""Multi-GPU sharding of the Monte-Carlo loop.

Every frame of every cell is independent and its random streams are keyed by the *global*
frame index, so rank g of G simulates the contiguous frame range
[g*F/G, (g+1)*F/G) of every cell and the union over ranks is bit-identical to a
one-GPU run.  The only exchange is one all-reduce(sum) of the integer counters
(``torch.distributed``: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU).

The reference's own parallelism is task-level only: ``parfor`` over window files
(matlab/main_BER_calculation.m:31) and ``Pool.map`` over (system, CP) tuples
(python/wofdm_optimization.py:127-129).
"""
import numpy as np


def frame_shard(total_frames, rank, world_size, frame_offset=0):
    """(offset, count) of rank's contiguous share of ``total_frames`` frames per cell."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    lo = (total_frames * rank) // world_size
    hi = (total_frames * (rank + 1)) // world_size
    return frame_offset + lo, hi - lo


def all_reduce_counts(counts, group=None):
    """Sum the counter tensor over all ranks in place (no-op without an initialised process
    group).  ``counts``: int64 torch tensor (device tensor under nccl/RCCL, CPU under gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def reduce_counts_numpy(counts, group=None):
    """Same for a host uint64 ndarray (goes through a CPU int64 tensor)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(counts).view(np.int64).copy())
    all_reduce_counts(t, group)
    return t.numpy().view(np.uint64).reshape(np.shape(counts))

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


def _backend(group=None):
    """Name of the initialised process group's backend ("nccl" = RCCL, "gloo"), or None."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return str(dist.get_backend(group)).lower()


def all_reduce_counts(counts, group=None):
    """Sum the int64 counter tensor over all ranks in place (no-op without an initialised process
    group).  The tensor is reduced where the backend can reach it: a device tensor directly under
    nccl (= RCCL over xGMI), through a host copy under gloo; a CPU tensor through a device copy
    under nccl."""
    import torch
    import torch.distributed as dist
    be = _backend(group)
    if be is None:
        return counts
    if be == "nccl" and not counts.is_cuda:
        t = counts.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        counts.copy_(t.cpu())
    elif be != "nccl" and counts.is_cuda:
        torch.cuda.synchronize(counts.device)
        t = counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        counts.copy_(t)
    else:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def reduce_counts_numpy(counts, group=None):
    """Same for a host uint64 ndarray (through an int64 tensor; under nccl that tensor lives on the
    current CUDA device for the reduction)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(counts).view(np.int64).copy())
    all_reduce_counts(t, group)
    return t.numpy().view(np.uint64).reshape(np.shape(counts))

"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests).

The path shards by PIXEL (all n blur rays of a pixel, and the start/end event pair of a pixel,
stay on one rank - SURVEY.md 8e).  Every rank draws the same global pixel-index vectors and takes
a contiguous slice; parameters and Adam state are replicated; loss means are over the GLOBAL
batch, so local gradients are already scaled by 1/global_count and ONE sum all-reduce of the
flat gradient buffer per step reproduces the single-GPU gradient.  The L2-normalised event loss
(train.py:238-292) additionally needs the global sums of squares: a 16-double all-reduce between
the two loss passes.
"""
import torch


def shard_indices(idx, rank, world):
    """Contiguous slice [rank*n/world, (rank+1)*n/world) of a global index vector."""
    n = idx.shape[0]
    if n % world != 0:
        raise ValueError("global batch (%d) must be divisible by world size (%d)" % (n, world))
    per = n // world
    return idx[rank * per:(rank + 1) * per].contiguous()


def allreduce_sum_(t, world, group=None):
    """In-place sum over ranks (no-op on one rank).  With the gloo backend (CPU tests, or the
    2-ranks-on-one-GPU equivalence test) device tensors are staged through host memory; the
    production backend is nccl (= RCCL over xGMI), which reduces the device buffer directly."""
    if world > 1:
        if t.is_cuda and torch.distributed.get_backend(group) == "gloo":
            h = t.cpu()
            torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


def broadcast_(t, world, src=0, group=None):
    if world > 1:
        torch.distributed.broadcast(t, src=src, group=group)
    return t

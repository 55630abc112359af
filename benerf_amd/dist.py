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

# Self-test switch: issue every collective even on a world of ONE rank (needs an initialised process group).  A one-rank RCCL
# communicator is the only way to run the device branch below - in-place all-reduce of slices of the flat gradient buffer on
# the communicator's stream, the handle's wait() on a side stream - on a box with a single GPU (tests/test_api_gpu.py).
ALWAYS_COMMUNICATE = False


def _active(world):
    return world > 1 or (ALWAYS_COMMUNICATE and torch.distributed.is_initialized())


def shard_bounds(n, rank, world, uneven=False):
    """[lo, hi) of rank `rank`'s contiguous slice of n items.  uneven: the n % world left-over items go one each to the low ranks
    (a strong-scaled batch that the ranks cannot split evenly - C4's 215 blur pixels over 8 GPUs - is rendered whole)."""
    if n % world != 0 and not uneven:
        raise ValueError("global batch (%d) must be divisible by world size (%d)" % (n, world))
    per, rem = divmod(n, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def balanced_shard_bounds(n_evt, n_rgb, rays_per_evt, rays_per_rgb, world):
    """Ray-balanced contiguous slices of an uneven strong-scaled batch: [((e_lo, e_hi), (r_lo, r_hi))] * world.
    The blur pixels are split as shard_bounds(uneven=True) does (left-over pixels one each to the low ranks); a blur pixel is
    rays_per_rgb rays (the n virtual poses), so ranks with one blur pixel more get FEWER event pixels (rays_per_evt rays each: 2
    poses, or bins + 1), such that every rank renders as close to total / world rays as whole pixels allow (largest-remainder
    rounding, ties to the low rank - every rank computes the same table).  Why it matters beyond balance: the fused MLP kernels
    work in 128-point tiles, one per CU and round - at 1/8 of C2 a rank with 14 instead of 13 blur pixels renders 522 rays =
    261 coarse tiles = TWO rounds of the 256 CUs instead of one (and three instead of two in the fine network): +0.18 ms on a
    1.35 ms step, while 510 +- 1 rays on every rank fit one round (profiles/r05_one_eighth_batch_same_box.log)."""
    rb = [shard_bounds(n_rgb, k, world, uneven=True) for k in range(world)]
    target = (rays_per_evt * n_evt + rays_per_rgb * n_rgb) / world
    ideal = [max((target - rays_per_rgb * (hi - lo)) / rays_per_evt, 0.0) for lo, hi in rb]
    # the ideal shares sum to n_evt unless one was clipped at zero (a rank whose blur pixels alone exceed the mean: tiny batches) -
    # then the event pixels are dealt in proportion to what is left of the ideal shares
    tot = sum(ideal)
    ideal = [x * n_evt / tot for x in ideal] if tot > 0 else [n_evt / world] * world
    e = [min(int(x + 1e-9), n_evt) for x in ideal]
    left = n_evt - sum(e)
    order = sorted(range(world), key=lambda k: (-(ideal[k] - e[k]), k))
    i = 0
    while left > 0:
        e[order[i % world]] += 1
        left -= 1
        i += 1
    while left < 0:     # float round-off only
        k = max(range(world), key=lambda j: (e[j], -j))
        e[k] -= 1
        left += 1
    out, lo = [], 0
    for k in range(world):
        out.append(((lo, lo + e[k]), rb[k]))
        lo += e[k]
    return out


def shard_indices(idx, rank, world, uneven=False):
    """Contiguous slice of a global index vector for this rank (shard_bounds)."""
    lo, hi = shard_bounds(idx.shape[0], rank, world, uneven)
    return idx[lo:hi].contiguous()


def _through_host(t, group):
    """True when a device tensor has to be staged through host memory: the group has no device backend.  The backend
    may be a plain name ("gloo", "nccl") or a per-device map ("cpu:gloo,cuda:nccl"): anything that names nccl (= RCCL
    on ROCm) reduces device buffers in place; a custom backend that does not is treated like gloo (slow but correct)."""
    return t.is_cuda and "nccl" not in str(torch.distributed.get_backend(group)).lower()


def allreduce_sum_(t, world, group=None):
    """In-place sum over ranks (no-op on one rank).  With the gloo backend (CPU tests, or the
    2-ranks-on-one-GPU equivalence test) device tensors are staged through host memory; the
    production backend is nccl (= RCCL over xGMI), which reduces the device buffer directly."""
    if _active(world):
        if _through_host(t, group):
            h = t.cpu()
            torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


class _Done:
    def wait(self):
        return True


def allreduce_sum_async_(t, world, group=None):
    """Starts the in-place sum over ranks and returns a handle whose wait() makes the CURRENT stream wait for it.
    With nccl (= RCCL) the collective runs on the communicator's own stream behind everything already queued on the
    current stream, so kernels launched after this call overlap with it: TrainStep issues the fine network's bucket
    as soon as its weight gradients are complete and computes the coarse backward meanwhile.  gloo (CPU
    tests / several ranks on one GPU) has no device path: reduced synchronously through host memory."""
    if not _active(world):
        return _Done()
    if _through_host(t, group):      # blocking round trip through host memory: no overlap with the kernels that follow
        allreduce_sum_(t, world, group)
        return _Done()
    return torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group, async_op=True)


def broadcast_(t, world, src=0, group=None):
    """Rank `src`'s values everywhere (parameters + optimiser state at start-up: replicas must not rely on seeds)."""
    if _active(world):
        if _through_host(t, group):
            h = t.cpu()
            torch.distributed.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            torch.distributed.broadcast(t, src=src, group=group)
    return t

"""CPU, world_size 2, gloo: the data-parallel decomposition used by engine.TrainStep
(pixel sharding + global-count loss scaling + sum all-reduce of stats and gradients) reproduces
the single-process gradient.  The per-rank arithmetic here is the oracle (the HIP kernels need a
GPU); what is under test is benerf_amd.dist and the normalisation scheme."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_backward(O, rgb_e, rgb0_e, tgt_acc, rgb_r, rgb0_r, tgt_rgb, C, dataset, thr, P, Re_g, Rr_g, stats_reduce):
    """Backward of the GLOBAL loss from one rank's shard, split the way kernels K6 split it:
    pass 1 local sums -> all-reduce -> pass 2 closed-form d loss / d diff with global counts and
    norms (benerf_amd/csrc/loss.hip), then autograd through the local render."""
    Re = tgt_acc.shape[0]

    def diff(img):
        a, b = img[:Re], img[Re:]
        if C == 3:
            a, b = O.to_gray(a), O.to_gray(b)
        return O.bright_log(b, dataset) - O.bright_log(a, dataset)

    df, dc = diff(rgb_e), diff(rgb0_e)
    t = tgt_acc * (thr if thr > 0 else 1.0)
    stats = torch.stack([(df.detach() ** 2).sum(), (df.detach() * t).sum(), (dc.detach() ** 2).sum(),
                         (dc.detach() * t).sum(), (t ** 2).sum()]).double()
    stats_reduce(stats)
    outs, grads = [], []
    for d, s_dd, s_dt in ((df, stats[0], stats[1]), (dc, stats[2], stats[3])):
        if thr > 0:
            g = 0.1 * 2.0 * (d.detach() - t) / Re_g
        else:
            n_d, n_t = s_dd.sqrt(), stats[4].sqrt()
            sc_d, sc_t = n_d + 1e-9, n_t + 1e-9
            gi = 2.0 * 2.0 * (d.detach() / sc_d - t / sc_t) / Re_g
            sum_gd = 2.0 * 2.0 * (s_dd / sc_d - s_dt / sc_t) / Re_g
            g = gi / sc_d - d.detach() / (sc_d * sc_d * n_d) * sum_gd
        outs.append(d)
        grads.append(g.float())
    Rr = tgt_rgb.shape[0]
    for img in (rgb_r, rgb0_r):
        b = sum(img[j * Rr:(j + 1) * Rr] for j in range(P)) / P
        outs.append(b)
        grads.append((2.0 * (b.detach() - tgt_rgb) / (Rr_g * C)).float())
    torch.autograd.backward(outs, grads)


def _worker(rank, world, port, thr, q, Rr_g=8, uneven=False, balanced=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import benerf_oracle as O
    from benerf_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    rng = np.random.default_rng(0)           # identical on every rank
    C, P, Re_g, dataset = 3, 5, 16, "E2NeRF_Real" if thr <= 0 else "BeNeRF_Unreal"
    W = torch.from_numpy(rng.standard_normal((C, 8)).astype(np.float32)).requires_grad_(True)   # stand-in "network"
    feat_e = torch.from_numpy(rng.standard_normal((2, Re_g, 8)).astype(np.float32))
    feat_r = torch.from_numpy(rng.standard_normal((P, Rr_g, 8)).astype(np.float32))
    acc = torch.from_numpy(rng.integers(-3, 4, (Re_g, 1)).astype(np.float32))
    tgt = torch.from_numpy(rng.random((Rr_g, C)).astype(np.float32))

    def render(feat):     # [poses, pixels, 8] -> pose-major [poses*pixels, C] in (0,1)
        return torch.sigmoid(feat @ W.t()).reshape(-1, C)

    if balanced:      # TrainStep(uneven_shards=True): blur left-overs to the low ranks, event pixels dealt to equalise the RAY counts
        (e0, e1), (r0, r1) = dist.balanced_shard_bounds(Re_g, Rr_g, 2, P, world)[rank]
        pix_e, pix_r = torch.arange(e0, e1), torch.arange(r0, r1)
    else:
        pix_e = dist.shard_indices(torch.arange(Re_g), rank, world, uneven)
        pix_r = dist.shard_indices(torch.arange(Rr_g), rank, world, uneven)
    _local_backward(O, render(feat_e[:, pix_e]), render(feat_e[:, pix_e] * 0.9), acc[pix_e], render(feat_r[:, pix_r]),
                    render(feat_r[:, pix_r] * 1.1), tgt[pix_r], C, dataset, thr, P, Re_g, Rr_g,
                    lambda s: dist.allreduce_sum_(s, world))
    g = W.grad.clone()
    if world == 1:   # single rank: the closed form must equal plain autograd of the reference loss
        W2 = W.detach().clone().requires_grad_(True)

        def render2(feat):
            return torch.sigmoid(feat @ W2.t()).reshape(-1, C)
        le, _, _ = O.event_loss(render2(feat_e), render2(feat_e * 0.9), Re_g, acc.double(), C, dataset, thr, 0.1, 2.0)
        lr, _, _ = O.blur_loss(render2(feat_r), render2(feat_r * 1.1), tgt, P, 1.0)
        (le + lr).backward()
        np.testing.assert_allclose(g.numpy(), W2.grad.numpy(), rtol=2e-4, atol=1e-7)
    dist.allreduce_sum_(g, world)
    # the range guard's verdict rides the trajectory bucket as a float (engine.TrainStep): ranks 2 and 5 of 8 flag a violation,
    # EVERY rank must see a non-zero sum (= skip the step), and the same count
    flag = torch.tensor([1.0 if rank in (2, 5) else 0.0])
    h = dist.allreduce_sum_async_(flag, world)
    h.wait()
    assert float(flag) == float(sum(1 for r in range(world) if r in (2, 5))), (rank, float(flag))
    if rank == 0:
        q.put(g.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("thr", [0.1, -1.0])
def test_sharded_gradient_equals_single_rank(thr):
    """world sizes 2, 4 and 8 against the single process (C4 / C5 run on 8 ranks; two of the eight ranks flag a
    range-guard violation and every rank sees the same non-zero verdict)."""
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2, 4, 8):
        q = ctx.Queue()
        port = 29500 + int(abs(thr) * 10) + world * 7 + (os.getpid() % 200)
        procs = [ctx.Process(target=_worker, args=(r, world, port, thr, q)) for r in range(world)]
        for p in procs:
            p.start()
        out[world] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    for world in (2, 4, 8):
        np.testing.assert_allclose(out[world], out[1], rtol=2e-5, atol=1e-7)


def test_uneven_shards_equal_single_rank():
    """A global batch the ranks cannot split evenly (11 blur pixels, 16 event pixels over 8 and over 3 ranks; C4 strong-scaled:
    215 blur pixels over 8 GPUs): the left-over pixels go one each to the low ranks (dist.shard_bounds(uneven=True)), the loss
    means use the global counts, and the summed gradient equals the single process's.  World 8 runs the ray-balanced table of
    TrainStep(uneven_shards=True) (dist.balanced_shard_bounds: ranks with a blur pixel more take fewer event pixels), world 3 the
    plain per-vector split."""
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 3, 8):
        q = ctx.Queue()
        port = 29900 + world * 5 + (os.getpid() % 90)
        procs = [ctx.Process(target=_worker, args=(r, world, port, -1.0, q, 11, True, world == 8)) for r in range(world)]
        for p in procs:
            p.start()
        out[world] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    for world in (3, 8):
        np.testing.assert_allclose(out[world], out[1], rtol=2e-5, atol=1e-7)


def test_shard_indices():
    sys.path.insert(0, ROOT)
    from benerf_amd import dist
    idx = torch.arange(12)
    parts = [dist.shard_indices(idx, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts), idx)
    with pytest.raises(ValueError):
        dist.shard_indices(torch.arange(10), 0, 4)
    # uneven: nothing dropped, nothing twice, the low ranks take the left-over items
    parts = [dist.shard_indices(torch.arange(215), r, 8, uneven=True) for r in range(8)]
    assert torch.equal(torch.cat(parts), torch.arange(215)) and [len(p_) for p_ in parts] == [27] * 7 + [26]
    assert dist.shard_bounds(5, 7, 8, uneven=True) == (5, 5)
    # ray-balanced table (TrainStep(uneven_shards=True)): contiguous, complete, blur pixels as above, every rank within one event
    # pixel's rays of the mean - C2 / C4 / C5 over 8 ranks stay inside whole rounds of 128-point tiles (<= 512 / 1024 rays)
    for ne, nr, pe, pn, cap in ((1024, 107, 2, 19, 512), (2048, 215, 2, 19, 1024), (2048, 132, 2, 31, 1024), (16, 11, 2, 5, None), (31, 5, 2, 5, None)):
        for world in (2, 3, 8):
            tab = dist.balanced_shard_bounds(ne, nr, pe, pn, world)
            assert tab[0][0][0] == 0 and tab[0][1][0] == 0 and tab[-1][0][1] == ne and tab[-1][1][1] == nr
            assert all(tab[k][0][1] == tab[k + 1][0][0] and tab[k][1][1] == tab[k + 1][1][0] for k in range(world - 1))
            assert [r1 - r0 for _, (r0, r1) in tab] == [dist.shard_bounds(nr, k, world, uneven=True)[1] - dist.shard_bounds(nr, k, world, uneven=True)[0]
                                                        for k in range(world)]
            rays = [pe * (e1 - e0) + pn * (r1 - r0) for (e0, e1), (r0, r1) in tab]
            assert max(rays) - min(rays) <= pe + 1 or min(e1 - e0 for (e0, e1), _ in tab) == 0, (ne, nr, world, rays)
            if cap is not None and world == 8:
                assert max(rays) <= cap, rays
    # dense event bins (BASELINE.json configs[4]): B + 1 event poses per event pixel - rays per event pixel = B + 1.  C5 over 8 ranks
    # at B = 4 and 8: a partition, blur pixels as above, every rank within one event pixel's rays (B + 1) of the mean, nobody empty
    for bins in (4, 8):
        pe = bins + 1
        for ne, nr, pn in ((2048, 132, 31), (1024, 107, 19)):
            tab = dist.balanced_shard_bounds(ne, nr, pe, pn, 8)
            assert tab[0][0][0] == 0 and tab[0][1][0] == 0 and tab[-1][0][1] == ne and tab[-1][1][1] == nr
            assert all(tab[k][0][1] == tab[k + 1][0][0] and tab[k][1][1] == tab[k + 1][1][0] for k in range(7))
            rays = [pe * (e1 - e0) + pn * (r1 - r0) for (e0, e1), (r0, r1) in tab]
            assert max(rays) - min(rays) <= pe + 1, (bins, ne, nr, rays)
            assert min(e1 - e0 for (e0, e1), _ in tab) > 0 and min(r1 - r0 for _, (r0, r1) in tab) > 0
            assert sum(rays) == pe * ne + pn * nr
    # degenerate batches (a rank whose blur pixels alone exceed the mean share of rays; no event or no blur pixels): still a partition
    for ne, nr, pe, pn, world in ((1, 5, 2, 19, 4), (2, 20, 2, 19, 4), (0, 7, 2, 19, 3), (5, 0, 2, 19, 3)):
        tab = dist.balanced_shard_bounds(ne, nr, pe, pn, world)
        assert tab[0][0][0] == 0 and tab[-1][0][1] == ne and tab[-1][1][1] == nr
        assert all(tab[k][0][1] == tab[k + 1][0][0] and tab[k][1][1] == tab[k + 1][1][0] for k in range(world - 1))
        assert all(e1 >= e0 and r1 >= r0 for (e0, e1), (r0, r1) in tab)


def test_async_allreduce_and_broadcast_single_process():
    """world 1: the asynchronous exchange and the start-up broadcast are no-ops with a waitable handle."""
    sys.path.insert(0, ROOT)
    from benerf_amd import dist
    t = torch.arange(5.0)
    h = dist.allreduce_sum_async_(t, 1)
    assert h.wait() and torch.equal(t, torch.arange(5.0))
    assert dist.broadcast_(t, 1) is t


def test_backend_detection_for_device_tensors(monkeypatch):
    """dist picks the host-staged path from the group's backend description, whatever its spelling: plain names, per-device
    maps ("cpu:gloo,cuda:nccl") and unknown backends (treated like gloo: correct, if slow)."""
    sys.path.insert(0, ROOT)
    from benerf_amd import dist

    class FakeCuda:
        is_cuda = True

    class FakeCpu:
        is_cuda = False

    for be, want in (("gloo", True), ("nccl", False), ("cpu:gloo,cuda:nccl", False), ("cpu:gloo", True), ("NCCL", False),
                     ("custom_thing", True)):
        monkeypatch.setattr(torch.distributed, "get_backend", lambda group=None, be=be: be)
        assert dist._through_host(FakeCuda(), None) is want, be
        assert dist._through_host(FakeCpu(), None) is False

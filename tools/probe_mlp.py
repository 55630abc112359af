#!/usr/bin/env python
"""Times the three fused-MLP launches in isolation at a given point count (HIP events on the
launch stream).  Development probe; bench.py is the measurement of record."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4081)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--mode", default="f32", choices=["f32", "split"])
    a = ap.parse_args()
    import benerf_oracle as O
    from benerf_amd import kernels as K, workloads as WL
    dev = torch.device("cuda", 0)
    K.set_mlp_precision(a.mode)
    rng = np.random.default_rng(0)
    p = O.xavier_params(rng, a.channels)
    net = K.PackedMlp([p[n + ".weight"].to(dev) for n in K.LAYER_NAMES], [p[n + ".bias"].to(dev) for n in K.LAYER_NAMES],
                      a.channels)
    net.pack()
    N, S, C = a.rays, a.samples, a.channels
    M = N * S
    ro = torch.rand(N, 3, device=dev) - 0.5
    rd = torch.rand(N, 3, device=dev) - 0.5
    vd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    z = torch.sort(torch.rand(N, S, device=dev), -1)[0]
    d_raw = torch.randn(M, C + 1, device=dev)
    gw = [torch.zeros_like(w) for w in net.weights]
    gb = [torch.zeros_like(b) for b in net.biases]
    fpp = WL.mlp_flops_per_point(C)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
        ev[0].record()
        for i in range(a.iters):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
        return ts[len(ts) // 2], ts[0]

    raw, acts = K.mlp_fwd(net, ro, rd, vd, z, True)
    res = {}
    res["fwd_infer"] = timed(lambda: K.mlp_fwd(net, ro, rd, vd, z, False))
    res["fwd_train"] = timed(lambda: K.mlp_fwd(net, ro, rd, vd, z, True))
    K.TIMERS.enabled = True
    K.TIMERS.records.clear()
    for _ in range(a.iters):
        K.mlp_bwd(net, d_raw, acts, N, S, gw, gb, False)
    torch.cuda.synchronize()
    K.TIMERS.enabled = False
    for name, (n, ms, pts) in K.TIMERS.summary().items():
        res[name] = (ms / n, ms / n)
    for k, (med, mn) in res.items():
        print("[%s] %-12s M=%d  median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (algorithmic %d flop/pt)"
              % (a.mode, k, M, med, mn, M * fpp / (med * 1e-3) / 1e12, fpp))


if __name__ == "__main__":
    main()

"""Numerical experiment (CPU, oracle only - not product code): how far do the gradients of one training step move
when the BACKWARD GEMMs of the MLP use f16-rounded operands (f32 accumulate) while the forward pass stays exact?

variants   dX = dY W            dW = dY^T X
  exact    f32                  f32
  A        q(dY) q(W)           q(dY)^T q(X)
  B        q(dY) W              q(dY)^T q(X)
  C        f32                  q(dY)^T q(X)
q = round to f16 after a per-64-row power-of-two scale (the dX kernels' tile scaling).
Reference = the same step in float64."""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, "oracle")
import benerf_oracle as O  # noqa: E402

MODE = {"v": "exact"}
ZCACHE = {"record": True, "z": [], "i": 0}
_fine_depths = O.fine_depths


def fine_depths_replay(z, weights, u, exact=False):
    """the first (f32 exact) run records the fine depths; every other run replays them, so that all runs differentiate
    the same function (sample_pdf is ill-conditioned in its inputs)"""
    if ZCACHE["record"]:
        out = _fine_depths(z, weights, u, exact)
        ZCACHE["z"].append(out)
        return out
    za, zs = ZCACHE["z"][ZCACHE["i"]]
    ZCACHE["i"] += 1
    return za.to(z.dtype), zs.to(z.dtype)


O.fine_depths = fine_depths_replay


def q16(v, per_rows=None):
    if per_rows is None:
        return v.half().float()
    n = v.shape[0]
    pad = (-n) % per_rows
    vp = torch.cat([v, v.new_zeros(pad, v.shape[1])]) if pad else v
    t = vp.view(-1, per_rows, v.shape[1])
    mx = t.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
    s = torch.exp2(-4.0 - torch.floor(torch.log2(mx)))
    return ((t * s).half().float() / s).view(-1, v.shape[1])[:n]


class Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        m = MODE["v"]
        if m == "exact" or dy.dtype == torch.float64:
            return dy @ w, dy.t() @ x, dy.sum(0)
        dyq = q16(dy, 64)
        xq = q16(x)
        if m == "A":
            dx = dyq @ q16(w)
        elif m == "B":
            dx = dyq @ w
        else:
            dx = dy @ w
        return dx, dyq.t() @ xq, dyq.sum(0)


def patched_linear(x, w, b):
    return Lin.apply(x, w, b)


def run(seed, dtype, mode, cfg, sizes):
    MODE["v"] = mode
    rng = np.random.default_rng(seed)
    Re, Rr = sizes
    pc = {k: v.to(dtype).requires_grad_(True) for k, v in O.xavier_params(rng, cfg.channels).items()}
    pf = {k: v.to(dtype).requires_grad_(True) for k, v in O.xavier_params(rng, cfg.channels).items()}
    for p in (pc, pf):   # "trained-like": open up density so that weights/pose grads are not dominated by the far plane
        p["alpha_linear.bias"].data += 2.0
    knots = torch.from_numpy(rng.uniform(0, 0.01, (4, 6))).to(dtype).requires_grad_(True)
    transform = torch.zeros(1, 6, dtype=dtype, requires_grad=True)
    idx_e = torch.from_numpy(rng.permutation(cfg.H * cfg.W)[:Re])
    idx_r = torch.from_numpy(rng.permutation(cfg.H * cfg.W)[:Rr])
    tacc = torch.from_numpy(rng.integers(-3, 4, Re).astype(np.float64)).to(dtype)
    trgb = torch.from_numpy(rng.random((Rr, cfg.channels))).to(dtype)

    def draws(n):
        S, Ni = cfg.n_samples, cfg.n_importance
        return {"t_rand": torch.from_numpy(rng.random((n, S))).to(dtype), "noise0": torch.from_numpy(rng.standard_normal((n, S))).to(dtype),
                "u": torch.from_numpy(rng.random((n, Ni)).astype(np.float32)), "noise1": torch.from_numpy(rng.standard_normal((n, S + Ni))).to(dtype)}
    de, dr = draws(2 * Re), draws(cfg.n_poses * Rr)
    ts = torch.tensor([0.0, 1.0], dtype=dtype)
    old = torch.nn.functional.linear
    torch.nn.functional.linear = patched_linear
    try:
        loss, _ = O.step_loss(cfg, pc, pf, knots, transform, ts * 0.1 + 0.3, ts, idx_e, idx_r, tacc, trgb, de, dr, exact_pdf=True)
        loss.backward()
    finally:
        torch.nn.functional.linear = old
    g = {"knots": knots.grad, "transform": transform.grad}
    for tag, p in (("c", pc), ("f", pf)):
        for k, v in p.items():
            g[tag + "." + k] = v.grad
    return float(loss), {k: v.double() for k, v in g.items()}


def main():
    torch.set_default_dtype(torch.float32)
    old_default = torch.get_default_dtype()
    cfg = O.StepConfig(n_samples=32, n_importance=32, n_poses=9)
    sizes = (96, 20)
    for seed in (0, 1):
        ZCACHE.update(record=True, z=[], i=0)
        run(seed, torch.float32, "exact", cfg, sizes)
        ZCACHE.update(record=False, i=0)
        torch.set_default_dtype(torch.float64)
        l64, g64 = run(seed, torch.float64, "exact", cfg, sizes)
        torch.set_default_dtype(old_default)
        print("seed %d  loss(f64) %.6f" % (seed, l64))
        g32 = None
        for mode in ("exact", "C", "B", "A"):
            ZCACHE["i"] = 0
            l, g = run(seed, torch.float32, mode, cfg, sizes)
            if mode == "exact":
                g32 = g
            else:   # against the exact-f32 backward of the SAME forward pass (same masks): the arithmetic's own error
                keys = [k for k in g if k.endswith("bias") or k in ("knots", "transform")] + [k for k in g if k.endswith("weight")]
                worst = sorted(((float((g[k] - g32[k]).abs().max() / g32[k].abs().max()), k) for k in keys), reverse=True)[:4]
                print(" mode %-5s vs exact-f32 backward, worst max-relative errors: %s" % (mode, ", ".join("%s %.1e" % (k, e) for e, k in worst)))
                print("            knots %.1e transform %.1e" % (float((g["knots"] - g32["knots"]).abs().max() / g32["knots"].abs().max()),
                                                                 float((g["transform"] - g32["transform"]).abs().max() / g32["transform"].abs().max())))
            rows = []
            for k in ("knots", "transform", "c.pts_linears.0.weight", "c.pts_linears.0.bias", "c.pts_linears.4.weight",
                      "c.pts_linears.7.bias", "f.pts_linears.0.weight", "f.pts_linears.0.bias", "f.pts_linears.5.weight",
                      "f.views_linears.0.weight", "f.rgb_linear.weight", "f.alpha_linear.weight"):
                ref = g64[k]
                err_max = float((g[k] - ref).abs().max() / ref.abs().max())
                err_norm = abs(float(g[k].norm() / ref.norm()) - 1.0)
                rows.append("%s max %.1e norm %.1e" % (k.replace("pts_linears", "L").replace("weight", "w").replace("bias", "b"), err_max, err_norm))
            print(" mode %-5s loss %.6f" % (mode, l))
            for i in range(0, len(rows), 3):
                print("    " + " | ".join(rows[i:i + 3]))


if __name__ == "__main__":
    main()

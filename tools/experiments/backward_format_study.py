"""Numerical experiment (CPU, oracle only - not product code): which operand formats does the MLP BACKWARD need so that its
error against float64 (same ReLU masks) is the exact-f32 backward's own?  Mirrors tests/test_f64_truth_gpu.py::
test_mlp_backward_arithmetic_vs_float64 (one network, identical points, fixed upstream gradient, masks forced), with the
backward GEMMs emulated in torch:

  dX chain   dY_(l-1) = mask * (q_g(dY_l) @ q_w(W_l))        q_g: gradient operand format, q_w: weight operand format
  dW         dW_l     = q_y(dY_l)^T @ q_x(X_l)               q_y / q_x: the STORED gradient / activation formats

formats:  "f32"  exact;  "h" one f16 (11 bits);  "hl" hi + lo, two f16, lo unscaled (the product drops lo x lo);
          "h8"   f16 + an 8-bit residual (19 bits);  "h4" f16 + 4-bit residual (15 bits)
Gradients are scaled per 128-point tile (dX chain) / per call (stored dY) by a power of two so that the maximum lies in
[2^6, 2^7) before rounding (mlp_split.h, pow2_scale6)."""
import itertools
import sys

import numpy as np
import torch

sys.path.insert(0, "oracle")
import benerf_oracle as O  # noqa: E402
import golden_inputs as GI  # noqa: E402

CFG = {"g": "f32", "w": "f32", "y": "f32", "x": "f32"}


def split(v, fmt):
    """-> list of (part, ) f32 tensors whose sum represents v in format fmt; [hi] or [hi, lo]"""
    if fmt == "f32":
        return [v]
    hi = v.half().float()
    if fmt == "h":
        return [hi]
    r = v - hi
    if fmt == "hl":
        return [hi, r.half().float()]
    bits = {"h8": 8, "h4": 4}[fmt]
    # residual in units of ulp(hi) / 2^bits; ulp(hi) = 2^(e - 10) for normal hi
    e = torch.floor(torch.log2(hi.abs().clamp_min(2.0 ** -14)))
    ulp = torch.exp2(e - 10)
    q = torch.round(r / ulp * (1 << bits)) / (1 << bits) * ulp
    return [hi, q]


def pow2_scale(mx):
    mx = mx.clamp_min(2.0 ** -119)
    return torch.exp2(6.0 - torch.floor(torch.log2(mx)))


def scaled_split(dy, fmt, per_rows):
    """gradient operand: power-of-two scale per `per_rows` rows (None: one for the call), split, unscale"""
    if fmt == "f32":
        return [dy]
    n = dy.shape[0]
    if per_rows is None:
        s = pow2_scale(dy.abs().max())
        return [p / s for p in split(dy * s, fmt)]
    pad = (-n) % per_rows
    vp = torch.cat([dy, dy.new_zeros(pad, dy.shape[1])]) if pad else dy
    t = vp.view(-1, per_rows, dy.shape[1])
    s = pow2_scale(t.abs().amax(dim=(1, 2), keepdim=True))
    return [(p / s).view(-1, dy.shape[1])[:n] for p in split(t * s, fmt)]


def prod(a_parts, b_parts, f):
    """sum of the part products, dropping lo x lo; every matmul accumulates in f32 (f returns a @ b-like result)"""
    out = f(a_parts[0], b_parts[0])
    if len(b_parts) > 1:
        out = out + f(a_parts[0], b_parts[1])
    if len(a_parts) > 1:
        out = out + f(a_parts[1], b_parts[0])
    return out


class Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if dy.dtype == torch.float64 or dy.shape[1] < 8:        # heads (alpha, rgb) run on the VALU in f32
            return dy @ w, dy.t() @ x, dy.sum(0)
        dx = prod(scaled_split(dy, CFG["g"], 128), split(w, CFG["w"]), lambda a, b: a @ b)
        yp = scaled_split(dy, CFG["y"], None)
        dw = prod(yp, split(x, CFG["x"]), lambda a, b: a.t() @ b)
        return dx, dw, sum(yp).sum(0)


def main():
    n_rays_list = [int(a) for a in sys.argv[1:]] or [8, 128]
    for n_rays in n_rays_list:
        rng = np.random.default_rng(91)
        C, N, S = 1, n_rays, 128
        p = O.xavier_params(rng, C)
        p["alpha_linear.bias"] += 1.0
        ro = GI.f32(rng.uniform(-0.3, 0.3, (N, 3)))
        rd = GI.f32(rng.uniform(-1, 1, (N, 3)))
        vd = torch.nn.functional.normalize(GI.f32(rng.standard_normal((N, 3))), dim=-1)
        z = GI.f32(np.sort(rng.random((N, S)), -1))
        noise = GI.f32(rng.standard_normal((N, S)))
        target = GI.f32(rng.random((N, C)))
        pts32 = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
        raw32, acts = O.mlp_forward({k: v.clone() for k, v in p.items()}, pts32, vd, want_acts=True)
        raw32 = raw32.detach().requires_grad_(True)
        rgb = O.composite(raw32, z, rd, noise, C)[0]
        ((rgb - target) ** 2).mean().backward()
        d_raw = raw32.grad.reshape(-1, C + 1).contiguous()
        masks = {k: (acts[k] > 0).to(torch.float64) for k in ["h%d" % i for i in range(8)] + ["hv"]}

        def grads(dtype):
            old_d = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            old = torch.nn.functional.linear
            torch.nn.functional.linear = lambda x, w, b: Lin.apply(x, w, b)
            try:
                q = {k: v.to(dtype).clone().requires_grad_(True) for k, v in p.items()}
                pts = pts32.to(dtype).clone().requires_grad_(True)
                raw = O.mlp_forward(q, pts, vd.to(dtype), relu_masks=masks)
                (raw.reshape(-1, C + 1) * d_raw.to(dtype)).sum().backward()
                out = {k: v.grad.double() for k, v in q.items()}
                out["d_pts"] = pts.grad.reshape(-1, 3).double()
                return out
            finally:
                torch.nn.functional.linear = old
                torch.set_default_dtype(old_d)

        g64 = grads(torch.float64)
        print("== %d points" % (N * S))
        variants = [("f32", "f32", "f32", "f32"), ("h", "hl", "h", "h"), ("hl", "hl", "hl", "hl"), ("hl", "hl", "hl", "h"),
                    ("hl", "hl", "h", "hl"), ("hl", "hl", "h", "h"), ("hl", "hl", "h8", "h8"), ("hl", "hl", "h4", "h4"),
                    ("hl", "hl", "hl", "h8"), ("hl", "hl", "hl", "h4"), ("h8", "hl", "h8", "h8"), ("hl", "h", "hl", "hl")]
        for g_, w_, y_, x_ in variants:
            CFG.update(g=g_, w=w_, y=y_, x=x_)
            g = grads(torch.float32)
            worst_w = max(((float((g[k] - g64[k]).abs().max() / g64[k].abs().max()), k) for k in g if k.endswith("weight")))
            worst_b = max(((float((g[k] - g64[k]).abs().max() / g64[k].abs().max()), k) for k in g if k.endswith("bias")))
            worst_n = max(((abs(float(g[k].norm() / g64[k].norm()) - 1.0), k) for k in g if k.endswith("weight")))
            e_pts = float((g["d_pts"] - g64["d_pts"]).abs().max() / g64["d_pts"].abs().max())
            print("dX: g=%-3s w=%-3s | dW: y=%-3s x=%-3s | weights max %.2e (%s) bias max %.2e (%s) norm %.2e | d_pts %.2e" %
                  (g_, w_, y_, x_, worst_w[0], worst_w[1].replace("_linears", "").replace(".weight", ""), worst_b[0],
                   worst_b[1].replace("_linears", "").replace(".bias", ""), worst_n[0], e_pts))


if __name__ == "__main__":
    main()

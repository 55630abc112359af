"""Float64 'truth' of one training step, and the float32 oracle step beside it, in pixel chunks.

TEST INFRASTRUCTURE ONLY (lives under oracle/; imported by tests/ and tools/experiments/ - never by benerf_amd/).

Why: the fused MLP kernels' default arithmetic ('split': f16 MFMA operands, f32 accumulation) is not the reference's
fp32 arithmetic.  The only arithmetic-neutral yardstick for "no worse than the reference's own fp32" is the same step
evaluated in float64: for every gradient, err(HIP vs f64) is compared with err(fp32 oracle vs f64).

How: `step_grads` evaluates benerf_oracle.step_loss (train.py:160-340 restated) over chunks of pixels and accumulates
the gradients with loss.backward() per chunk - the loss terms are means over pixels (train.py:207-236, 299-331), so a
chunk contributes its mean times (chunk pixels / batch pixels); the L2-normalised event loss (train.py:238-292) is not
separable and is refused.  Chunks bound the autograd graph (a full C2 step holds ~16 GB in fp32, twice that in f64).
The float64 evaluation is handed the float32 evaluation's coarse and fine depths (`z_forced`): sample_pdf turns 1e-7
differences of the coarse weights into different samples, so without forcing the two would not differentiate the same
function.  Ray order of a render is pose-major [P * R] (model/nerf.py:241-254); `pose_major_rows` maps a pixel chunk to
its rows.
"""
import contextlib

import numpy as np
import torch

import benerf_oracle as O


@contextlib.contextmanager
def default_dtype(dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def pose_major_rows(n_poses, n_pix, lo, hi):
    """Row indices, in a pose-major [n_poses * n_pix] ray batch, of pixels [lo, hi) - ordered pose-major over the chunk."""
    return (torch.arange(n_poses)[:, None] * n_pix + torch.arange(lo, hi)[None, :]).reshape(-1)


def _chunk_draws(d, rows, dtype):
    return {k: (None if v is None else v[rows].to(dtype if k != "u" else v.dtype)) for k, v in d.items()}


def step_grads(cfg, pc, pf, knots, transform, evt_ts, rgb_ts, idx_evt, idx_rgb, target_acc, target_rgb, draws_evt, draws_rgb,
               dtype=torch.float32, z_forced=None, n_chunks=1, exact_pdf=False, force_inputs=True):
    """One training step's loss and gradients in `dtype`.

    Inputs are the float32 / int64 tensors of a step (draws: dicts over the WHOLE batch, pose-major rows).
    z_forced: None, or the dict this function returned under "z" in another precision: per render the coarse depths, the
    fine depths and the network inputs (pts_coarse, pts_fine, viewdirs); the depths replace this evaluation's own, the
    network inputs are substituted with a straight-through gradient (benerf_oracle.render, mlp_inputs_forced) unless
    force_inputs is False.
    Returns dict(loss=float, grads={name: float64 tensor}, z={"evt": (...), "rgb": (...)} float32)."""
    if cfg.threshold <= 0 and n_chunks != 1:
        raise ValueError("the L2-normalised event loss is not a sum over pixels: no chunked evaluation")
    Re, Rr, P = idx_evt.shape[0], idx_rgb.shape[0], cfg.n_poses
    S, F = cfg.n_samples, cfg.n_samples + cfg.n_importance
    with default_dtype(dtype):
        leaf = lambda t: t.detach().to(dtype).clone().requires_grad_(True)   # noqa: E731
        qc = {k: leaf(v) for k, v in pc.items()}
        qf = {k: leaf(v) for k, v in pf.items()}
        kn, tr = leaf(knots), leaf(transform)
        z_out = {k: tuple(torch.empty(shape, dtype=torch.float32) for shape in ((n, S), (n, F), (n, S, 3), (n, F, 3), (n, 3)))
                 for k, n in (("evt", 2 * Re), ("rgb", P * Rr))}
        total = 0.0
        eb = np.linspace(0, Re, n_chunks + 1).astype(int)
        rb = np.linspace(0, Rr, n_chunks + 1).astype(int)
        for c in range(n_chunks):
            e0, e1, r0, r1 = int(eb[c]), int(eb[c + 1]), int(rb[c]), int(rb[c + 1])
            rows_e, rows_r = pose_major_rows(2, Re, e0, e1), pose_major_rows(P, Rr, r0, r1)
            zf_e = zf_r = in_e = in_r = None
            if z_forced is not None:
                zf_e = tuple(t[rows_e].to(dtype) for t in z_forced["evt"][:2])
                zf_r = tuple(t[rows_r].to(dtype) for t in z_forced["rgb"][:2])
                if force_inputs:
                    in_e = tuple(t[rows_e].to(dtype) for t in z_forced["evt"][2:])
                    in_r = tuple(t[rows_r].to(dtype) for t in z_forced["rgb"][2:])
            loss, parts = O.step_loss(cfg, qc, qf, kn, tr, evt_ts, rgb_ts, idx_evt[e0:e1], idx_rgb[r0:r1],
                                      target_acc[e0:e1].to(torch.float64), target_rgb[r0:r1].to(dtype),
                                      _chunk_draws(draws_evt, rows_e, dtype), _chunk_draws(draws_rgb, rows_r, dtype),
                                      exact_pdf=exact_pdf, z_forced_evt=zf_e, z_forced_rgb=zf_r, want_extras=True,
                                      inputs_forced_evt=in_e, inputs_forced_rgb=in_r)
            part = parts["event"] * ((e1 - e0) / Re) + parts["rgb"] * ((r1 - r0) / Rr)
            part.backward()
            total += float(part.detach())
            for key, rows, ex in (("evt", rows_e, parts["extras_evt"]), ("rgb", rows_r, parts["extras_rgb"])):
                for j, name in enumerate(("z_coarse", "z_fine", "pts_coarse", "pts_fine", "viewdirs")):
                    z_out[key][j][rows] = ex[name].detach().float()
            del loss, parts, part
        grads = {"knots": kn.grad.double(), "transform": tr.grad.double()}
        for tag, q in (("nerf", qc), ("nerf_fine", qf)):
            for k, v in q.items():
                grads[tag + "." + k] = v.grad.double()
    return {"loss": total, "grads": grads, "z": z_out}


def error_table(truth, candidates):
    """truth: {name: f64 tensor}; candidates: {label: {name: tensor}} -> {name: {label: (max_err / max|truth|,
    | ||x|| / ||truth|| - 1 |, ||x - truth|| / ||truth||)}}."""
    out = {}
    for name, t in truth.items():
        mx, nt = float(t.abs().max()), float(t.norm())
        row = {}
        for label, g in candidates.items():
            x = g[name].detach().double().cpu().reshape(t.shape)
            row[label] = (float((x - t).abs().max()) / mx, abs(float(x.norm()) / nt - 1.0), float((x - t).norm()) / nt)
        out[name] = row
    return out


def format_table(tab, labels):
    lines = ["%-40s" % "gradient" + "".join(" | %-31s" % ("%s: max/norm/L2" % lb) for lb in labels)]
    for name, row in tab.items():
        lines.append("%-40s" % name + "".join(" | %.2e %.2e %.2e    " % row[lb] for lb in labels))
    return "\n".join(lines)

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
        maps = {k: {} for k in ("evt", "rgb")}       # per-ray outputs of both renders (float32): rgb_map, rgb0, acc_map, acc0, disp_map, disp0
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
            _store_maps(maps["evt"], parts["ret_event"], rows_e, 2 * Re)
            _store_maps(maps["rgb"], parts["ret_rgb"], rows_r, P * Rr)
            del loss, parts, part
        grads = {"knots": kn.grad.double(), "transform": tr.grad.double()}
        for tag, q in (("nerf", qc), ("nerf_fine", qf)):
            for k, v in q.items():
                grads[tag + "." + k] = v.grad.double()
    return {"loss": total, "grads": grads, "z": z_out, "maps": maps}


MAP_KEYS = ("rgb_map", "rgb0", "acc_map", "acc0", "disp_map", "disp0")


def _store_maps(dst, ret, rows, n_total):
    """rows of the per-ray outputs of one render chunk -> float32 arrays over the whole batch"""
    for k in MAP_KEYS:
        v = ret[k].detach().float()
        if k not in dst:
            dst[k] = torch.empty((n_total,) + tuple(v.shape[1:]), dtype=torch.float32)
        dst[k][rows] = v


def step_grads_vjp(cfg, pc, pf, knots, transform, evt_ts, rgb_ts, idx_evt, idx_rgb, target_acc, target_rgb, draws_evt, draws_rgb,
                   dtype=torch.float32, n_chunks=8, event_bins=1, z_forced=None):
    """One training step's loss and gradients in `dtype`, in pixel chunks, for ANY loss - also the L2-normalised event loss
    (train.py:238-292), which is not a sum over pixels and which step_grads therefore refuses to chunk.  Two passes:
      1. without autograd, chunk by chunk: the four rendered colour arrays (rgb_map / rgb0 of the event and of the blur render)
         and the depths of both renders;
      2. the loss on those arrays (tiny graph) gives d loss / d colours; then chunk by chunk WITH autograd the same render (the
         depths of pass 1 forced in, so sample_pdf is not re-drawn) is back-propagated with its slice of those vectors.
    z_forced: the "z" dict another evaluation returned (e.g. the float32 one for a float64 run): its depths replace this
    evaluation's own in BOTH passes, so that the two differentiate the same function (sample_pdf turns 1e-7 differences of the
    coarse weights into different samples).
    By the chain rule the accumulated parameter gradients are the gradients of the whole step (a chunk's colours depend on the
    chunk's rays and the parameters only).  Returns the dict of step_grads."""
    Re, Rr, P, C = idx_evt.shape[0], idx_rgb.shape[0], cfg.n_poses, cfg.channels
    S, F = cfg.n_samples, cfg.n_samples + cfg.n_importance
    Pe = event_bins + 1
    K = cfg.K()
    with default_dtype(dtype):
        leaf = lambda t: t.detach().to(dtype).clone().requires_grad_(True)   # noqa: E731
        qc = {k: leaf(v) for k, v in pc.items()}
        qf = {k: leaf(v) for k, v in pf.items()}
        kn, tr = leaf(knots), leaf(transform)
        z_out = {k: tuple(torch.empty(shape, dtype=torch.float32) for shape in ((n, S), (n, F)))
                 for k, n in (("evt", Pe * Re), ("rgb", P * Rr))}
        col = {k: torch.empty((n, C), dtype=dtype) for k, n in (("e1", Pe * Re), ("e0", Pe * Re), ("r1", P * Rr), ("r0", P * Rr))}
        eb = np.linspace(0, Re, n_chunks + 1).astype(int)
        rb = np.linspace(0, Rr, n_chunks + 1).astype(int)

        def renders(c, forced):
            e0, e1, r0, r1 = int(eb[c]), int(eb[c + 1]), int(rb[c]), int(rb[c + 1])
            rows_e, rows_r = pose_major_rows(Pe, Re, e0, e1), pose_major_rows(P, Rr, r0, r1)
            poses_e = O.trajectory_poses(kn, None, evt_ts.to(dtype), Pe, cfg.traj)
            poses_r = O.trajectory_poses(kn, tr, rgb_ts.to(dtype), P, cfg.traj)
            zf_e = zf_r = None
            if forced or z_forced is not None:
                zsrc = z_out if z_forced is None else z_forced
                zf_e = tuple(t[rows_e].to(dtype) for t in zsrc["evt"][:2])
                zf_r = tuple(t[rows_r].to(dtype) for t in zsrc["rgb"][:2])
            ret_e, ex_e = O.render(qc, qf, poses_e, idx_evt[e0:e1], cfg.H, cfg.W, K, C, S, cfg.n_importance,
                                   _chunk_draws(draws_evt, rows_e, dtype), z_forced=zf_e, want_extras=True)
            ret_r, ex_r = O.render(qc, qf, poses_r, idx_rgb[r0:r1], cfg.H, cfg.W, K, C, S, cfg.n_importance,
                                   _chunk_draws(draws_rgb, rows_r, dtype), z_forced=zf_r, want_extras=True)
            return rows_e, rows_r, ret_e, ret_r, ex_e, ex_r

        maps = {k: {} for k in ("evt", "rgb")}
        with torch.no_grad():
            for c in range(n_chunks):
                rows_e, rows_r, ret_e, ret_r, ex_e, ex_r = renders(c, False)
                col["e1"][rows_e], col["e0"][rows_e] = ret_e["rgb_map"], ret_e["rgb0"]
                col["r1"][rows_r], col["r0"][rows_r] = ret_r["rgb_map"], ret_r["rgb0"]
                _store_maps(maps["evt"], ret_e, rows_e, Pe * Re)
                _store_maps(maps["rgb"], ret_r, rows_r, P * Rr)
                for key, rows, ex in (("evt", rows_e, ex_e), ("rgb", rows_r, ex_r)):
                    z_out[key][0][rows] = ex["z_coarse"].float()
                    z_out[key][1][rows] = ex["z_fine"].float()
        lv = {k: v.clone().requires_grad_(True) for k, v in col.items()}
        if event_bins == 1:
            le, _, _ = O.event_loss(lv["e1"], lv["e0"], Re, target_acc.to(torch.float64), C, cfg.dataset, cfg.threshold, cfg.coeff_syn,
                                    cfg.coeff_real)
        else:
            le, _, _ = O.event_loss_binned(lv["e1"], lv["e0"], Re, [t.to(torch.float64) for t in target_acc], C, cfg.dataset,
                                           cfg.threshold, cfg.coeff_syn, cfg.coeff_real)
        lr_, _, _ = O.blur_loss(lv["r1"], lv["r0"], target_rgb.to(dtype), P, cfg.rgb_coeff)
        loss = le + lr_
        loss.backward()
        for c in range(n_chunks):
            rows_e, rows_r, ret_e, ret_r, _, _ = renders(c, True)
            torch.autograd.backward([ret_e["rgb_map"], ret_e["rgb0"], ret_r["rgb_map"], ret_r["rgb0"]],
                                    [lv["e1"].grad[rows_e], lv["e0"].grad[rows_e], lv["r1"].grad[rows_r], lv["r0"].grad[rows_r]])
        grads = {"knots": kn.grad.double(), "transform": tr.grad.double()}
        for tag, q in (("nerf", qc), ("nerf_fine", qf)):
            for k, v in q.items():
                grads[tag + "." + k] = v.grad.double()
    return {"loss": float(loss.detach()), "grads": grads, "z": z_out, "maps": maps}


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

"""GPU tests against a FLOAT64 evaluation - the arithmetic-neutral yardstick for the MFMA arithmetic modes.  The default
('split') mode runs every GEMM - forward, dX chain, dW - as 3 f16 MFMAs on hi/lo operands (22 bits in flight, 19-bit saved
operands); these tests hold it to the SAME bounds as the exact-f32 mode.

1. test_step_gradients_vs_float64: one training step (train.py:160-340).  For every gradient (knots, transform, every weight
   and bias of both networks) the error of the HIP path against the float64 evaluation of the same step is compared with
   the error of the float32 oracle (= the reference's arithmetic, torch CPU) against the same float64 evaluation:
       err(HIP vs f64) <= 5e-4 + 1.5 * err(float32 oracle vs f64)
   (1.5 instead of 1 for the scatter of a single draw; the floor is half of SURVEY 8c's contract - see FLOOR for why not less).  The bound says
   "no further from the truth than the reference's own fp32 arithmetic", and it does not depend on which ReLU masks flip between
   two float32 implementations: two float32
   evaluations of this path differ from each other by about the contract's 1e-3 already (an ulp in pts = o + d z is amplified
   2^9 times by the positional encoding) - the exact-f32 mode, held to the same bound, is the control.
   tools/experiments/f64_truth.py prints the full decomposition (profiles/r03_f64_truth_*.log).
   Sizes: the three G8 specs with a mean-squared event loss (~4 k points), 1/8 of C2, and the full C2 step of BASELINE.json
   (4081 rays, 783 552 points).
2. test_full_size_step_vs_oracle: the full C2 step, HIP against the float32 oracle directly (loss, pose gradients, sampled
   weight-gradient entries, gradient norms) - the comparison SURVEY 8c words its tolerances for, at benchmark size.
3. test_mlp_backward_arithmetic_vs_float64: one network on IDENTICAL points (no trajectory / ray / sampling in front), forward
   and backward, against torch float64 evaluating the SAME piecewise-linear branch (the HIP forward's own ReLU masks), at 1 024
   and 130 560 points: what the MFMA arithmetic itself contributes, with the ReLU-flip lottery taken out.
"""
import numpy as np
import pytest
import torch

import benerf_oracle as O
import f64_truth as T
import golden_inputs as GI
from conftest import REPORT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

G8_SPECS = [
    ("unreal_C1", "unreal", 1, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3),
    ("unreal_C3", "unreal", 3, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3),
    ("e2syn_C3", "e2nerf_syn", 3, "E2NeRF_Synthetic", 0.2, 7, 32, 32, 16, 5),
    ("e2real_C3", "e2nerf_real", 3, "E2NeRF_Real", -1.0, 31, 16, 32, 16, 2),
]
# err(HIP vs f64) <= FLOOR + FACTOR * err(float32 oracle vs f64).  Round 3's floor was the contract itself (1e-3 of the largest
# entry); it is HALF the contract now, and it cannot go lower for ANY float32 implementation: measured in round 4 with the
# exact-f32 mode (bit-exact f32 products = the reference's own arithmetic on another summation order), a floor of 1e-4 fails
# at full C2 size - nerf_fine.pts_linears.3.weight sits 1.37e-3 from float64 where the float32 oracle sits 7.1e-4 (one more
# ReLU unit of a heavy sample flipped in one float32 evaluation than in the other); at G8 size an entry whose oracle error
# happens to be 3e-7 sits 4e-5 away.  Both arithmetic modes are held to the same bound; what the split ARITHMETIC adds on
# identical masks is held to float32's own error (1e-5) by test_mlp_backward_arithmetic_vs_float64.
FACTOR = {"L2": 1.5, "max": 1.5, "norm": 1.5}
FLOOR = {"L2": 5e-4, "max": 5e-4, "norm": 1e-4}


def _case(name):
    from benerf_amd import workloads as WL
    if name.startswith("g8_"):
        si = int(name[3:])
        tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = G8_SPECS[si]
        rng = np.random.default_rng(1808 + si)
        cam = GI.CAMERAS[cname]
        window, chunks = (0.1 if "unreal" in tag else 0.25), 1
    else:
        wl = WL.WORKLOADS[name[:2]]
        frac = 8 if name.endswith("eighth") else 1
        cname, C, dataset, thr, P, S, Ni = wl["cam"], wl["channels"], wl["dataset"], wl["threshold"], wl["n"], wl["S"], wl["Ni"]
        Re, Rr = wl["Re"] // frac, max(wl["Rr"] // frac, 1)
        rng = np.random.default_rng(2024 + int(name[1]) - 2)
        cam = WL.CAMERAS[cname]
        window, chunks = wl["window"], (2 if frac == 8 else {"C2": 12, "C3": 12, "C4": 24, "C5": 32}[name[:2]])
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    x = dict(cam=cam, C=C, dataset=dataset, thr=thr, P=P, S=S, Ni=Ni, Re=Re, Rr=Rr, chunks=chunks, pc=pc, pf=pf,
             knots=GI.knots_init(rng) * 3, tr=GI.transform_small(rng) * 0.1, idx_e=GI.pixel_indices(rng, cam, Re),
             idx_r=GI.pixel_indices(rng, cam, Rr))
    HW = cam["H"] * cam["W"]
    x["accu"] = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32))
    x["img"] = torch.from_numpy(rng.random((HW, C)).astype(np.float32))
    low = float(rng.random() * (1 - window))
    x["evt_ts"] = torch.tensor([low, low + window], dtype=torch.float32)
    x["d_e"], x["d_r"] = GI.render_draws(rng, 2 * Re, S, Ni), GI.render_draws(rng, P * Rr, S, Ni)
    return x


def _oracle_args(x):
    cam = x["cam"]
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=x["C"],
                       n_samples=x["S"], n_importance=x["Ni"], n_poses=x["P"], dataset=x["dataset"], threshold=x["thr"])
    tacc = x["accu"].double().reshape(-1, 1)[x["idx_e"]]
    return (cfg, x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"], tacc,
            x["img"][x["idx_r"]], x["d_e"], x["d_r"])


def _hip_step(x, mode, z_forced):
    from benerf_amd import engine, kernels as K, workloads as WL
    from benerf_amd.model import optimize
    prev = K.get_mlp_precision()
    K.set_mlp_precision(mode)
    try:
        cam = x["cam"]
        wl = dict(cam="_t", channels=x["C"], dataset=x["dataset"], threshold=x["thr"], window=float(x["evt_ts"][1] - x["evt_ts"][0]),
                  n=x["P"], S=x["S"], Ni=x["Ni"], Re=x["Re"], Rr=x["Rr"])
        WL.CAMERAS["_t"] = cam
        args = WL.make_args(wl, optimize_trans=True)
        model = optimize.Model(args)
        model.graph.to(DEV)
        g = model.build_network(args)
        with torch.no_grad():
            for net, p in ((g.nerf, x["pc"]), (g.nerf_fine, x["pf"])):
                for name in K.LAYER_NAMES:
                    lin = engine.getattr_path(net, name)
                    lin.weight.copy_(p[name + ".weight"])
                    lin.bias.copy_(p[name + ".bias"])
            g.evt_knot_pose_se3.params.weight.copy_(x["knots"])
            g.transform.params.weight.copy_(x["tr"])
        cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
        step.keep_maps = True

        def dd(d):
            return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
        zf = torch.cat([z_forced["evt"][1], z_forced["rgb"][1]]).to(DEV)
        losses = step.step(x["evt_ts"].to(DEV), torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV),
                           x["accu"].to(DEV), x["img"].to(DEV), dd(x["d_e"]), dd(x["d_r"]), z_fine_forced=zf)
        step.check_range()
        grads = {"knots": step.g_knots.cpu().clone(), "transform": step.g_transform.cpu().clone()}
        for nn_, fn in (("nerf", step.net_c), ("nerf_fine", step.net_f)):
            for i, name in enumerate(K.LAYER_NAMES):
                grads["%s.%s.weight" % (nn_, name)] = fn.gviews_w[i].cpu().clone()
                grads["%s.%s.bias" % (nn_, name)] = fn.gviews_b[i].cpu().clone()
        ne = step.last_maps["n_event_rays"]
        LAST_MAPS[0] = {"evt": {k: v[:ne].cpu() for k, v in step.last_maps.items() if k != "n_event_rays"},
                        "rgb": {k: v[ne:].cpu() for k, v in step.last_maps.items() if k != "n_event_rays"}}
        return float(losses[0]), grads
    finally:
        K.set_mlp_precision(prev)


def _assert_split_vs_control(tab, case):
    """Round 6 (advisor): the lottery factor widens BOTH modes against the oracle; what it must not hide is a regression of the split
    arithmetic itself.  The exact-f32 mode is the control: the same kernels' tiling, the same summation orders, exact products - on
    every gradient the split mode's whole-tensor error against float64 may exceed the control's by the scatter of one draw at most
    (1.5 x + a fifth of the L2 floor)."""
    worst, bad = (0.0, ""), []
    for name, row in tab.items():
        e_s, e_c = row["split"][2], row["f32"][2]
        bound = (2.0 if name in ("knots", "transform") else 1.5) * e_c + FLOOR["L2"] / 5      # 24 + 6 pose numbers: no averaging, one more draw
        if e_s / bound > worst[0]:
            worst = (e_s / bound, "%s: split %.2e, exact-f32 control %.2e" % (name, e_s, e_c))
        if e_s > bound:
            bad.append("%s: split L2 error %.3e > 1.5 x control %.3e + %.0e" % (name, e_s, e_c, FLOOR["L2"] / 5))
    REPORT.append("f64 truth %-10s split vs exact-f32 control, L2: closest to its bound %.2f of it (%s)" % (case, worst[0], worst[1]))
    assert not bad, "%s - split further from float64 than the exact-f32 control allows:\n%s" % (case, "\n".join(bad))


LAST_MAPS = [None]      # per-ray outputs (rgb_map, rgb0, acc, disp of both renders) of the last _hip_step


def _assert_no_worse(tab, label, case, lottery_factor=None):
    """tab: f64_truth.error_table rows {name: {label: (max, norm, L2)}} with the float32 oracle under 'o32'.
    lottery_factor: replaces FACTOR for the statistics a single flipped ReLU unit dominates - the largest entry error ("max") of
    every gradient and, for the 24 + 6 pose numbers, all of them (their whole-tensor L2 is no average) - on batches too small to
    average the flips out (round 5: the exact-f32 mode itself, bit-exact f32 products on another summation order, sits at 2.1 x
    the float32 oracle's error on the pose gradients of a binned C5 step and at 2.6 x on one weight entry of an 0.17 M-point
    batch), and the norm error of every gradient: |norm(g) - norm(g64)| / norm(g64) is ONE signed number per tensor (the projection
    of the error vector onto the gradient), so "HIP's draw <= 1.5 x the oracle's draw" fails by chance for two implementations of
    identical quality whenever the oracle's own draw happens to be small (round 5, 4 bins at C5 shape: the exact-f32 control sits at
    1.83 x the float32 oracle's norm error on nerf.pts_linears.0.bias - 4.32e-4 against 2.36e-4 - the split mode at 2.16 x).  The
    whole-tensor L2 error of the weight gradients, which bounds the norm error from above, keeps FACTOR everywhere."""
    bad, worst = [], {"L2": (0.0, ""), "max": (0.0, ""), "norm": (0.0, "")}
    for name, row in tab.items():
        n_entries = 24 if name == "knots" else 6 if name == "transform" else 1 << 20
        for j, stat in ((2, "L2"), (0, "max"), (1, "norm")):
            if stat == "norm" and (n_entries < 256 or name.endswith(".bias") and "rgb_linear" in name):
                continue      # the norm of a handful of numbers is one more draw of the L2 error, which is bounded above
            e, ref = row[label][j], row["o32"][j]
            factor = lottery_factor if (lottery_factor is not None and (stat in ("max", "norm") or n_entries < 256)) else FACTOR[stat]
            bound = factor * ref + FLOOR[stat]
            ratio = e / bound
            if ratio > worst[stat][0]:
                worst[stat] = (ratio, "%s %.2e vs oracle %.2e" % (name, e, ref))
            if e > bound:
                bad.append("%s %s: %.3e > %.1f x %.3e + %.0e" % (name, stat, e, factor, ref, FLOOR[stat]))
    for stat, (ratio, what) in worst.items():
        REPORT.append("f64 truth %-10s %-5s %-4s closest to its bound: %.2f of it (%s)" % (case, label, stat, ratio, what))
    assert not bad, "%s, mode %s - gradients further from float64 than the float32 oracle allows:\n%s" % (case, label, "\n".join(bad))


# (the e2real spec of G8 - L2-normalised loss on 16 + 2 pixels - is not here: on ~4 k points ONE flipped ReLU moves a bias gradient
# of the fine network by 1e-2 in either HIP mode while the float32 oracle happens to flip none; test_mlp_backward_arithmetic_
# vs_float64 takes the flips out instead)
@pytest.mark.parametrize("case", ["g8_0", "g8_1", "g8_2", "C2_eighth", "C2"])
def test_step_gradients_vs_float64(case):
    x = _case(case)
    a = _oracle_args(x)
    o32 = T.step_grads(*a, dtype=torch.float32, z_forced=None, n_chunks=x["chunks"])
    o64 = T.step_grads(*a, dtype=torch.float64, z_forced=o32["z"], n_chunks=x["chunks"], force_inputs=False)
    assert abs(o32["loss"] - o64["loss"]) <= 2e-6 * max(1.0, abs(o64["loss"]))
    cands = {"o32": o32["grads"]}
    for mode in ("f32", "split"):
        loss, cands[mode] = _hip_step(x, mode, o32["z"])
        assert abs(loss - o64["loss"]) <= 2e-5 * max(1.0, abs(o64["loss"])), (mode, loss, o64["loss"])
    tab = T.error_table(o64["grads"], cands)
    for mode in ("f32", "split"):
        # G8-sized batches (~4 k points): one flipped unit of a heavy sample IS the largest entry error - the lottery factor there
        _assert_no_worse(tab, mode, case, lottery_factor=3.0 if case.startswith("g8_") else None)
    _assert_split_vs_control(tab, case)


# (the e2real spec of G8 - L2-normalised loss on 16 + 2 pixels - is not here: on ~4 k points ONE flipped ReLU moves a bias gradient
# of the fine network by 1e-2 in either HIP mode while the float32 oracle happens to flip none; test_mlp_backward_arithmetic_
# vs_float64 takes the flips out instead)
@pytest.mark.parametrize("case", ["g8_0", "g8_1", "g8_2", "C2_eighth", "C2"])
def test_step_gradients_vs_float64(case):
    x = _case(case)
    a = _oracle_args(x)
    o32 = T.step_grads(*a, dtype=torch.float32, z_forced=None, n_chunks=x["chunks"])
    o64 = T.step_grads(*a, dtype=torch.float64, z_forced=o32["z"], n_chunks=x["chunks"], force_inputs=False)
    assert abs(o32["loss"] - o64["loss"]) <= 2e-6 * max(1.0, abs(o64["loss"]))
    cands = {"o32": o32["grads"]}
    for mode in ("f32", "split"):
        loss, cands[mode] = _hip_step(x, mode, o32["z"])
        assert abs(loss - o64["loss"]) <= 2e-5 * max(1.0, abs(o64["loss"])), (mode, loss, o64["loss"])
    tab = T.error_table(o64["grads"], cands)
    for mode in ("f32", "split"):
        # G8-sized batches (~4 k points): one flipped unit of a heavy sample IS the largest entry error - the lottery factor there
        _assert_no_worse(tab, mode, case, lottery_factor=3.0 if case.startswith("g8_") else None)
    # Round 6 (advisor): the lottery factor widens BOTH modes against the oracle; what it must not hide is a regression of the split
    # arithmetic itself.  The exact-f32 mode is the control: same kernels' tiling, same summation orders, exact products - on every
    # gradient the split mode's whole-tensor error against float64 may exceed the control's by the scatter of one draw at most.
    worst, bad = (0.0, ""), []
    for name, row in tab.items():
        e_s, e_c = row["split"][2], row["f32"][2]
        bound = 1.5 * e_c + FLOOR["L2"] / 5
        if e_s / bound > worst[0]:
            worst = (e_s / bound, "%s: split %.2e, exact-f32 control %.2e" % (name, e_s, e_c))
        if e_s > bound:
            bad.append("%s: split L2 error %.3e > 1.5 x control %.3e + %.0e" % (name, e_s, e_c, FLOOR["L2"] / 5))
    REPORT.append("f64 truth %-10s split vs exact-f32 control, L2: closest to its bound %.2f of it (%s)" % (case, worst[0], worst[1]))
    assert not bad, "%s - split further from float64 than the exact-f32 control allows:\n%s" % (case, "\n".join(bad))


def test_full_size_step_vs_oracle():
    """The C2 step of BASELINE.json on explicit draws, HIP (both modes) against the float32 oracle itself: see
    _full_size_vs_oracle for what is compared and at which tolerances."""
    x = _case("C2")
    a = _oracle_args(x)
    o32 = T.step_grads(*a, dtype=torch.float32, z_forced=None, n_chunks=x["chunks"])
    _full_size_vs_oracle("C2", x, o32)


# north_star: "outputs match the reference render on identical inputs within 1e-4 RGB abs tol".  Held PER RAY at full size, on the
# forced fine depths of the gradient comparison (everything behind sample_pdf): every colour of both compositing passes, the
# accumulated opacity (a number in [0, 1]: same absolute tolerance) and the disparity (unbounded: 1e-4 absolute + 1e-4 relative).
MAP_ATOL = 1e-4


def _maps_vs_oracle(case, mode, hip, ref, bad):
    for part in ("evt", "rgb"):
        for k in T.MAP_KEYS:
            g_, r_ = hip[part][k].double().reshape(-1), ref[part][k].double().reshape(-1)
            assert g_.shape == r_.shape, (part, k, g_.shape, r_.shape)
            d = (g_ - r_).abs()
            tol = MAP_ATOL + (1e-4 * r_.abs() if k.startswith("disp") else 0.0)
            n_bad = int((d > tol).sum())
            REPORT.append("full-size %s per ray, %-5s %-3s %-8s max|d| %.2e over %d values  beyond tolerance: %d" % (case, mode, part, k, float(d.max()), d.numel(), n_bad))
            if n_bad:
                bad.append("%s %s %s: %d of %d values beyond %.0e (max %.2e)" % (mode, part, k, n_bad, d.numel(), MAP_ATOL, float(d.max())))


# SURVEY 8c's contract at FULL size, in 8c's own terms: loss 2e-5; gradients "1e-3 of the largest entry, 1e-4 on norms" TIMES
# TWO, for both arithmetic modes alike - because that is what the reference's own arithmetic can hold against itself at these
# sizes.  Measured against the float32 oracle (round 4, profiles/r04_gpu_parity_report_full_suite.txt; worst pose gradient /
# worst of 64 sampled entries per gradient / worst norm, each relative as in 8c):
#                exact-f32 mode (bit-exact f32 products, the oracle's arithmetic on another summation order)      split mode
#     C2         0.80e-3 / 1.11e-3 / 0.97e-4                                                                     0.94e-3 / 0.54e-3 / 0.40e-4
#     C3         0.47e-3 / 0.25e-3 / 0.44e-4                                                                     1.04e-3 / 0.31e-3 / 0.58e-4
#     C4         1.29e-3 / 0.46e-3 / 0.58e-4                                                                     1.07e-3 / 0.50e-3 / 0.34e-4
#     C5         1.41e-3 / 0.72e-3 / 1.02e-4                                                                     1.57e-3 / 0.62e-3 / 0.53e-4
# (ReLU-kink flips behind the 2^9 x positional encoding; C5's L2-normalised loss makes every gradient the remainder of cancelling
# sums).  Neither mode is systematically closer.  What the ARITHMETIC contributes is held to float32's own error by
# test_mlp_backward_arithmetic_vs_float64, the whole step to the float64 yardstick by test_step_gradients_vs_float64.
# Round 5: next to those max-type statistics (which a single flipped ReLU unit of a heavy sample dominates) every gradient is
# held to SURVEY 8c's UNDOUBLED 1e-3 on a statistic the lottery does not dominate: the whole-tensor relative L2 error
# ||HIP - oracle||_2 / ||oracle||_2 ("l2").  Measured (profiles/r05_gpu_parity_report.txt): 46 of the 50 gradients of C2-C5
# sit at 0.1-0.9e-3 in BOTH modes; the four that do not are the same in both modes - the pose gradients (24 + 6 numbers: their
# "whole tensor" is no average at all; C4 exact-f32 1.15e-3, C5 1.58e-3 exact-f32 / 1.68e-3 split) and the first layer's weight
# gradient, directly behind the encoding's 2^9 x derivative (C5: 1.25e-3 exact-f32, 1.22e-3 split).  Those keep the doubled figure
# ("l2_pose_and_first_layer"), everything else is asserted at 1e-3.
FULL_SIZE_TOL = {"pose": 2e-3, "entries": 2e-3, "norm": 2e-4, "l2": 1e-3, "l2_pose_and_first_layer": 2e-3}


def _full_size_vs_oracle(case, x, o32):
    rng = np.random.default_rng(7)
    picks = {name: torch.from_numpy(rng.integers(0, ref.numel(), 64)) for name, ref in o32["grads"].items()}
    bad = []
    for mode in ("f32", "split"):
        loss, g = _hip_step(x, mode, {k: (None, v[1]) for k, v in o32["z"].items()})
        if abs(loss - o32["loss"]) > 2e-5 * max(1.0, abs(o32["loss"])):
            bad.append("%s loss %r vs %r" % (mode, loss, o32["loss"]))
        _maps_vs_oracle(case, mode, LAST_MAPS[0], o32["maps"], bad)
        worst = {"pose": 0.0, "entries": 0.0, "norm": 0.0, "l2": 0.0}
        for name, ref in o32["grads"].items():
            got = g[name].double().reshape(ref.shape)
            ref = ref.double()
            mx = float(ref.abs().max())
            l2 = float((got - ref).norm() / ref.norm())          # whole tensor, relative: SURVEY 8c's 1e-3, not doubled
            worst["l2"] = max(worst["l2"], l2)
            first = name in ("knots", "transform") or name.endswith("pts_linears.0.weight")
            if l2 > FULL_SIZE_TOL["l2_pose_and_first_layer" if first else "l2"]:
                bad.append("%s %s: relative L2 error %.2e" % (mode, name, l2))
            if name in ("knots", "transform"):
                e = float((got - ref).abs().max()) / mx
                worst["pose"] = max(worst["pose"], e)
                REPORT.append("full-size %s vs oracle, %-5s d%-36s max err %.2e of the largest entry  rel-L2 %.2e" % (case, mode, name, e, l2))
                if e > FULL_SIZE_TOL["pose"]:
                    bad.append("%s %s: %.2e" % (mode, name, e))
                continue
            idx = picks[name]
            e = float((got.reshape(-1)[idx] - ref.reshape(-1)[idx]).abs().max()) / mx
            en = abs(float(got.norm() / ref.norm()) - 1.0)
            worst["entries"], worst["norm"] = max(worst["entries"], e), max(worst["norm"], en)
            REPORT.append("full-size %s vs oracle, %-5s d%-36s sampled entries %.2e  norm %.2e  rel-L2 %.2e" % (case, mode, name, e, en, l2))
            if e > FULL_SIZE_TOL["entries"] or en > FULL_SIZE_TOL["norm"]:
                bad.append("%s %s: entries %.2e norm %.2e" % (mode, name, e, en))
        REPORT.append("full-size %s vs oracle, %-5s WORST pose %.2e  sampled entries %.2e  norms %.2e  rel-L2 %.2e" %
                      (case, mode, worst["pose"], worst["entries"], worst["norm"], worst["l2"]))
    assert not bad, "%s:\n%s" % (case, "\n".join(bad))


# C4 (1.57 M points: ~2.5 minutes of oracle time on the host) is part of the default suite since round 5
@pytest.mark.parametrize("case", ["C3", "C4", "C5"])
def test_full_size_step_vs_oracle_colour_configs(case):
    """The full-size steps of the other BASELINE.json GPU configurations, HIP (both modes) against the float32 oracle on explicit
    draws with the oracle's fine depths forced in: C3 (colour kernels, 0.78 M points), C4 (800 x 800 camera, lin-log brightness,
    8181 rays, 1.57 M points) and C5 (31 poses, 64 + 192 samples, 2.1 M points, the L2-NORMALISED event loss, train.py:238-292 -
    the configuration where round 3's f16 backward needed a widened tolerance already at G8 size).  The oracle evaluates the
    step in pixel chunks by a two-pass vector-Jacobian product (f64_truth.step_grads_vjp: any loss, bounded memory).
    Tolerances: FULL_SIZE_TOL, the same for both modes."""
    x = _case(case)
    cfg, pc, pf, kn, tr, ets, rts, idx_e, idx_r, tacc, timg, d_e, d_r = _oracle_args(x)
    o32 = T.step_grads_vjp(cfg, pc, pf, kn, tr, ets, rts, idx_e, idx_r, tacc, timg, d_e, d_r, dtype=torch.float32, n_chunks=x["chunks"])
    _full_size_vs_oracle(case, x, o32)


@pytest.mark.parametrize("n_rays", [8, 1020])
def test_mlp_backward_arithmetic_vs_float64(n_rays):
    """What the MFMA arithmetic itself contributes: one network, forward + backward on IDENTICAL points with a fixed upstream
    gradient, against torch float64 evaluating exactly the piecewise-linear branch the HIP forward took (its saved ReLU masks
    forced into the float64 network).  Without that, a pre-activation within round-off of zero flips between ANY two
    evaluations - torch float32 against float64 on these very points moves dW_0 by 1e-2 - and the comparison measures a
    lottery, not arithmetic.  Bounds: the exact-f32 mode has to be float32-clean (1e-5; d_pts, behind the 2^9 x derivative of the
    encoding, 5e-5); the split mode's f16 backward operands
    have to stay inside the contract, 1e-3 of the largest entry and 1e-4 on the norms of the weight gradients, at a G4-sized
    batch (1 024 points, no averaging to speak of) and at 130 560 points."""
    from benerf_amd import kernels as K
    from test_kernels_gpu import _act_views
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
    pts32 = ro[:, None, :] + rd[:, None, :] * z[:, :, None]        # float32 values every evaluation consumes
    M = N * S

    # upstream gradient: what compositing hands back for a mean-squared colour loss (structured, not noise); float32 torch
    q32 = {k: v.clone() for k, v in p.items()}
    raw32 = O.mlp_forward(q32, pts32, vd).detach().requires_grad_(True)
    rgb = O.composite(raw32, z, rd, noise, C)[0]
    ((rgb - target) ** 2).mean().backward()
    d_raw = raw32.grad.reshape(-1, C + 1).contiguous()

    def truth(masks):
        with T.default_dtype(torch.float64):
            q = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
            pts = pts32.double().requires_grad_(True)
            raw = O.mlp_forward(q, pts, vd.double(), relu_masks=masks)
            (raw.reshape(-1, C + 1) * d_raw.double()).sum().backward()
            out = {k: v.grad for k, v in q.items()}
            out["d_pts"] = pts.grad.reshape(-1, 3)
            return out, raw.detach()

    dv = lambda t: t.to(DEV).contiguous()   # noqa: E731
    prev = K.get_mlp_precision()
    bad = []
    try:
        for mode, tol_max, tol_norm in (("f32", 1e-5, 1e-5), ("split", 1e-5, 1e-5), ("split_f16bwd", 1e-3, 1e-4)):
            K.set_mlp_precision(mode)
            net = K.PackedMlp([dv(p[n + ".weight"]) for n in K.LAYER_NAMES], [dv(p[n + ".bias"]) for n in K.LAYER_NAMES], C)
            net.pack()
            raw, acts = K.mlp_fwd(net, dv(ro), dv(rd), dv(vd), dv(z), True)
            av = _act_views(acts, M, mode)
            # the branch the HIP forward took: its own sign-bit words where the mode stores them apart from the values ('split':
            # v_cmp on the f32 value - a positive value below f16's subnormal grid has hi = 0 and IS active), else value > 0
            masks = {k: (av["mask" + k[1:]] if ("mask" + k[1:]) in av else av[k] > 0).to(torch.float64)
                     for k in ["h%d" % i for i in range(8)] + ["hv"]}
            g64, raw64 = truth(masks)
            e_raw = float((raw.cpu().double().reshape(raw64.shape) - raw64).abs().max() / raw64.abs().max())
            REPORT.append("MLP arithmetic vs f64 (own masks), %d points, %-5s raw: %.2e of the largest" % (M, mode, e_raw))
            assert e_raw <= 1e-5
            gw = [torch.zeros_like(w) for w in net.weights]
            gb = [torch.zeros_like(b) for b in net.biases]
            d_pts, _ = K.mlp_bwd(net, dv(d_raw), acts, N, S, gw, gb, False)
            g = {"d_pts": d_pts.cpu()}
            for i, n in enumerate(K.LAYER_NAMES):
                g[n + ".weight"], g[n + ".bias"] = gw[i].cpu(), gb[i].cpu()
            tab = T.error_table(g64, {mode: g})
            for name, row in tab.items():
                e_max, e_norm, e_l2 = row[mode]
                REPORT.append("MLP arithmetic vs f64 (own masks), %d points, %-5s d%-24s max %.2e norm %.2e L2 %.2e" % (M, mode, name, e_max, e_norm, e_l2))
                if e_max > (5e-5 if (mode != "split_f16bwd" and name == "d_pts") else tol_max) or (name.endswith("weight") and e_norm > tol_norm):
                    bad.append("%s %s: max %.2e norm %.2e" % (mode, name, e_max, e_norm))
    finally:
        K.set_mlp_precision(prev)
    assert not bad, "\n".join(bad)

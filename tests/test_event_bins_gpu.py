"""GPU tests of dense event bins (BASELINE.json configs[4] "31 virtual poses + dense event bins"; SURVEY 8: an extension, the
reference has ONE bin per step - parity is defined at B = 1).  TrainStep(event_bins=B): the event span is cut into B contiguous
equal bins, the event batch is rendered once at the B + 1 bin boundaries (get_pose_evt(args, ts, seg_num=B + 1),
model/optimize.py:58-82) and every bin contributes the reference's event term (train.py:204-292) on its pose pair.
Oracle: benerf_oracle.step_loss_binned = the sum of B single-window step_loss event parts (each pinned at B = 1 by G8) + one
blur part (tests/test_oracle_golden.py::test_binned_oracle_is_the_sum_of_single_window_steps)."""
import numpy as np
import pytest
import torch

import benerf_oracle as O
import f64_truth as T
import golden_inputs as GI
from conftest import REPORT, report
from test_path_gpu import build_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(rng, wl, B):
    from benerf_amd import workloads as WL
    cam = WL.CAMERAS[wl["cam"]]
    C, P, S, Ni, Re, Rr = wl["channels"], wl["n"], wl["S"], wl["Ni"], wl["Re"], wl["Rr"]
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    HW = cam["H"] * cam["W"]
    x = dict(cam=cam, pc=pc, pf=pf, knots=GI.knots_init(rng) * 3, tr=GI.transform_small(rng) * 0.1,
             idx_e=GI.pixel_indices(rng, cam, Re), idx_r=GI.pixel_indices(rng, cam, Rr),
             accu=torch.from_numpy(rng.integers(-3, 4, (B, HW)).astype(np.float32)),
             img=torch.from_numpy(rng.random((HW, C)).astype(np.float32)),
             d_e=GI.render_draws(rng, (B + 1) * Re, S, Ni), d_r=GI.render_draws(rng, P * Rr, S, Ni))
    low = float(rng.random() * (1 - wl["window"]))
    x["evt_ts"] = torch.tensor([low, low + wl["window"]], dtype=torch.float32)
    x["cfg"] = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C, n_samples=S,
                            n_importance=Ni, n_poses=P, dataset=wl["dataset"], threshold=wl["threshold"])
    return x


LAST_MAPS = [None]


def _hip_step(x, wl, B, mode, z_fine=None, event_bins_kw=True, keep_maps=False):
    from benerf_amd import engine, kernels as K, workloads as WL
    prev = K.get_mlp_precision()
    K.set_mlp_precision(mode)
    try:
        cam = x["cam"]
        WL.CAMERAS["_bins"] = cam
        args = WL.make_args(dict(wl, cam="_bins"), optimize_trans=True)
        _, g = build_graph(args, x["pc"], x["pf"], x["knots"], x["tr"])
        cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        kw = {"event_bins": B} if event_bins_kw else {}
        step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV), **kw)
        step.keep_maps = keep_maps

        def dd(d):
            return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
        accu = x["accu"].to(DEV) if event_bins_kw else x["accu"][0].to(DEV)
        losses = step.step(x["evt_ts"].to(DEV), torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV), accu,
                           x["img"].to(DEV), dd(x["d_e"]), dd(x["d_r"]), z_fine_forced=None if z_fine is None else z_fine.to(DEV))
        step.check_range()
        if keep_maps:
            ne = step.last_maps["n_event_rays"]
            LAST_MAPS[0] = {"evt": {k: v[:ne].cpu() for k, v in step.last_maps.items() if k != "n_event_rays"},
                            "rgb": {k: v[ne:].cpu() for k, v in step.last_maps.items() if k != "n_event_rays"}}
        grads = {"knots": step.g_knots.cpu().clone(), "transform": step.g_transform.cpu().clone()}
        for nn_, fn in (("nerf", step.net_c), ("nerf_fine", step.net_f)):
            for i, name in enumerate(K.LAYER_NAMES):
                grads["%s.%s.weight" % (nn_, name)] = fn.gviews_w[i].cpu().clone()
                grads["%s.%s.bias" % (nn_, name)] = fn.gviews_b[i].cpu().clone()
        return losses.cpu().clone(), grads, step.flat_p.cpu().clone()
    finally:
        K.set_mlp_precision(prev)


def test_one_bin_is_todays_step():
    """event_bins = 1 with a [1, H W] polarity image is bit-identical to the step without the argument (loss vector, every
    gradient, parameters after Adam)."""
    from benerf_amd import workloads as WL
    wl = dict(WL.WORKLOADS["C5"], S=16, Ni=32, Re=16, Rr=2, n=7)
    x = _inputs(np.random.default_rng(41), wl, 1)
    la, ga, pa = _hip_step(x, wl, 1, "split", event_bins_kw=True)
    lb, gb, pb = _hip_step(x, wl, 1, "split", event_bins_kw=False)
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def _binned_truths(x, wl, B, chunks):
    """float32 oracle and float64 truth of the binned step (f64_truth.step_grads_vjp: any loss, chunked); the float64 run is
    handed the float32 run's depths, so both differentiate the same function."""
    tacc = [x["accu"][b].double().reshape(-1, 1)[x["idx_e"]] for b in range(B)]
    a = (x["cfg"], x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"], tacc,
         x["img"][x["idx_r"]], x["d_e"], x["d_r"])
    o32 = T.step_grads_vjp(*a, dtype=torch.float32, n_chunks=chunks, event_bins=B)
    o64 = T.step_grads_vjp(*a, dtype=torch.float64, n_chunks=chunks, event_bins=B, z_forced=o32["z"])
    assert abs(o32["loss"] - o64["loss"]) <= 2e-6 * max(1.0, abs(o64["loss"]))
    return o32, o64


@pytest.mark.parametrize("case", ["C2_eighth", "C5_sixteenth"])
def test_four_bins_gradients_vs_float64(case):
    """B = 4, both arithmetic modes, every gradient of the step against a FLOAT64 evaluation of the binned oracle, by the
    criterion of tests/test_f64_truth_gpu.py: err(HIP vs f64) <= floor + 1.5 x err(float32 oracle vs f64) - "no further from the
    truth than the reference's own fp32 arithmetic".  Why not HIP vs the float32 oracle directly: with contiguous bins an
    interior pose is the END of one bin and the START of the next, its two gradient contributions largely cancel, and what is
    left of the pose gradients carries 3-4e-3 of relative float32 noise in ANY float32 evaluation (measured: exact-f32 mode
    3.7e-3 from the float32 oracle at C5 shape, the weights 6e-4).  Cases: an eighth of C2 (gray, safelog, mean-squared event
    loss: 128 event pixels x 5 poses + 19 x 13 blur rays = 887 rays, 0.17 M points) and a sixteenth of the C5 batch (colour,
    lin-log, 31 blur poses, 64 + 192 samples, the L2-NORMALISED loss of every bin: 128 event pixels x 5 poses + 31 x 8 blur
    rays = 888 rays, 0.23 M points).  The statistics one flipped ReLU unit dominates - the largest entry error, the norm error (one
    signed number per tensor) and everything about the 24 + 6 pose numbers - get the lottery factor of tests/test_f64_truth_gpu.py
    (3 instead of 1.5: measured with the EXACT-f32 mode, profiles/r05_gpu_parity_report.txt - it sits at 1.83 x the float32 oracle's
    norm error on the coarse network's layer-0 bias at C5 shape); the whole-tensor L2 error of every weight gradient, which bounds
    its norm error from above, is held to 1.5."""
    from benerf_amd import workloads as WL
    from test_f64_truth_gpu import _assert_no_worse, _assert_split_vs_control
    B = 4
    if case == "C2_eighth":
        wl, seed, chunks = dict(WL.WORKLOADS["C2"], Re=128, Rr=13), 1900, 3
    else:
        wl, seed, chunks = dict(WL.WORKLOADS["C5"], Re=128, Rr=8), 2031, 4
    x = _inputs(np.random.default_rng(seed), wl, B)
    o32, o64 = _binned_truths(x, wl, B, chunks)
    z_fine = torch.cat([o32["z"]["evt"][1], o32["z"]["rgb"][1]])
    cands = {"o32": o32["grads"]}
    for mode in ("f32", "split"):
        losses, cands[mode], _ = _hip_step(x, wl, B, mode, z_fine)
        assert abs(float(losses[0]) - o64["loss"]) <= 2e-5 * max(1.0, abs(o64["loss"])), (mode, float(losses[0]), o64["loss"])
    tab = T.error_table(o64["grads"], cands)
    for mode in ("f32", "split"):
        _assert_no_worse(tab, mode, "bins4 " + case, lottery_factor=3.0)
    _assert_split_vs_control(tab, "bins4 " + case)


def test_four_bins_full_size_c5_vs_oracle():
    """The FULL C5 batch (2048 event pixels x 5 poses + 132 blur pixels x 31 poses = 14 332 rays, 64 + 192 samples, 3.7 M sample
    points, the L2-normalised event loss of every bin) at B = 4, both arithmetic modes, against the float32 oracle with the
    oracle's fine depths forced in - the full-size comparison of tests/test_f64_truth_gpu.py (per-ray colours at north_star's
    1e-4, loss 2e-5, whole-tensor relative L2 of every gradient) for the configuration BASELINE.json configs[4] names.  The pose
    gradients of a binned step carry several 1e-3 of float32 noise in ANY float32 evaluation (an interior pose ends one bin and
    starts the next: its two contributions largely cancel - test_four_bins_gradients_vs_float64; measured here, round 6: knots /
    transform 4.7e-3 / 6.3e-3 in the EXACT-f32 mode, 4.4e-3 / 5.7e-3 in the split mode): they get 1e-2, everything else 1e-3 / 2e-3
    as in FULL_SIZE_TOL (measured: every weight gradient <= 7.4e-4 in both modes)."""
    from benerf_amd import workloads as WL
    import test_f64_truth_gpu as F
    B = 4
    wl = dict(WL.WORKLOADS["C5"])
    x = _inputs(np.random.default_rng(2032), wl, B)
    tacc = [x["accu"][b].double().reshape(-1, 1)[x["idx_e"]] for b in range(B)]
    o32 = T.step_grads_vjp(x["cfg"], x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"], tacc,
                           x["img"][x["idx_r"]], x["d_e"], x["d_r"], dtype=torch.float32, n_chunks=48, event_bins=B)
    z_fine = torch.cat([o32["z"]["evt"][1], o32["z"]["rgb"][1]])
    bad = []
    for mode in ("f32", "split"):
        losses, g, _ = _hip_step(x, wl, B, mode, z_fine, keep_maps=True)
        if abs(float(losses[0]) - o32["loss"]) > 2e-5 * max(1.0, abs(o32["loss"])):
            bad.append("%s loss %r vs %r" % (mode, float(losses[0]), o32["loss"]))
        F._maps_vs_oracle("C5 x 4 bins", mode, LAST_MAPS[0], o32["maps"], bad)
        worst = 0.0
        for name, ref in o32["grads"].items():
            got, ref = g[name].double().reshape(ref.shape), ref.double()
            l2 = float((got - ref).norm() / ref.norm())
            pose = name in ("knots", "transform")
            tol = 1e-2 if pose else F.FULL_SIZE_TOL["l2_pose_and_first_layer" if name.endswith("pts_linears.0.weight") else "l2"]
            if not pose:
                worst = max(worst, l2)
            REPORT.append("full-size C5 x 4 bins vs oracle, %-5s d%-36s rel-L2 %.2e" % (mode, name, l2))
            if l2 > tol:
                bad.append("%s %s: relative L2 error %.2e > %.0e" % (mode, name, l2, tol))
        REPORT.append("full-size C5 x 4 bins vs oracle, %-5s WORST weight-gradient rel-L2 %.2e" % (mode, worst))
    assert not bad, "\n".join(bad)


def test_bins_add_up_to_the_single_window_with_events_on_the_edges():
    """Advisor, round 5: K7's window is closed on both ends (the reference's one window), so with B > 1 an event whose timestamp
    equals an interior bin edge was counted in two bins.  kernels.event_bin_windows makes interior bins half-open: the per-bin
    polarity images must add up to the one-window image on a stream that has events EXACTLY on every edge."""
    from benerf_amd import kernels as K
    H, W, B = 24, 40, 4
    lo, up = 0.25, 0.75
    edges = torch.linspace(lo, up, B + 1, dtype=torch.float32).tolist()
    rng = np.random.default_rng(3)
    ts = np.sort(np.concatenate([rng.random(5000), np.repeat(np.asarray(edges, np.float64), 7)]))      # 7 events on each edge
    n = ts.size
    xs = torch.from_numpy(rng.integers(0, W, n).astype(np.int32)).to(DEV)
    ys = torch.from_numpy(rng.integers(0, H, n).astype(np.int32)).to(DEV)
    ps = torch.from_numpy((rng.integers(0, 2, n) * 2 - 1).astype(np.float32)).to(DEV)
    tsd = torch.from_numpy(ts).to(DEV)
    one = K.event_window_accumulate(xs, ys, ps, tsd, lo, up, H, W)
    per_bin = [K.event_window_accumulate(xs, ys, ps, tsd, a, b, H, W) for a, b in K.event_bin_windows(lo, up, B)]
    assert torch.equal(sum(per_bin), one), "bins must partition the window"
    counts = [int(((ts >= a) & (ts <= b)).sum()) for a, b in K.event_bin_windows(lo, up, B)]
    assert sum(counts) == int(((ts >= lo) & (ts <= up)).sum())
    naive = [K.event_window_accumulate(xs, ys, ps, tsd, edges[b], edges[b + 1], H, W) for b in range(B)]
    assert not torch.equal(sum(naive), one), "the stream is built so that closed bins double-count (otherwise this test shows nothing)"

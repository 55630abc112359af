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


def _hip_step(x, wl, B, mode, z_fine=None, event_bins_kw=True):
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

        def dd(d):
            return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
        accu = x["accu"].to(DEV) if event_bins_kw else x["accu"][0].to(DEV)
        losses = step.step(x["evt_ts"].to(DEV), torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV), accu,
                           x["img"].to(DEV), dd(x["d_e"]), dd(x["d_r"]), z_fine_forced=None if z_fine is None else z_fine.to(DEV))
        step.check_range()
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


def _compare(case, mode, losses, grads, ref_loss, ref_grads, tol_pose, tol_entry, tol_norm, tol_l2):
    rng = np.random.default_rng(7)
    assert abs(float(losses[0]) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (mode, float(losses[0]), ref_loss)
    bad, worst = [], {"pose": 0.0, "entries": 0.0, "norm": 0.0, "l2": 0.0}
    for name, ref in ref_grads.items():
        ref = ref.double()
        got = grads[name].double().reshape(ref.shape)
        mx = float(ref.abs().max())
        l2 = float((got - ref).norm() / ref.norm())
        worst["l2"] = max(worst["l2"], l2)
        if name in ("knots", "transform"):
            e = float((got - ref).abs().max()) / mx
            worst["pose"] = max(worst["pose"], e)
            if e > tol_pose or l2 > tol_l2:
                bad.append("%s %s: max %.2e rel-L2 %.2e" % (mode, name, e, l2))
            continue
        idx = torch.from_numpy(rng.integers(0, ref.numel(), 64))
        e = float((got.reshape(-1)[idx] - ref.reshape(-1)[idx]).abs().max()) / mx
        en = abs(float(got.norm() / ref.norm()) - 1.0)
        worst["entries"], worst["norm"] = max(worst["entries"], e), max(worst["norm"], en)
        if e > tol_entry or en > tol_norm or l2 > tol_l2:
            bad.append("%s %s: entries %.2e norm %.2e rel-L2 %.2e" % (mode, name, e, en, l2))
    REPORT.append("event bins %s, %-5s WORST pose %.2e  sampled entries %.2e  norms %.2e  rel-L2 %.2e" %
                  (case, mode, worst["pose"], worst["entries"], worst["norm"], worst["l2"]))
    assert not bad, "%s:\n%s" % (case, "\n".join(bad))


@pytest.mark.parametrize("spec", ["unreal_gray", "e2real_colour"])
def test_four_bins_vs_oracle_g8_size(spec):
    """B = 4 at G8 size, both arithmetic modes, against autograd through the binned oracle with its fine depths forced into the
    HIP step (sample_pdf's conditioning taken out, as in the G8 contract test): SURVEY 8c's tolerances."""
    from benerf_amd import workloads as WL
    B = 4
    base = "C2" if spec == "unreal_gray" else "C5"
    wl = dict(WL.WORKLOADS[base], S=16, Ni=16 if base == "C2" else 32, Re=24 if base == "C2" else 16, Rr=3 if base == "C2" else 2,
              n=19 if base == "C2" else 31)
    x = _inputs(np.random.default_rng(1900 + (base == "C5")), wl, B)
    cfg, Re = x["cfg"], wl["Re"]
    qc = {k: v.clone().requires_grad_(True) for k, v in x["pc"].items()}
    qf = {k: v.clone().requires_grad_(True) for k, v in x["pf"].items()}
    kn, tr = x["knots"].clone().requires_grad_(True), x["tr"].clone().requires_grad_(True)
    tacc = [x["accu"][b].double().reshape(-1, 1)[x["idx_e"]] for b in range(B)]
    loss, parts = O.step_loss_binned(cfg, qc, qf, kn, tr, x["evt_ts"], B, torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"], tacc,
                                     x["img"][x["idx_r"]], x["d_e"], x["d_r"])
    loss.backward()
    ref = {"knots": kn.grad, "transform": tr.grad}
    for tag, q in (("nerf", qc), ("nerf_fine", qf)):
        for k, v in q.items():
            ref[tag + "." + k] = v.grad
    z_fine = torch.cat([parts["extras_evt"]["z_fine"], parts["extras_rgb"]["z_fine"]]).detach()
    for mode in ("f32", "split"):
        losses, grads, _ = _hip_step(x, wl, B, mode, z_fine)
        report("event bins %s %s: event loss (sum over bins)" % (spec, mode), losses[1:2], parts["event"].detach().reshape(1).float(),
               atol=1e-6, rtol=2e-5)
        report("event bins %s %s: blur loss" % (spec, mode), losses[4:5], parts["rgb"].detach().reshape(1).float(), atol=1e-6, rtol=2e-5)
        _compare("G8-size " + spec, mode, losses, grads, float(loss.detach()), ref, 1e-3, 1e-3, 1e-4, 1e-3)


def test_four_bins_vs_oracle_c5_shape():
    """B = 4 on the C5 configuration (260 x 346 camera, colour, 31 blur poses, 64 + 192 samples, the L2-NORMALISED event loss of
    every bin): 1024 event pixels x 5 poses + 31 x 132 blur rays = 9 212 rays, 2.36 M sample points - the batch of a C5 step -
    against the chunked oracle (f64_truth.step_grads_vjp, event_bins = 4) with its fine depths forced in.  Tolerances: the
    full-size ones of tests/test_f64_truth_gpu.py (FULL_SIZE_TOL: sampled entries / pose gradients at twice SURVEY 8c, the
    whole-tensor relative L2 error at 8c's 1e-3)."""
    from benerf_amd import workloads as WL
    from test_f64_truth_gpu import FULL_SIZE_TOL
    B = 4
    wl = dict(WL.WORKLOADS["C5"], Re=1024)
    x = _inputs(np.random.default_rng(2031), wl, B)
    tacc = [x["accu"][b].double().reshape(-1, 1)[x["idx_e"]] for b in range(B)]
    o32 = T.step_grads_vjp(x["cfg"], x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"],
                           tacc, x["img"][x["idx_r"]], x["d_e"], x["d_r"], dtype=torch.float32, n_chunks=32, event_bins=B)
    z_fine = torch.cat([o32["z"]["evt"][1], o32["z"]["rgb"][1]])
    for mode in ("f32", "split"):
        losses, grads, _ = _hip_step(x, wl, B, mode, z_fine)
        _compare("C5-shape", mode, losses, grads, o32["loss"], o32["grads"], FULL_SIZE_TOL["pose"], FULL_SIZE_TOL["entries"],
                 FULL_SIZE_TOL["norm"], FULL_SIZE_TOL["l2"])

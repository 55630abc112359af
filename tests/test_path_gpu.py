"""GPU parity tests of the whole path through the reference-shaped API:
Graph.render vs golden G7, one full training iteration (loss + every gradient) vs golden G8
both through autograd (`loss.backward()` like train.py) and through the fused TrainStep,
the stand-alone helper mirrors, and size-independent properties at the full C2 size."""
import numpy as np
import pytest
import torch

import benerf_oracle as O
import golden_inputs as GI
from conftest import report

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("mlp_precision")]
DEV = "cuda:0"


class ReplayRNG:
    """torch.rand / torch.randn return prepared tensors (reference draw order)."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __enter__(self):
        self._rand, self._randn = torch.rand, torch.randn

        def pop(*shape, **kw):
            t = self.queue.pop(0)
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert tuple(t.shape) == shp, (t.shape, shp)
            return t.clone().to(kw.get("device", "cpu"))

        torch.rand = pop
        torch.randn = pop
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn = self._rand, self._randn
        assert not self.queue, "unused draws"


def build_graph(args, pc, pf, knots, transform):
    from benerf_amd import engine, kernels as K
    from benerf_amd.model import optimize
    torch.manual_seed(0)
    model = optimize.Model(args)
    model.graph.to(DEV)
    g = model.build_network(args)
    with torch.no_grad():
        for net, p in ((g.nerf, pc), (getattr(g, "nerf_fine", None), pf)):
            if net is None:      # N_importance == 0: no fine network (model/nerf.py:156-157)
                continue
            for name in K.LAYER_NAMES:
                lin = engine.getattr_path(net, name)
                lin.weight.copy_(p[name + ".weight"])
                lin.bias.copy_(p[name + ".bias"])
        g.evt_knot_pose_se3.params.weight.copy_(knots)
        g.transform.params.weight.copy_(transform)
    return model, g


def test_render_golden_g7(golden):
    from benerf_amd import workloads as WL
    g7 = golden("g7_render")
    cam = GI.CAMERAS["unreal"]
    Kmat = GI.cam_K(cam)
    ci = 0
    for C in (1, 3):
        for (P, Rn) in ((2, 32), (19, 4)):
            for (S, Ni) in ((16, 16), (64, 64)):
                rng = np.random.default_rng(707 + ci)
                tag = "C%d_P%d_S%d" % (C, P, S)
                ci += 1
                pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
                pc["alpha_linear.bias"] += 1.0
                pf["alpha_linear.bias"] += 1.0
                knots = GI.knots_init(rng) * 5
                tr = GI.transform_small(rng) * 0.1
                idx = GI.pixel_indices(rng, cam, Rn)
                draws = GI.render_draws(rng, P * Rn, S, Ni)
                args = WL.make_args("C2", channels=C, N_samples=S, N_importance=Ni, num_interpolated_pose=P)
                _, g = build_graph(args, pc, pf, knots, tr)
                poses = g.get_pose_rgb(args, torch.tensor([0.0, 1.0]), seg_num=P)
                report("render: get_pose_rgb " + tag, poses, g7[tag + "_poses"], atol=2e-6)
                for training in (True, False):
                    with ReplayRNG([draws["t_rand"], draws["noise0"], draws["u"], draws["noise1"]]):
                        ret = g.render(0, torch.from_numpy(g7[tag + "_poses"]).to(DEV), idx.to(DEV), cam["H"], cam["W"],
                                       Kmat, args, True, "rgb", torch.tensor([]), training=training)
                    assert set(ret) == {"rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "sigma"}
                    for k in ("rgb_map", "rgb0"):
                        report("render: %s %s train=%d" % (k, tag, training), ret[k], g7[tag + "_" + k], atol=1e-4)
                    for k in ("acc_map", "acc0"):
                        report("render: %s %s train=%d" % (k, tag, training), ret[k], g7[tag + "_" + k], atol=1e-4)
                    # sigma is sampled AT the importance samples: a sample that sits on a cdf knot moves by
                    # the conditioning of sample_pdf (see test_kernels_gpu K5) and the density follows the
                    # MLP's large d sigma / d z there.  Bulk tight, tail bounded.
                    sg, sr = ret["sigma"].detach().cpu().numpy(), g7[tag + "_sigma"]
                    ssc = max(1.0, float(np.abs(sr).max()))
                    assert np.mean(np.abs(sg - sr) <= 1e-4 * ssc) >= 0.999, "sigma bulk " + tag
                    report("render: sigma (tail) %s train=%d" % (tag, training), sg, sr, atol=2e-2 * ssc)
                    dref = g7[tag + "_disp_map"]
                    ok = np.isfinite(dref) & (dref < 1e6)
                    report("render: disp_map %s train=%d" % (tag, training), ret["disp_map"].detach().cpu().numpy()[ok], dref[ok],
                           atol=1e-4, rtol=2e-4)


def test_render_near_far_and_coarse_only():
    """Graph.render honours its near / far arguments (model/nerf.py:236-240,297-299) and N_importance == 0 (model/nerf.py:322,
    336-343: three keys, no fine pass) - against the oracle on replayed draws, forward and the gradient to the poses."""
    from benerf_amd import workloads as WL
    cam = GI.CAMERAS["unreal"]
    Kmat = GI.cam_K(cam)
    rng = np.random.default_rng(717)
    C, P, Rn, S = 1, 3, 16, 32
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    knots = GI.knots_init(rng) * 5
    idx = GI.pixel_indices(rng, cam, Rn)
    for Ni, near, far in ((32, 0.2, 0.8), (0, 0.0, 1.0), (0, 0.1, 0.9)):
        draws = GI.render_draws(rng, P * Rn, S, max(Ni, 1))
        args = WL.make_args("C2", channels=C, N_samples=S, N_importance=Ni, num_interpolated_pose=P)
        _, g = build_graph(args, pc, pf, knots, torch.zeros(1, 6))
        poses = O.trajectory_poses(knots, None, (0.0, 1.0), P, "spline")
        po = poses.clone().requires_grad_(True)
        ref = O.render(pc, pf, po, idx, cam["H"], cam["W"], Kmat, C, S, Ni, draws, exact_pdf=True, near=near, far=far)
        ph = poses.to(DEV).requires_grad_(True)
        queue = [draws["t_rand"], draws["noise0"]] + ([draws["u"], draws["noise1"]] if Ni > 0 else [])
        with ReplayRNG(queue):
            ret = g.render(0, ph, idx.to(DEV), cam["H"], cam["W"], Kmat, args, True, "rgb", torch.tensor([]), near=near, far=far,
                           training=True)
        tag = "Ni=%d near=%g far=%g" % (Ni, near, far)
        assert set(ret) == ({"rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "sigma"} if Ni > 0 else {"rgb_map", "disp_map", "acc_map"})
        for k in ("rgb_map", "acc_map") + (("rgb0",) if Ni > 0 else ()):
            report("render(%s) %s" % (tag, k), ret[k], ref[k], atol=1e-4)
        Gm = GI.f32(rng.standard_normal(tuple(ref["rgb_map"].shape)))
        (ref["rgb_map"] * Gm).sum().backward()
        (ret["rgb_map"] * Gm.to(DEV)).sum().backward()
        sc = float(po.grad.abs().max())
        report("render(%s) d poses" % tag, ph.grad, po.grad, atol=2e-3 * sc, rtol=2e-3)


G8_SPECS = [
    ("unreal_C1", "unreal", 1, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3),
    ("unreal_C3", "unreal", 3, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3),
    ("e2syn_C3", "e2nerf_syn", 3, "E2NeRF_Synthetic", 0.2, 7, 32, 32, 16, 5),
    ("e2real_C3", "e2nerf_real", 3, "E2NeRF_Real", -1.0, 31, 16, 32, 16, 2),
]


def _g8_inputs(si, spec):
    """Regenerates the inputs of gen_golden.g8_step (same rng stream)."""
    tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = spec
    rng = np.random.default_rng(808 + si)
    cam = GI.CAMERAS[cname]
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    knots = GI.knots_init(rng) * 3
    tr = GI.transform_small(rng) * 0.1
    idx_e = GI.pixel_indices(rng, cam, Re)
    idx_r = GI.pixel_indices(rng, cam, Rr)
    ev = GI.synthetic_events(rng, cam, 200000)
    window = 0.1 if "unreal" in tag else 0.25
    low_t = float(rng.random() * (1 - window))
    img = torch.from_numpy(rng.random((1, cam["H"], cam["W"], C)))
    d_e = GI.render_draws(rng, 2 * Re, S, Ni)
    d_r = GI.render_draws(rng, P * Rr, S, Ni)
    return dict(cam=cam, pc=pc, pf=pf, knots=knots, tr=tr, idx_e=idx_e, idx_r=idx_r, ev=ev, window=window, low_t=low_t,
                img=img, d_e=d_e, d_r=d_r)


def _check_grads(g8, tag, named, knots_g, tr_g, who, fine_entry_tol=2e-2, fine_norm_tol=1e-3, coarse_entry_tol=5e-3, pose_tol=2e-3):
    # Pose gradients (and bias gradients of the early layers) are sums over every sample point with
    # heavy cancellation (|sum| << sum|.|): f32 round-off of ANY summation order is ~1e-3 of the
    # largest entry, so tolerances are relative to that entry (1e-3 on entries).
    sc = float(np.abs(g8[tag + "_dknots"]).max())
    report("%s dknots %s" % (who, tag), knots_g, g8[tag + "_dknots"], atol=pose_tol * sc, rtol=pose_tol)
    report("%s dtransform %s" % (who, tag), tr_g, g8[tag + "_dtransform"], atol=pose_tol * sc, rtol=pose_tol)
    for key, got in named.items():
        base = "%s_g_%s" % (tag, key)
        flat = got.reshape(-1).detach().cpu().numpy()
        # coarse-net weight norms agree to ~1e-6; the fine net is evaluated AT the importance samples,
        # which inherit sample_pdf's conditioning (a sample next to a cdf knot moves by ~1e-5 when the
        # cdf differs by one ulp), so its gradient norms carry a ~1e-4 relative wobble on tiny batches.
        loose = key.endswith(".bias") or key.startswith("nerf_fine.")
        report("%s |d%s| %s" % (who, key, tag), np.array(np.linalg.norm(flat.astype(np.float64))),
               g8[base + "__norm"], atol=1e-12,
               rtol=(fine_norm_tol if key.startswith("nerf_fine.") else 1e-3) if loose else 2e-4)
        ref_v = g8[base + "__val"]
        report("%s d%s[64] %s" % (who, key, tag), flat[g8[base + "__idx"]], ref_v,
               atol=(fine_entry_tol if key.startswith("nerf_fine.") else coarse_entry_tol) * float(np.abs(ref_v).max()) + 1e-12, rtol=2e-3)


@pytest.mark.parametrize("si", range(len(G8_SPECS)))
def test_training_iteration_golden_g8_autograd(golden, si):
    """Reference-style driver: graph.forward pieces + train.py's loss lines on the mirrored
    helper modules + loss.backward() (train.py:160-340)."""
    from benerf_amd import workloads as WL
    from benerf_amd.loss import imgloss
    from benerf_amd.utils import img_utils, math_utils
    g8 = golden("g8_step")
    spec = G8_SPECS[si]
    tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = spec
    x = _g8_inputs(si, spec)
    cam = x["cam"]
    Kmat = GI.cam_K(cam)
    args = WL.make_args("C2", channels=C, N_samples=S, N_importance=Ni, num_interpolated_pose=P, dataset=dataset,
                        event_threshold=thr, event_height=cam["H"], event_width=cam["W"])
    _, g = build_graph(args, x["pc"], x["pf"], x["knots"], x["tr"])
    from benerf_amd.utils import event_utils
    sel, upper_t = O.event_window(x["ev"]["ts"], x["low_t"], x["window"])
    accu = event_utils.accumulate_events_on_gpu(np.zeros((cam["H"], cam["W"])), x["ev"]["x"][sel], x["ev"]["y"][sel],
                                                x["ev"]["pol"][sel])
    assert accu.dtype == torch.float64
    evt_ts = torch.tensor(np.stack((x["low_t"], upper_t)).reshape(2), dtype=torch.float32)
    pe = g.get_pose_evt(args, evt_ts)
    pr = g.get_pose_rgb(args, torch.tensor([0.0, 1.0]))
    d_e, d_r = x["d_e"], x["d_r"]
    with ReplayRNG([d_e["t_rand"], d_e["noise0"], d_e["u"], d_e["noise1"]]):
        ret_e = g.render(0, pe, x["idx_e"].to(DEV), cam["H"], cam["W"], Kmat, args, True, "event", torch.tensor([]),
                         training=True)
    with ReplayRNG([d_r["t_rand"], d_r["noise0"], d_r["u"], d_r["noise1"]]):
        ret_r = g.render(0, pr, x["idx_r"].to(DEV), cam["H"], cam["W"], Kmat, args, True, "rgb", torch.tensor([]),
                         training=True)
    for k, ref in (("rgb_map", "_rgb_map_evt"), ("rgb0", "_rgb0_evt")):
        report("step(autograd) evt %s %s" % (k, tag), ret_e[k], g8[tag + ref], atol=1e-4)
    for k, ref in (("rgb_map", "_rgb_map_rgb"), ("rgb0", "_rgb0_rgb")):
        report("step(autograd) rgb %s %s" % (k, tag), ret_r[k], g8[tag + ref], atol=1e-4)
    # train.py:163-337 on the mirrored helpers
    mse, gray, bl = imgloss.MSELoss(), img_utils.RGB2Gray(), math_utils.rgb2brightlog
    target_s = accu.reshape(-1, 1)[x["idx_e"].to(DEV)]

    def diff(imgs):
        a, b = imgs[:Re], imgs[Re:]
        if C == 3:
            return bl(gray(b), dataset) - bl(gray(a), dataset)
        return bl(b, dataset) - bl(a, dataset)

    if thr > 0:
        tgt = (target_s * torch.tensor(thr)).float()
        ef = mse(diff(ret_e["rgb_map"]), tgt) * args.event_coeff_syn
        ec = mse(diff(ret_e["rgb0"]), tgt) * args.event_coeff_syn
    else:
        def nrm(v):
            return v / (torch.linalg.norm(v, dim=0, keepdim=True) + 1e-9)
        tgt = nrm(target_s).float()
        ef = mse(nrm(diff(ret_e["rgb_map"])), tgt) * args.event_coeff_real
        ec = mse(nrm(diff(ret_e["rgb0"])), tgt) * args.event_coeff_real
    target_rgb = torch.Tensor(x["img"][0].numpy()).reshape(-1, cam["H"] * cam["W"], C)[:, x["idx_r"]].reshape(-1, C).to(DEV)
    sb = sum(ret_r["rgb_map"][j * Rr:(j + 1) * Rr] for j in range(P)) / P
    sb0 = sum(ret_r["rgb0"][j * Rr:(j + 1) * Rr] for j in range(P)) / P
    rf, rc = mse(sb, target_rgb) * args.rgb_coeff, mse(sb0, target_rgb) * args.rgb_coeff
    loss = (ec + ef) + (rf + rc)
    report("step(autograd) loss " + tag, loss, g8[tag + "_loss"].astype(np.float32), atol=1e-6, rtol=2e-4)
    report("step(autograd) event_fine " + tag, ef, g8[tag + "_event_fine"].astype(np.float32), atol=1e-7, rtol=2e-4)
    report("step(autograd) rgb_fine " + tag, rf, g8[tag + "_rgb_fine"].astype(np.float32), atol=1e-7, rtol=2e-4)
    loss.backward()
    named = {}
    for nn_, net in (("nerf", g.nerf), ("nerf_fine", g.nerf_fine)):
        for k, v in net.named_parameters():
            named[nn_ + "." + k] = v.grad
    _check_grads(g8, tag, named, g.evt_knot_pose_se3.params.weight.grad, g.transform.params.weight.grad, "step(autograd)")


@pytest.mark.parametrize("si", range(len(G8_SPECS)))
def test_training_iteration_golden_g8_fused(golden, si):
    """Same iteration through the fused TrainStep (K6 loss kernels, batched renders)."""
    from benerf_amd import engine, workloads as WL
    g8 = golden("g8_step")
    spec = G8_SPECS[si]
    tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = spec
    x = _g8_inputs(si, spec)
    cam = x["cam"]
    args = WL.make_args("C2", channels=C, N_samples=S, N_importance=Ni, num_interpolated_pose=P, dataset=dataset,
                        event_threshold=thr, event_height=cam["H"], event_width=cam["W"], optimize_trans=True)
    _, g = build_graph(args, x["pc"], x["pf"], x["knots"], x["tr"])
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    sel, upper_t = O.event_window(x["ev"]["ts"], x["low_t"], x["window"])
    from benerf_amd import kernels as K
    ev = x["ev"]
    accu = K.event_window_accumulate(torch.from_numpy(ev["x"].astype(np.int32)).to(DEV),
                                     torch.from_numpy(ev["y"].astype(np.int32)).to(DEV),
                                     torch.from_numpy(ev["pol"]).to(DEV), torch.from_numpy(ev["ts"]).to(DEV), x["low_t"],
                                     upper_t, cam["H"], cam["W"])
    ref_accu = O.accumulate_events(cam["H"], cam["W"], ev["x"][sel], ev["y"][sel], ev["pol"][sel])
    assert np.array_equal(accu.cpu().numpy(), ref_accu.numpy().astype(np.float32))
    img = torch.Tensor(x["img"][0].numpy()).reshape(cam["H"] * cam["W"], C).to(DEV)

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    evt_ts = torch.tensor(np.stack((x["low_t"], upper_t)).reshape(2), dtype=torch.float32).to(DEV)
    p_before = step.flat_p.clone()
    losses = step.step(evt_ts, torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV),
                       accu.view(-1), img, dd(x["d_e"]), dd(x["d_r"]))
    ref = np.array([g8[tag + "_loss"], g8[tag + "_event_fine"] + g8[tag + "_event_coarse"], g8[tag + "_event_fine"],
                    g8[tag + "_event_coarse"], g8[tag + "_rgb_fine"] + g8[tag + "_rgb_coarse"], g8[tag + "_rgb_fine"],
                    g8[tag + "_rgb_coarse"], 0.0], np.float32)
    report("step(fused) losses " + tag, losses, ref, atol=1e-6, rtol=2e-4)
    named = {}
    for nn_, fn in (("nerf", step.net_c), ("nerf_fine", step.net_f)):
        for i, name in enumerate(K.LAYER_NAMES):
            named["%s.%s.weight" % (nn_, name)] = fn.gviews_w[i]
            named["%s.%s.bias" % (nn_, name)] = fn.gviews_b[i]
    _check_grads(g8, tag, named, step.g_knots, step.g_transform, "step(fused)")
    # Fine-net gradients are looser than coarse-net ones for a structural reason: K5 is bit-exact given
    # identical inputs (test_kernels_gpu), but its input - the coarse weights - carries ~1e-7 f32 noise
    # between ANY two implementations, and sample_pdf amplifies that into ~1e-5 shifts of samples that sit
    # next to a cdf knot; the high-frequency PE turns those shifts into ~1e-3-of-max wobble on the
    # cancellation-heavy early-layer gradients.  The MLP / compositing backward kernels themselves are
    # pinned at 1e-6 on fixed inputs (K3 / K4 tests).  Loss and all coarse quantities stay tight here.
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C,
                       n_samples=S, n_importance=Ni, n_poses=P, dataset=dataset, threshold=thr, window=x["window"])
    oc = {k: v.clone().requires_grad_(True) for k, v in x["pc"].items()}
    of = {k: v.clone().requires_grad_(True) for k, v in x["pf"].items()}
    ko, to = x["knots"].clone().requires_grad_(True), x["tr"].clone().requires_grad_(True)
    target_acc = ref_accu.reshape(-1, 1)[x["idx_e"]]
    target_rgb = img.cpu()[x["idx_r"]]
    loss_o, _ = O.step_loss(cfg, oc, of, ko, to, evt_ts.cpu(), torch.tensor([0.0, 1.0]), x["idx_e"], x["idx_r"], target_acc,
                            target_rgb, x["d_e"], x["d_r"], exact_pdf=True)
    loss_o.backward()
    report("step(fused) loss vs exact-pdf oracle " + tag, losses[0], loss_o.float(), atol=1e-6, rtol=2e-5)
    for i, name in enumerate(K.LAYER_NAMES):
        r = of[name + ".weight"].grad
        report("step(fused) dnerf_fine.%s.weight vs exact-pdf oracle %s" % (name, tag), step.net_f.gviews_w[i], r,
               atol=3e-2 * float(r.abs().max()), rtol=2e-3)
        rc_ = oc[name + ".weight"].grad
        # ReLU-kink flips bound how close two f32 implementations can be on the layers behind a ReLU:
        # test_fine_pass_gradients_with_forced_samples measures the oracle's own 1-ulp sensitivity (1e-3..1e-2)
        report("step(fused) dnerf.%s.weight vs exact-pdf oracle %s" % (name, tag), step.net_c.gviews_w[i], rc_,
               atol=5e-3 * float(rc_.abs().max()), rtol=2e-3)
    # Adam touched every optimised parameter, parameters still alias the module's tensors
    assert not torch.equal(p_before, step.flat_p)
    assert g.nerf.pts_linears[0].weight.data_ptr() == step.net_c.views_w[0].data_ptr()
    first = O.decayed_lr(5e-4, 0.1, 0)
    moved = (step.flat_p - p_before).abs().max()
    assert float(moved) <= first * 1.0001, "first Adam step moves every parameter by at most lr"


def test_helper_mirrors(golden):
    """Stand-alone mirrors of the reference's public helpers vs golden vectors / oracle."""
    from benerf_amd import run_nerf_helpers as H, spline, workloads as WL
    from benerf_amd.model import embedder
    g2, g3, g6, g1 = golden("g2_rays"), golden("g3_posenc"), golden("g6_sample_pdf"), golden("g1_spline")
    for cname, cam in GI.CAMERAS.items():
        poses = torch.from_numpy(g2[cname + "_poses"]).to(DEV)
        idx = torch.from_numpy(g2[cname + "_idx"]).to(DEV)
        Kmat = GI.cam_K(cam)
        P, R = poses.shape[0], idx.shape[0]
        idx_ = idx.repeat(P)
        pp = poses.unsqueeze(1).repeat(1, R, 1, 1).reshape(-1, 3, 4)
        ro, rd = H.get_specific_rays(idx_ % cam["W"], idx_ // cam["W"], Kmat, pp)
        report("helpers get_specific_rays o " + cname, ro, g2[cname + "_rays_o"], atol=1e-6)
        report("helpers get_specific_rays d " + cname, rd, g2[cname + "_rays_d"], atol=1e-6)
        no, nd = H.ndc_rays(cam["H"], cam["W"], Kmat[0][0], 1.0, ro, rd)
        report("helpers ndc_rays o " + cname, no, g2[cname + "_ndc_o"], atol=2e-6, rtol=2e-6)
        report("helpers ndc_rays d " + cname, nd, g2[cname + "_ndc_d"], atol=2e-6, rtol=2e-6)
        args = WL.make_args("C2")
        fo, fd = H.get_rays(cam["H"], cam["W"], Kmat, poses[1], args, torch.tensor([]))
        report("helpers get_rays d " + cname, fd.reshape(-1, 3)[idx], g2[cname + "_rays_d"][R:2 * R], atol=1e-6)
    args = WL.make_args("C2")
    fn, dim = embedder.get_embedder(args, 10, 0)
    fnd, dimd = embedder.get_embedder(args, 4, 0)
    assert (dim, dimd) == (63, 27)
    report("helpers embed pts", fn(torch.from_numpy(g3["pts"]).to(DEV)), g3["pe"], atol=2e-6)
    report("helpers embed dirs", fnd(torch.from_numpy(g3["dirs"]).to(DEV)), g3["ped"], atol=2e-6)
    tag = "peaky_S64_N64"
    t_rand = torch.from_numpy(g6[tag + "_t_rand"])
    z = O.stratified_z(t_rand.shape[0], 64, t_rand)
    bins = 0.5 * (z[..., 1:] + z[..., :-1])
    u = torch.from_numpy(g6[tag + "_u"])
    with ReplayRNG([u]):
        s = H.sample_pdf(bins.to(DEV), torch.from_numpy(g6[tag + "_w"]).to(DEV), 64)
    assert np.array_equal(s.cpu().numpy(), g6[tag + "_samples_exact"]), "helpers sample_pdf must be bit-exact vs the oracle"
    tagc = "c05_spline"
    k = torch.from_numpy(g1[tagc + "_knots"]).to(DEV) + torch.from_numpy(g1[tagc + "_transform"]).to(DEV)
    k = k.requires_grad_(True)
    P = g1[tagc + "_poses"].shape[0]
    ts = torch.linspace(float(g1[tagc + "_ts"][0]), float(g1[tagc + "_ts"][1]), P).to(DEV)
    poses = spline.cubic_spline_pose_unit_time(k[0].reshape(1, 1, 6), k[1].reshape(1, 1, 6), k[2].reshape(1, 1, 6),
                                               k[3].reshape(1, 1, 6), ts)
    report("helpers cubic_spline_pose_unit_time", poses, g1[tagc + "_poses"], atol=2e-6)
    (poses * torch.from_numpy(g1[tagc + "_G"]).to(DEV)).sum().backward()
    report("helpers spline backward", k.grad, g1[tagc + "_dknots"], atol=2e-5 * float(np.abs(g1[tagc + "_dknots"]).max()),
           rtol=1e-3)
    tagl = "c05_linear"
    lin = spline.linear_pose_unit_time(k[0].detach().reshape(1, 1, 6), k[3].detach().reshape(1, 1, 6), ts)
    report("helpers linear_pose_unit_time", lin, g1[tagl + "_poses"], atol=2e-6)


@pytest.mark.parametrize("wl_name", ["C2", "C3", "C4", "C5"])
def test_full_size_properties(wl_name):
    """Checks at the full size of every BASELINE.json GPU configuration (C2: gray; C3: colour; C4: 800 x 800 camera, linlog,
    8181 rays; C5: 31 poses, 64 + 192 samples) that need no oracle run: compositing weights sum to acc <= 1, merged depths
    sorted and a superset of the coarse depths, MLP linear in d_raw (backward), Philox reproducibility of the whole
    render - the blur batch of one step (n poses x Rr pixels)."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = WL.WORKLOADS[wl_name]
    args = WL.make_args(wl_name)
    C, S, Ni = wl["channels"], wl["S"], wl["Ni"]
    rng = np.random.default_rng(11)
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    _, g = build_graph(args, pc, pf, GI.knots_init(rng), torch.zeros(1, 6))
    cam = WL.CAMERAS[wl["cam"]]
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    poses = g.get_pose_rgb(args, torch.tensor([0.0, 1.0])).detach()
    assert poses.shape[0] == wl["n"]
    idx = torch.from_numpy(rng.permutation(cam["H"] * cam["W"])[:wl["Rr"]]).to(DEV)
    net_c, net_f = g.nerf.packed(), g.nerf_fine.packed()
    net_c.pack_if_stale()
    net_f.pack_if_stale()
    d = engine.Draws(seed=5, offset=3)
    out, saved = engine._render_forward(cam_o, True, S, Ni, d, poses, idx, net_c, net_f, True)
    out2, _ = engine._render_forward(cam_o, True, S, Ni, d, poses, idx, net_c, net_f, False)
    assert torch.equal(out["rgb_map"], out2["rgb_map"]), "same Philox stream => identical render"
    zf, zc = saved["z_fine"], saved["z"]
    assert bool((zf[:, 1:] >= zf[:, :-1]).all()), "merged depths must be sorted"
    pos = torch.searchsorted(zf, zc.contiguous())
    assert bool((torch.gather(zf, 1, pos.clamp(max=zf.shape[1] - 1)) == zc).all()), "coarse depths survive the merge"
    acc = out["acc_map"]
    assert float(acc.max()) <= 1.0 + 1e-5 and float(acc.min()) >= 0.0
    assert float(out["rgb_map"].min()) >= 0.0 and float(out["rgb_map"].max()) <= 1.0 + 1e-5
    # backward is linear in the upstream gradient: bwd(2g) == 2 bwd(g)
    N = zc.shape[0]
    assert out["rgb_map"].shape == (wl["n"] * wl["Rr"], C) and zf.shape == (N, S + Ni)
    graw = torch.randn(N * (S + Ni), C + 1, device=DEV)
    gw1 = [torch.zeros_like(w) for w in net_f.weights]
    gb1 = [torch.zeros_like(b) for b in net_f.biases]
    gw2 = [torch.zeros_like(w) for w in net_f.weights]
    gb2 = [torch.zeros_like(b) for b in net_f.biases]
    dp1, _ = K.mlp_bwd(net_f, graw, saved["acts1"], N, S + Ni, gw1, gb1, False)
    dp1 = dp1.clone()
    dp2, _ = K.mlp_bwd(net_f, (2 * graw).contiguous(), saved["acts1"], N, S + Ni, gw2, gb2, False)
    report("full-size linearity d_pts", dp2, 2 * dp1, atol=1e-5 * float(dp1.abs().max()), rtol=1e-5)
    report("full-size linearity dW4", gw2[4], 2 * gw1[4], atol=1e-5 * float(gw1[4].abs().max()), rtol=1e-5)


@pytest.mark.parametrize("wl_name", ["C3", "C4", "C5"])
def test_full_size_step_modes_agree(wl_name):
    """One full-size fused training step of C3 / C4 / C5 (colour kernels at 0.78 M points, the 800 x 800 camera with
    the linlog loss, 31 poses with 64 + 192 samples and the globally normalised loss) in both arithmetic modes on the same
    explicit draws: identical event image, losses to 1e-5, pose gradients within the contract's 1e-3 of the largest."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = WL.WORKLOADS[wl_name]
    cam = WL.CAMERAS[wl["cam"]]
    C, S, Ni, P, Re, Rr = wl["channels"], wl["S"], wl["Ni"], wl["n"], wl["Re"], wl["Rr"]
    HW = cam["H"] * cam["W"]
    res = {}
    prev = K.get_mlp_precision()
    try:
        for mode in ("f32", "split"):
            K.set_mlp_precision(mode)
            rng = np.random.default_rng(21)
            args = WL.make_args(wl_name, optimize_trans=True)
            pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
            pc["alpha_linear.bias"] += 1.0
            pf["alpha_linear.bias"] += 1.0
            _, g = build_graph(args, pc, pf, GI.knots_init(rng) * 3, GI.transform_small(rng) * 0.1)
            cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
            step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
            idx_e = torch.from_numpy(rng.permutation(HW)[:Re]).to(DEV)
            idx_r = torch.from_numpy(rng.permutation(HW)[:Rr]).to(DEV)
            accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
            img = torch.from_numpy(rng.random((HW, C)).astype(np.float32)).to(DEV)
            g_t = torch.Generator(device=DEV)
            g_t.manual_seed(5)

            def draws(n):
                return engine.Draws(torch.rand((n, S), device=DEV, generator=g_t), torch.randn((n, S), device=DEV, generator=g_t),
                                    torch.rand((n, Ni), device=DEV, generator=g_t), torch.randn((n, S + Ni), device=DEV, generator=g_t))
            losses = step.step(torch.tensor([0.3, 0.3 + wl["window"]], device=DEV), torch.tensor([0.0, 1.0], device=DEV), idx_e, idx_r,
                               accu, img, draws(2 * Re), draws(P * Rr))
            step.check_range()
            res[mode] = (losses.cpu().numpy(), step.g_knots.cpu().numpy().copy(), step.g_transform.cpu().numpy().copy(),
                         step.net_f.gviews_w[7].cpu().numpy().copy())
            del step, g
            torch.cuda.empty_cache()
    finally:
        K.set_mlp_precision(prev)
    a, b = res["f32"], res["split"]
    assert np.isfinite(b[0]).all() and np.isfinite(b[1]).all()
    report("full-size step %s: losses, split vs f32" % wl_name, b[0], a[0], atol=1e-7, rtol=2e-5)
    sc = float(np.abs(a[1]).max())
    # 1e-3 of the largest entry + the ReLU-kink flips any two forward arithmetics show (test_kernels_gpu)
    report("full-size step %s: d knots, split vs f32" % wl_name, b[1], a[1], atol=2e-3 * sc, rtol=2e-3)
    report("full-size step %s: d transform, split vs f32" % wl_name, b[2], a[2], atol=2e-3 * sc, rtol=2e-3)
    assert float(np.linalg.norm(b[3] - a[3]) / np.linalg.norm(a[3])) < 3e-3


@pytest.mark.parametrize("si", range(len(G8_SPECS)))
def test_training_iteration_golden_g8_forced_fine_depths(golden, si):
    """G8 again, with sample_pdf's conditioning taken out: the fused step is handed the fine depths the REFERENCE used
    (the oracle's op-for-op sample_pdf_torch on its own coarse weights - bit-identical to the reference's on the CPU,
    tests/golden/PINNING_REPORT.txt), so the fine network is evaluated at the same points as in the golden run and its
    gradients are held to the same tolerances as the coarse network's."""
    from benerf_amd import engine, kernels as K, workloads as WL
    g8 = golden("g8_step")
    spec = G8_SPECS[si]
    tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = spec
    x = _g8_inputs(si, spec)
    cam = x["cam"]
    Kmat = GI.cam_K(cam)
    args = WL.make_args("C2", channels=C, N_samples=S, N_importance=Ni, num_interpolated_pose=P, dataset=dataset,
                        event_threshold=thr, event_height=cam["H"], event_width=cam["W"], optimize_trans=True)
    _, g = build_graph(args, x["pc"], x["pf"], x["knots"], x["tr"])
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    sel, upper_t = O.event_window(x["ev"]["ts"], x["low_t"], x["window"])
    ref_accu = O.accumulate_events(cam["H"], cam["W"], x["ev"]["x"][sel], x["ev"]["y"][sel], x["ev"]["pol"][sel])
    img = torch.Tensor(x["img"][0].numpy()).reshape(cam["H"] * cam["W"], C).to(DEV)
    evt_ts = torch.tensor(np.stack((x["low_t"], upper_t)).reshape(2), dtype=torch.float32)
    with torch.no_grad():   # the reference's fine depths of both renders
        pe = O.trajectory_poses(x["knots"], None, evt_ts, 2, "spline")
        pr = O.trajectory_poses(x["knots"], x["tr"], torch.tensor([0.0, 1.0]), P, "spline")
        _, ex_e = O.render(x["pc"], x["pf"], pe, x["idx_e"], cam["H"], cam["W"], Kmat, C, S, Ni, x["d_e"], want_extras=True)
        _, ex_r = O.render(x["pc"], x["pf"], pr, x["idx_r"], cam["H"], cam["W"], Kmat, C, S, Ni, x["d_r"], want_extras=True)
    z_fine = torch.cat([ex_e["z_fine"], ex_r["z_fine"]]).to(DEV)

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    losses = step.step(evt_ts.to(DEV), torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV),
                       ref_accu.float().reshape(-1).to(DEV), img, dd(x["d_e"]), dd(x["d_r"]), z_fine_forced=z_fine)
    report("step(forced z) loss " + tag, losses[0:1], np.array([g8[tag + "_loss"]], np.float32), atol=1e-6, rtol=2e-5)
    named = {}
    for nn_, fn in (("nerf", step.net_c), ("nerf_fine", step.net_f)):
        for i, name in enumerate(K.LAYER_NAMES):
            named["%s.%s.weight" % (nn_, name)] = fn.gviews_w[i]
            named["%s.%s.bias" % (nn_, name)] = fn.gviews_b[i]
    # THE contract test of the gradients: 1e-3 of the largest entry on the pose gradients and on 64 sampled entries
    # per layer, 1e-4 on the norms - every spec (the L2-normalised e2real loss included), both arithmetic modes, no exceptions:
    # the split mode's backward GEMMs take 22-bit operands (round 4), what they contribute on identical inputs and masks is
    # float32's own error (tests/test_f64_truth_gpu.py::test_mlp_backward_arithmetic_vs_float64).
    _check_grads(g8, tag, named, step.g_knots, step.g_transform, "step(forced z)", fine_entry_tol=1e-3, fine_norm_tol=1e-4,
                 coarse_entry_tol=1e-3, pose_tol=1e-3)


def test_fine_pass_gradients_with_forced_samples():
    """Isolates the fine pass from sample_pdf's conditioning: the HIP kernels are fed the oracle's own
    merged depths (clustered importance samples, near-duplicate z), then fine + coarse MLP / compositing
    forward and backward are compared with the oracle's autograd at tight tolerance."""
    from benerf_amd import kernels as K
    rng = np.random.default_rng(99)
    C, S, Ni, P, Rn = 3, 16, 32, 31, 3
    cam = GI.CAMERAS["e2nerf_real"]
    Kmat = GI.cam_K(cam)
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    poses = O.trajectory_poses(GI.knots_init(rng) * 3, None, (0.0, 1.0), P, "spline").detach()
    idx = GI.pixel_indices(rng, cam, Rn)
    N = P * Rn
    draws = GI.render_draws(rng, N, S, Ni)
    G1, G0 = GI.f32(rng.standard_normal((N, C))), GI.f32(rng.standard_normal((N, C)))
    oc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    of = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
    ret, ex = O.render(oc, of, poses, idx, cam["H"], cam["W"], Kmat, C, S, Ni, draws, exact_pdf=True, want_extras=True)
    ex["raw1"].retain_grad()
    ((ret["rgb_map"] * G1).sum() + (ret["rgb0"] * G0).sum()).backward()

    def dev(t):
        return t.detach().to(DEV).contiguous()

    def net_of(p):
        n = K.PackedMlp([dev(p[nm + ".weight"]) for nm in K.LAYER_NAMES], [dev(p[nm + ".bias"]) for nm in K.LAYER_NAMES], C)
        n.pack()
        return n

    net_c, net_f = net_of(pc), net_of(pf)
    ro, rd, vd = K.rays_fwd(dev(poses), dev(idx), cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], True)
    z = K.stratified_z(N, S, DEV, dev(draws["t_rand"]))
    z_fine = dev(ex["z_fine"])                              # forced: the oracle's merged depths
    raw1, acts1 = K.mlp_fwd(net_f, ro, rd, vd, z_fine, True)
    c1 = K.composite_fwd(raw1, z_fine, rd, dev(draws["noise1"]), want=("rgb_map",))
    report("forced-z fine rgb_map", c1["rgb_map"], ret["rgb_map"], atol=2e-6)
    d_raw1, _ = K.composite_bwd(raw1, z_fine, rd, dev(draws["noise1"]), 0.0, 0, 0, dev(G1))
    ref_draw = ex["raw1"].grad
    report("forced-z d_raw (compositing backward)", d_raw1, ref_draw, atol=2e-5 * float(ref_draw.abs().max()), rtol=1e-3)

    # Self-calibrated tolerance.  Layers behind a ReLU are discontinuous in their inputs: a pre-activation
    # within f32 round-off of zero is "on" in one implementation and "off" in another, which flips a whole
    # rank-1 term of dW.  Measure the ORACLE's own sensitivity: the same fine pass with every depth moved
    # by one ulp; the HIP result has to be as close to the oracle as the oracle is to its perturbed self.
    def oracle_fine_grads(zf):
        p = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
        pts = ex["rays_o"].detach()[:, None, :] + ex["rays_d"].detach()[:, None, :] * zf[:, :, None]
        raw = O.mlp_forward(p, pts, ex["viewdirs"].detach())
        rgb = O.composite(raw, zf, ex["rays_d"].detach(), draws["noise1"], C)[0]
        (rgb * G1).sum().backward()
        return {k: v.grad for k, v in p.items()}

    zf0 = ex["z_fine"].detach()
    base = oracle_fine_grads(zf0)
    pert = oracle_fine_grads(zf0 * (1.0 + 2.0 ** -23))
    gw = [torch.zeros_like(w) for w in net_f.weights]
    gb = [torch.zeros_like(b) for b in net_f.biases]
    K.mlp_bwd(net_f, d_raw1.view(-1, C + 1), acts1, N, S + Ni, gw, gb, False)
    errs = []
    for i, name in enumerate(K.LAYER_NAMES):
        for kind, got in (("weight", gw[i]), ("bias", gb[i])):
            key = "%s.%s" % (name, kind)
            r = base[key]
            own = float((pert[key] - r).abs().max())
            sc = float(r.abs().max())
            try:
                report("forced-z dnerf_fine.%s (oracle 1-ulp sensitivity %.1e)" % (key, own / sc), got, r,
                       atol=4.0 * own + 2e-5 * sc, rtol=2e-3)
            except AssertionError as e:
                errs.append(str(e))
    assert not errs, "\n".join(errs)

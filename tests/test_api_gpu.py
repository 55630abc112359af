"""GPU tests of the reference-shaped API surface beyond render(): Graph.forward (event window +
both renders), render_video / render_image_test (full-image inference), checkpoint
compatibility (state-dict keys and shapes of the reference), and the
data-parallel TrainStep: two ranks sharding the global batch == one rank rendering all of it."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import benerf_oracle as O
import golden_inputs as GI
from conftest import report

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("mlp_precision")]
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# enumerated by instantiating the reference, C = 1
REFERENCE_STATE_DICT_SHAPES = {
    "pts_linears.0.weight": (256, 63), "pts_linears.1.weight": (256, 256), "pts_linears.2.weight": (256, 256),
    "pts_linears.3.weight": (256, 256), "pts_linears.4.weight": (256, 256), "pts_linears.5.weight": (256, 319),
    "pts_linears.6.weight": (256, 256), "pts_linears.7.weight": (256, 256), "views_linears.0.weight": (128, 283),
    "feature_linear.weight": (256, 256), "alpha_linear.weight": (1, 256), "rgb_linear.weight": (1, 128),
}


def _graph(args, seed=0, device=DEV):
    from benerf_amd import engine, kernels as K
    from benerf_amd.model import optimize
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = optimize.Model(args)
    model.graph.to(device)
    g = model.build_network(args)
    with torch.no_grad():
        for net in (g.nerf, g.nerf_fine):
            p = O.xavier_params(rng, args.channels)
            p["alpha_linear.bias"] += 1.0
            for name in K.LAYER_NAMES:
                lin = engine.getattr_path(net, name)
                lin.weight.copy_(p[name + ".weight"])
                lin.bias.copy_(p[name + ".bias"])
        g.evt_knot_pose_se3.params.weight.copy_(GI.knots_init(rng) * 3)
    return model, g


def test_state_dict_matches_reference_keys():
    from benerf_amd import workloads as WL
    args = WL.make_args("C2")
    model, g = _graph(args)
    sd = g.state_dict()
    for net in ("nerf", "nerf_fine"):
        for k, shp in REFERENCE_STATE_DICT_SHAPES.items():
            assert tuple(sd["%s.%s" % (net, k)].shape) == shp, k
            assert ("%s.%s" % (net, k.replace("weight", "bias"))) in sd
    assert tuple(sd["evt_knot_pose_se3.params.weight"].shape) == (4, 6)
    assert tuple(sd["rgb_knot_pose_se3.params.weight"].shape) == (4, 6)
    assert tuple(sd["transform.params.weight"].shape) == (1, 6)
    for k in ("rgb_crf.mlp_gray.0.weight", "rgb_crf.mlp_gray.2.weight", "event_crf.mlp_luminance.0.weight",
              "event_crf.mlp_luminance.2.bias"):
        assert k in sd, k
    # round trip through the reference's checkpoint format (train.py:443-455)
    optims = model.setup_optimizer(args)
    assert len(optims) == 5
    import io
    buf = io.BytesIO()
    torch.save({"global_step": 7, "graph": sd, "optimizer_nerf": optims[0].state_dict()}, buf)
    buf.seek(0)
    ck = torch.load(buf)
    _, g2 = _graph(args, seed=1)
    g2.load_state_dict(ck["graph"])
    assert torch.equal(g2.nerf.pts_linears[5].weight, g.nerf.pts_linears[5].weight)


def test_graph_forward_training_api():
    """Graph.forward: same return tuple / shapes / dtypes as model/nerf.py:160-234; the event image equals
    the oracle's accumulation of the window it drew; autograd reaches every parameter group."""
    from benerf_amd import workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=32, Rr=3, n=5)
    args = WL.make_args(wl)
    cam = WL.CAMERAS["unreal"]
    model, g = _graph(args)
    rng = np.random.default_rng(3)
    ev = GI.synthetic_events(rng, cam, 50000)
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
    np.random.seed(5)
    ret_e, ret_r, idx_e, idx_r, accu = g.forward(0, ev, np.array([0.0, 1.0]), cam["H"], cam["W"], K, K, args,
                                                 np.array([]), np.array([]))
    np.random.seed(5)
    low_t = np.random.rand(1) * (1 - args.accumulate_time_length)
    sel, _ = O.event_window(ev["ts"], low_t, args.accumulate_time_length)
    ref_accu = O.accumulate_events(cam["H"], cam["W"], ev["x"][sel], ev["y"][sel], ev["pol"][sel])
    assert accu.dtype == torch.float64 and accu.is_cuda
    assert np.array_equal(accu.cpu().numpy(), ref_accu.numpy())
    assert idx_e.shape == (32,) and idx_r.shape == (3,)
    assert ret_e["rgb_map"].shape == (64, 1) and ret_r["rgb_map"].shape == (15, 1) and ret_r["sigma"].shape == (15, 32)
    loss = ret_e["rgb_map"].mean() + ret_r["rgb0"].mean()
    loss.backward()
    assert g.evt_knot_pose_se3.params.weight.grad.abs().sum() > 0
    assert g.transform.params.weight.grad.abs().sum() > 0
    assert g.nerf.pts_linears[0].weight.grad.abs().sum() > 0 and g.nerf_fine.rgb_linear.weight.grad.abs().sum() > 0


def test_render_video_and_image_test(tmp_path):
    """Full-image inference through render_video (chunked) equals one un-chunked oracle render with the
    same draws; render_image_test returns 8-bit images."""
    from benerf_amd import run_nerf_helpers as H, workloads as WL
    from test_path_gpu import ReplayRNG
    Hh, Ww = 12, 20
    args = WL.make_args("C2", N_samples=16, N_importance=16, chunk=64)
    model, g = _graph(args)
    rng = np.random.default_rng(4)
    K = torch.tensor([[30.0, 0, Ww / 2], [0, 30.0, Hh / 2], [0, 0, 1]])
    poses = g.get_pose_rgb(args, [0, 1], seg_num=3).detach()
    pose = poses[1:2]
    n = Hh * Ww
    draws = GI.render_draws(rng, n, 16, 16)
    q = []
    for i in range(0, n, args.chunk):   # the reference draws per chunk, in chunk order
        q += [draws[k][i:i + args.chunk] for k in ("t_rand", "noise0", "u", "noise1")]
    with ReplayRNG(q):
        ret = g.render_video(0, pose, Hh, Ww, K, args, np.array([]), type="rgb")
    assert ret["rgb_map"].shape == (Hh, Ww, 1) and ret["disp_map"].shape == (Hh, Ww)
    pc = {k: v.detach().cpu() for k, v in g.nerf.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in g.nerf_fine.state_dict().items()}
    ref = O.render(pc, pf, pose.cpu(), torch.arange(n), Hh, Ww, K, 1, 16, 16, draws, exact_pdf=True)
    report("render_video rgb_map vs oracle", ret["rgb_map"].reshape(-1, 1), ref["rgb_map"], atol=1e-4)
    report("render_video acc_map vs oracle", ret["acc_map"].reshape(-1), ref["acc_map"], atol=1e-4)
    # where that difference comes from (round-5 review: 6.1e-5 in the split mode against 2.4e-5 in exact f32 on these 240 rays):
    # the same pixels with the ORACLE's fine depths forced in - what is left is arithmetic, the rest was sample_pdf's conditioning
    from benerf_amd import engine
    _, ex = O.render(pc, pf, pose.cpu(), torch.arange(n), Hh, Ww, K, 1, 16, 16, draws, exact_pdf=True, want_extras=True)
    with torch.no_grad():
        d_all = engine.Draws(*(draws[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
        out_f, sv = engine._render_forward(engine.Camera.from_K(Hh, Ww, K), True, 16, 16, d_all, pose.contiguous(), torch.arange(n, device=DEV),
                                           g.nerf.packed(), g.nerf_fine.packed(), False, z_fine_forced=ex["z_fine"].to(DEV))
        out_o, sv_o = engine._render_forward(engine.Camera.from_K(Hh, Ww, K), True, 16, 16, d_all, pose.contiguous(), torch.arange(n, device=DEV),
                                             g.nerf.packed(), g.nerf_fine.packed(), False)
    n_diff = int((sv_o["z_fine"].cpu() != ex["z_fine"]).any(1).sum())
    report("render_video rgb_map vs oracle, oracle's fine depths forced (%d of %d rays merged other depths on their own)" % (n_diff, n),
           out_f["rgb_map"], ref["rgb_map"], atol=2e-5)
    args.optimize_rgb_crf = False
    imgs, depth = H.render_image_test(5, g, poses, Hh, Ww, K, args, str(tmp_path), np.array([]), dir="images_test",
                                      need_depth=True)
    assert len(imgs) == 3 and imgs[0].dtype == np.uint8 and imgs[0].shape == (Hh, Ww, 1) and len(depth) == 3
    rgbs, disps = H.render_video_test(5, g, poses, Hh, Ww, K, args, np.array([]))
    assert rgbs.shape == (3, Hh, Ww, 1) and disps.shape == (3, Hh, Ww) and rgbs.dtype == np.float32
    # the device-side 8-bit conversion is utils/img_utils.to8bit (255 * clip(x, 0, 1) truncated), bit for bit
    from benerf_amd.utils import img_utils
    x = np.concatenate([np.random.default_rng(0).uniform(-0.2, 1.2, 4096), [0.0, 1.0, 1.0 / 255, 254.999 / 255, 0.5]]).astype(np.float32)
    assert np.array_equal(H._quantise(torch.from_numpy(x).to(DEV)).cpu().numpy(), img_utils.to8bit(x))
    for d8, d in zip(depth, depth):
        assert d8.max() == 255            # per-frame disparity normalisation


def _oracle_frame(pc, pf, pose, n, Hh, Ww, Kt, C, S, Ni, draws, chunk, z_forced=None, device=DEV, rows=None):
    """The oracle's render of pixels `rows` (default: all n) of one frame in chunks of `chunk` rays, its torch parts evaluated on
    `device` (float32 torch either way; sample_pdf_exact is numpy on the host).  z_forced = (z_coarse, z_fine) indexed like `rows`.
    Returns {rgb_map, acc_map, disp_map, z_coarse, z_fine} for those rows."""
    rows = torch.arange(n) if rows is None else rows
    out = {k: [] for k in ("rgb_map", "acc_map", "disp_map", "z_coarse", "z_fine")}
    with torch.device(device), torch.no_grad():
        qc = {k: v.to(device) for k, v in pc.items()}
        qf = {k: v.to(device) for k, v in pf.items()}
        for i in range(0, rows.numel(), chunk):
            r = rows[i:i + chunk]
            d = {k: v[r].to(device) for k, v in draws.items()}
            zf = None if z_forced is None else tuple(t[i:i + chunk].to(device) for t in z_forced)
            ret, ex = O.render(qc, qf, pose.to(device), r.to(device), Hh, Ww, Kt.to(device), C, S, Ni, d, exact_pdf=True, want_extras=True,
                               z_forced=zf)
            for k in ("rgb_map", "acc_map", "disp_map"):
                out[k].append(ret[k].float().cpu())
            out["z_coarse"].append(ex["z_coarse"].float().cpu())
            out["z_fine"].append(ex["z_fine"].float().cpu())
    return {k: torch.cat(v) for k, v in out.items()}


def test_full_resolution_frame_vs_oracle():
    """One full-resolution frame - 480 x 768, 64 + 128 samples, chunk = 4096: what test.py:112-135 / train.py:404-441 render through
    Graph.render_video (model/nerf.py:353-390) - against the oracle on the same draws, per pixel, at north_star's 1e-4:
      (1) the oracle's fine depths forced in: ZERO pixels beyond 1e-4 (rgb, acc; disparity 1e-4 relative);
      (2) render_video with its OWN importance samples against the oracle with ITS own (sample_pdf_exact): the two coarse passes
          agree to ~1e-6, so nearly every ray merges depths that differ in their last bits, and the fine network - positional
          encoding up to 2^9 x - turns a few ulps of depth into 1e-5 .. 1e-4 of colour.  Every pixel beyond 1e-4 must be EXPLAINED
          by exactly that: the oracle evaluated at the HIP path's depths reproduces the HIP colour within 1e-4.
    The oracle's torch parts run on the GPU box's device for the whole frame (368 640 rays x 192 samples: minutes on the host);
    two chunks are re-evaluated on the host at the same depths and must agree with the device evaluation to 2e-6."""
    from benerf_amd import engine, kernels as Kk, workloads as WL
    from conftest import REPORT
    from test_path_gpu import ReplayRNG
    cam = WL.CAMERAS["unreal"]
    Hh, Ww, S, Ni, chunk, C = cam["H"], cam["W"], 64, 64, 4096, 1
    args = WL.make_args("C2")
    assert args.N_samples == S and args.N_importance == Ni and args.chunk == chunk
    model, g = _graph(args, seed=5)
    fallback_before = Kk.auto_fallback_max(DEV)      # other tests of the session overflow the f16 range on purpose
    rng = np.random.default_rng(6)
    Kt = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=torch.float32)
    pose = g.get_pose_rgb(args, [0, 1], seg_num=3).detach()[1:2]
    n = Hh * Ww
    draws = GI.render_draws(rng, n, S, Ni)
    q = []
    for i in range(0, n, chunk):
        q += [draws[k][i:i + chunk] for k in ("t_rand", "noise0", "u", "noise1")]
    with ReplayRNG(q):
        ret = g.render_video(0, pose, Hh, Ww, Kt, args, np.array([]), type="rgb")
    assert ret["rgb_map"].shape == (Hh, Ww, C)
    pc = {k: v.detach().cpu() for k, v in g.nerf.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in g.nerf_fine.state_dict().items()}
    ref = _oracle_frame(pc, pf, pose.cpu(), n, Hh, Ww, Kt, C, S, Ni, draws, chunk)
    # the device evaluation of the oracle IS the oracle: two chunks (the first and one in the middle of the frame) on the host, at the
    # device evaluation's depths (left to themselves the two would draw depths that differ in their last bits, like any two evaluations)
    rows = torch.cat([torch.arange(0, chunk), torch.arange(45 * chunk, 46 * chunk)])
    host = _oracle_frame(pc, pf, pose.cpu(), n, Hh, Ww, Kt, C, S, Ni, draws, chunk, device="cpu", rows=rows,
                         z_forced=(ref["z_coarse"][rows], ref["z_fine"][rows]))
    report("oracle evaluated on the device vs on the host (same depths), rgb_map of %d rays" % rows.numel(), ref["rgb_map"][rows], host["rgb_map"], atol=2e-6)

    cam_o = engine.Camera.from_K(Hh, Ww, Kt)
    net_c, net_f = g.nerf.packed(), g.nerf_fine.packed()

    def hip_frame(z_forced=None):
        """the kernel sequence of render_video's chunks (engine._render_forward), to read the merged depths / force them"""
        maps = {k: [] for k in ("rgb_map", "acc_map", "disp_map", "z", "z_fine")}
        with torch.no_grad():
            for i in range(0, n, chunk):
                d = engine.Draws(*(draws[k][i:i + chunk].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
                out, saved = engine._render_forward(cam_o, True, S, Ni, d, pose.contiguous(), torch.arange(i, min(i + chunk, n), device=DEV), net_c,
                                                    net_f, False, z_fine_forced=None if z_forced is None else z_forced[i:i + chunk].to(DEV))
                for k in ("rgb_map", "acc_map", "disp_map"):
                    maps[k].append(out[k].cpu())
                maps["z"].append(saved["z"].cpu())
                maps["z_fine"].append(saved["z_fine"].cpu())
        return {k: torch.cat(v) for k, v in maps.items()}

    # (1) the oracle's fine depths forced in
    forced = hip_frame(ref["z_fine"])
    report("full frame, forced fine depths: rgb_map per pixel (%d pixels)" % n, forced["rgb_map"], ref["rgb_map"], atol=1e-4)
    report("full frame, forced fine depths: acc_map per pixel", forced["acc_map"], ref["acc_map"], atol=1e-4)
    report("full frame, forced fine depths: disp_map per pixel", forced["disp_map"], ref["disp_map"], atol=1e-4, rtol=1e-4)

    # (2) own importance samples
    own = hip_frame()
    assert torch.equal(own["rgb_map"], ret["rgb_map"].reshape(n, C).cpu()), "render_video must be these very launches"
    d_rgb = (own["rgb_map"] - ref["rgb_map"]).abs().max(1).values
    dz = (own["z_fine"] - ref["z_fine"]).abs().max(1).values
    beyond = torch.nonzero(d_rgb > 1e-4).reshape(-1)
    REPORT.append("full frame %dx%d, %d+%d samples, own importance samples: rgb max|d| %.2e, %d of %d pixels beyond 1e-4; merged depths differ "
                  "on %d pixels (median max|dz| %.1e, of the pixels beyond 1e-4: %.1e)"
                  % (Hh, Ww, S, S + Ni, float(d_rgb.max()), beyond.numel(), n, int((dz > 0).sum()), float(dz.median()),
                     float(dz[beyond].median()) if beyond.numel() else 0.0))
    if beyond.numel():
        sel = beyond[:8192]
        at_hip = _oracle_frame(pc, pf, pose.cpu(), n, Hh, Ww, Kt, C, S, Ni, draws, chunk, rows=sel, z_forced=(own["z"][sel], own["z_fine"][sel]))
        report("full frame, own samples: the %d pixels beyond 1e-4, oracle evaluated at the HIP path's depths" % sel.numel(),
               own["rgb_map"][sel], at_hip["rgb_map"], atol=1e-4)
    assert Kk.auto_fallback_max(DEV) == fallback_before      # no inference launch of this frame fell back to exact f32


def test_checkpoint_resume_equals_uninterrupted(tmp_path):
    """a reference-format .tar (train.py:443-455) written after 3 fused steps and loaded into a fresh
    graph + optimisers + TrainStep continues exactly like the uninterrupted run (Philox draws are a function of
    the step index, so the comparison is bit for bit); the file also loads through plain torch.optim objects the way
    test.py:98-107 does."""
    from benerf_amd import engine, workloads as WL, checkpoint as CK, kernels as K
    wl = dict(WL.WORKLOADS["C3"], S=16, Ni=16, Re=32, Rr=4, n=5)
    args = WL.make_args(wl, optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    HW = cam["H"] * cam["W"]
    rng = np.random.default_rng(5)
    accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
    img = torch.from_numpy(rng.random((HW, 3)).astype(np.float32)).to(DEV)
    ts_e, ts_r = torch.tensor([0.2, 0.3], device=DEV), torch.tensor([0.0, 1.0], device=DEV)

    def run(step, n0, n1):
        for it in range(n0, n1):
            idx_e = K.sample_pixels(HW, 32, 9, 2 * it, torch.device(DEV))
            idx_r = K.sample_pixels(HW, 4, 9, 2 * it + 1, torch.device(DEV))
            step.step(ts_e, ts_r, idx_e, idx_r, accu, img)

    model_a, g_a = _graph(args, seed=3)
    step_a = engine.TrainStep(g_a, args, cam_o, cam_o, torch.device(DEV), seed=21)
    run(step_a, 0, 5)

    model_b, g_b = _graph(args, seed=3)
    step_b = engine.TrainStep(g_b, args, cam_o, cam_o, torch.device(DEV), seed=21)
    run(step_b, 0, 3)
    optims_b = model_b.setup_optimizer(args)
    step_b.export_optimizer_state(optims_b)
    path = str(tmp_path / "000003.tar")
    CK.save(path, g_b, optims_b, step_b.global_step)
    ck = torch.load(path)
    assert sorted(ck) == sorted(["global_step", "graph"] + list(CK.OPTIMIZER_KEYS)) and ck["global_step"] == 3
    st0 = ck["optimizer_nerf"]["state"][0]
    assert sorted(st0) == ["exp_avg", "exp_avg_sq", "step"] and float(st0["step"]) == 3.0
    assert len(ck["optimizer_nerf"]["state"]) == 48 and len(ck["optimizer_pose"]["state"]) == 1

    model_c, g_c = _graph(args, seed=99)                      # different initial weights: everything must come from the file
    optims_c = model_c.setup_optimizer(args)
    step_c = engine.TrainStep(g_c, args, cam_o, cam_o, torch.device(DEV), seed=21)
    gs = CK.load(path, g_c, optims_c)
    step_c.import_optimizer_state(optims_c, gs)
    assert torch.equal(step_c.flat_p, step_b.flat_p) and torch.equal(step_c.flat_m, step_b.flat_m)
    run(step_c, 3, 5)
    assert torch.equal(step_c.flat_p, step_a.flat_p), "resumed run must equal the uninterrupted one bit for bit"
    assert torch.equal(step_c.flat_v, step_a.flat_v) and step_c.global_step == 5


def me_mode():
    from benerf_amd import kernels as K
    return K.get_mlp_precision()


def _dp_worker(rank, world, port, out_q, mode, base="C5", bins=1, uneven=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import test_api_gpu as me
    from benerf_amd import engine, kernels, workloads as WL
    kernels.set_mlp_precision(mode)
    torch.cuda.set_device(0)
    pg = None
    if world > 1:
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        pg = torch.distributed.group.WORLD
    # C5 = E2NeRF_Real: the globally normalised loss (blocking exchange of the sums of squares before the gradient); C4 =
    # E2NeRF_Synthetic: mean-squared losses (no exchange before the gradient, the loss VALUES are summed asynchronously)
    wl = dict(WL.WORKLOADS[base], S=16, Ni=16, Re=32, Rr=4, n=5)
    args = WL.make_args(wl, optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    _, g = me._graph(args, seed=11 + 100 * rank)      # replicas initialise DIFFERENTLY: TrainStep broadcasts rank 0's parameters
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV), world_size=world, rank=rank, process_group=pg, event_bins=bins,
                            uneven_shards=uneven)
    rng = np.random.default_rng(2)
    HW = cam["H"] * cam["W"]
    n_e, n_r = (31, 5) if uneven else (32, 4)      # uneven: a global batch the ranks cannot split evenly (low ranks take one more)
    idx_e = torch.from_numpy(rng.permutation(HW)[:n_e]).to(DEV)
    idx_r = torch.from_numpy(rng.permutation(HW)[:n_r]).to(DEV)
    accu = torch.from_numpy(rng.integers(-3, 4, HW if bins == 1 else (bins, HW)).astype(np.float32)).to(DEV)
    img = torch.from_numpy(rng.random((HW, 3)).astype(np.float32)).to(DEV)
    # explicit draws for the GLOBAL batch; each rank takes the rows of its pixels (pose-major)
    P, S, Ni = 5, 16, 16
    Pe = bins + 1       # dense event bins: the event batch is rendered at the bins + 1 boundaries
    d_e, d_r = GI.render_draws(rng, Pe * n_e, S, Ni), GI.render_draws(rng, P * n_r, S, Ni)

    def shard(d, n_poses, n_pix, which):
        from benerf_amd import dist
        if uneven and world > 1:     # TrainStep(uneven_shards=True): the ray-balanced table (event pixels compensate the blur pixels' left-overs)
            lo, hi = dist.balanced_shard_bounds(n_e, n_r, Pe, P, world)[rank][which]
        else:
            lo, hi = dist.shard_bounds(n_pix, rank, world, uneven)
        sel = torch.cat([torch.arange(p * n_pix + lo, p * n_pix + hi) for p in range(n_poses)])
        return engine.Draws(*(d[k][sel].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    losses = step.step(torch.tensor([0.2, 0.45], device=DEV), torch.tensor([0.0, 1.0], device=DEV), idx_e, idx_r, accu, img,
                       shard(d_e, Pe, n_e, 0), shard(d_r, P, n_r, 1))
    torch.cuda.synchronize()
    if world > 1 and not uneven:   # a global batch the ranks cannot split evenly is refused (unless uneven_shards), not silently truncated
        with pytest.raises(ValueError):
            step.step(torch.tensor([0.2, 0.45], device=DEV), torch.tensor([0.0, 1.0], device=DEV), idx_e[:31], idx_r, accu, img)
    result = (losses.cpu().numpy(), step.flat_g.cpu().numpy(), step.flat_p.cpu().numpy())
    if world > 1 and mode == "split" and base == "C5":
        # range guard across ranks: ONE rank leaves the f16 range (its words are poisoned here; its gradients would be inf /
        # NaN and reach everybody through the sum) -> the verdict rides the trajectory bucket and EVERY replica skips the step
        from benerf_amd import _lib
        p_before, m_before = step.flat_p.clone(), step.flat_m.clone()
        if rank == world - 1:
            step.guard.words[_lib.ST_ACT] = 0x7f800000
        step.step(torch.tensor([0.2, 0.45], device=DEV), torch.tensor([0.0, 1.0], device=DEV), idx_e, idx_r, accu, img,
                  shard(d_e, Pe, n_e, 0), shard(d_r, P, n_r, 1))
        torch.cuda.synchronize()
        assert torch.equal(step.flat_p, p_before) and torch.equal(step.flat_m, m_before), "rank %d did not skip the step" % rank
        w = step.guard.words.cpu().tolist()
        assert w[_lib.ST_SKIPPED] == 1 and w[_lib.ST_SKIP] == 1 and w[_lib.ST_ACT] == 0
    if rank == 0:
        out_q.put(result)
    if world > 1:
        torch.distributed.destroy_process_group()


def test_sharded_step_equals_single_rank():
    """Real HIP path, all ranks on cuda:0, gloo transport, world sizes 2 and 4: loss, all-reduced gradients and updated
    parameters of the sharded step equal the single-rank step on the same global batch; replicas start from rank 0's
    parameters whatever their own initialisation; uneven global batches are rejected."""
    ctx = mp.get_context("spawn")
    # third leg: dense event bins (3 bins = 4 event poses, one stacked exchange of all bins' sums for the normalised loss)
    # fourth leg: uneven shards (31 event / 5 blur pixels over 2 and 4 ranks, TrainStep(uneven_shards=True): nothing dropped)
    for base, worlds, bins, uneven in (("C5", (1, 2, 4), 1, False), ("C4", (1, 2), 1, False), ("C5", (1, 2), 3, False), ("C5", (1, 2, 4), 1, True)):
        res = {}
        for world in worlds:
            q = ctx.Queue()
            port = 29650 + world + (os.getpid() % 100) + (10 if base == "C4" else 0) + (20 if bins > 1 else 0) + (30 if uneven else 0)
            procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, me_mode(), base, bins, uneven)) for r in range(world)]
            for p in procs:
                p.start()
            res[world] = q.get(timeout=300)
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
        l1, g1, p1 = res[1]
        for world in worlds[1:]:
            l2, g2, p2 = res[world]
            base = base.split("/")[0] + ("" if bins == 1 else "/%d bins" % bins) + ("/uneven" if uneven else "")
            report("DP %s loss (%d ranks vs 1)" % (base, world), l2, l1, atol=1e-6, rtol=1e-5)
            # both modes: only the summation order differs (the split mode's per-rank gradient scales are powers of two)
            report("DP %s flat gradient (%d ranks vs 1)" % (base, world), g2, g1, atol=2e-6 * float(np.abs(g1).max()), rtol=1e-4)
            # the first Adam step is lr * g / (|g| + eps): entries with |g| ~ eps amplify the 1e-7 gradient wobble,
            # bounded by a few percent of lr = 5e-4
            report("DP %s parameters after Adam (%d ranks vs 1)" % (base, world), p2, p1, atol=2e-5, rtol=1e-5)


def _rccl_worker(port, out_q, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import test_api_gpu as me
    from benerf_amd import dist, engine, kernels, workloads as WL
    kernels.set_mlp_precision(mode)
    torch.cuda.set_device(0)
    dev = torch.device(DEV)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # RCCL, a communicator of one
    pg = torch.distributed.group.WORLD
    assert not dist._through_host(torch.zeros(1, device=dev), pg), "nccl must reduce device buffers in place"
    res = {}
    # C5: the normalised loss (the blocking 16-double exchange too); C4: mean-squared losses (asynchronous sum of the loss values)
    for base in ("C5", "C4"):
        wl = dict(WL.WORKLOADS[base], S=32, Ni=32, Re=64, Rr=8, n=5)
        args = WL.make_args(wl, optimize_trans=True)
        cam = WL.CAMERAS[wl["cam"]]
        cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        rng = np.random.default_rng(5)
        HW = cam["H"] * cam["W"]
        idx_e = torch.from_numpy(rng.permutation(HW)[:64]).to(DEV)
        idx_r = torch.from_numpy(rng.permutation(HW)[:8]).to(DEV)
        accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
        img = torch.from_numpy(rng.random((HW, 3)).astype(np.float32)).to(DEV)
        for communicate in (False, True):
            dist.ALWAYS_COMMUNICATE = communicate
            _, g = me._graph(args, seed=21)
            step = engine.TrainStep(g, args, cam_o, cam_o, dev, world_size=1, rank=0, process_group=pg, seed=3)
            for k in range(3):
                losses = step.step(torch.tensor([0.2, 0.45], device=DEV), torch.tensor([0.0, 1.0], device=DEV), idx_e, idx_r, accu, img)
            step.check_range()
            res[(base, communicate)] = (losses.cpu().numpy(), step.flat_g.cpu().numpy(), step.flat_p.cpu().numpy())
    # the bench's collective self-check on the same communicator
    sys.path.insert(0, ROOT)
    import bench
    dist.ALWAYS_COMMUNICATE = False
    comm = bench.collective_selfcheck(1, dev, 595586)
    out_q.put((res, comm))
    torch.distributed.destroy_process_group()


def test_step_over_rccl_communicator_of_one():
    """The device branch of the gradient exchange on real RCCL: a one-rank communicator (all a single-GPU box offers), every
    collective of the step issued (dist.ALWAYS_COMMUNICATE) - three asynchronous in-place all-reduces of slices of the flat
    gradient buffer on the communicator's stream, waited for on the side / main streams, the 16-double statistics exchange and
    the start-up broadcast.  A sum over one rank is the identity, so three steps must reproduce the plain run BIT FOR BIT: any
    missing stream dependency between the kernels and the communicator would show as a changed gradient or parameter."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29720 + (os.getpid() % 100), q, me_mode()))
    p.start()
    res, comm = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    for base in ("C5", "C4"):
        for a, b, what in zip(res[(base, False)], res[(base, True)], ("losses", "flat gradient", "parameters after 3 steps")):
            assert np.array_equal(a, b), "%s: %s changed when the collectives ran over RCCL" % (base, what)
    assert comm["rccl_ranks_seen"] == 1 and comm["allreduce_ms"] > 0
    report("RCCL one-rank bucketed all-reduce [ms]", np.array([comm["allreduce_ms"]]), np.array([comm["allreduce_ms"]]), atol=1, rtol=0)


def test_render_after_fused_steps_uses_current_weights():
    """Graph.render / render_video between TrainStep.step calls must see the UPDATED weights: the fused Adam kernel
    rewrites parameter storage without bumping torch's version counters, so the module-level packed copies are
    invalidated through kernels.params_changed()."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=16, Rr=2, n=5)
    args = WL.make_args(wl, benerf_raw_noise_std=0.0, chunk=2048)
    cam = WL.CAMERAS["unreal"]
    model, g = _graph(args)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    Kmat = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    rng = np.random.default_rng(17)
    HW = cam["H"] * cam["W"]
    accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
    img = torch.from_numpy(rng.random((HW, 1)).astype(np.float32)).to(DEV)
    pose = torch.eye(3, 4, device=DEV)[None]
    idx = torch.arange(0, 512, device=DEV)
    outs = []
    for it in range(3):
        torch.manual_seed(7)     # same draws every time: only the weights change
        with torch.no_grad():
            ret = g.render(it, pose, idx, cam["H"], cam["W"], torch.Tensor(Kmat), args, False, None, torch.tensor([]))
        torch.manual_seed(7)
        N = idx.shape[0]
        d = engine.Draws(torch.rand((N, 16), device=DEV), None, torch.rand([N, 16], device=DEV), None, noise_std=0.0)
        ref, _ = engine._render_forward(cam_o, True, 16, 16, d, pose, idx, step.net_c.packed, step.net_f.packed, False)
        assert torch.equal(ret["rgb_map"], ref["rgb_map"]), "render() used stale packed weights at iteration %d" % it
        outs.append(ret["rgb_map"].clone())
        step.step(torch.tensor([0.2, 0.3], device=DEV), torch.tensor([0.0, 1.0], device=DEV),
                  torch.from_numpy(rng.permutation(HW)[:16]).to(DEV), torch.from_numpy(rng.permutation(HW)[:2]).to(DEV), accu, img)
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2]), "training must change the render"


def test_f16_range_guard():
    """Hidden weights x300 drive activations past f16's maximum.  Split mode: the training forward reports it through
    the status words (BenerfRangeError from check_mlp_status, never a silent inf); inference launches run as
    BENERF_MLP_AUTO, return the exact-f32 result and keep their words to themselves (an overflow handled during e.g.
    render_image_test must not gate training); the fused Adam step leaves the parameters untouched."""
    from benerf_amd import _lib, kernels as K
    if K.get_mlp_precision() != "split":
        pytest.skip("range guard concerns the split-f16 mode")
    rng = np.random.default_rng(5)
    C = 1
    p = O.xavier_params(rng, C)
    for k in p:
        if k.startswith("pts_linears") and k.endswith("weight"):
            p[k] = p[k] * (300.0 if k.startswith("pts_linears.0") else 3.0)
    net = K.PackedMlp([p[n + ".weight"].to(DEV) for n in K.LAYER_NAMES], [p[n + ".bias"].to(DEV) for n in K.LAYER_NAMES], C)
    net.pack()
    N, S = 64, 32
    ro = GI.f32(rng.uniform(-0.5, 0.5, (N, 3))).to(DEV)
    rd = GI.f32(rng.uniform(-1, 1, (N, 3))).to(DEV)
    vd = torch.nn.functional.normalize(GI.f32(rng.standard_normal((N, 3))), dim=-1).to(DEV)
    z = GI.f32(np.sort(rng.random((N, S)), -1)).to(DEV)
    K.range_guard(DEV).words.zero_()                              # clean slate
    raw_f32, _ = K.mlp_fwd(net, ro, rd, vd, z, False, precision="f32")
    assert torch.isfinite(raw_f32).all() and float(raw_f32.abs().max()) > 0
    # inference (AUTO): valid output although the split launch overflowed; the device's training words stay clean
    raw_auto, _ = K.mlp_fwd(net, ro, rd, vd, z, False, precision="split")
    assert torch.equal(raw_auto, raw_f32), "BENERF_MLP_AUTO must fall back to the exact-f32 kernels"
    assert K.auto_fallback_max(DEV) >= 65504.0
    K.check_mlp_status(torch.device(DEV))
    prm = torch.ones(1000, device=DEV)
    before = prm.clone()
    K.adam_step(prm, torch.ones_like(prm), torch.zeros_like(prm), torch.zeros_like(prm), 1e-3, 1)
    assert not torch.equal(prm, before), "an overflow that AUTO handled must not gate an optimiser step"
    # training forward: no silent inf - Adam skips, the status check raises
    raw_s, acts = K.mlp_fwd(net, ro, rd, vd, z, True, precision="split")
    before = prm.clone()
    K.adam_step(prm, torch.ones_like(prm), torch.zeros_like(prm), torch.zeros_like(prm), 1e-3, 1)
    assert torch.equal(prm, before), "Adam must skip a step whose status shows a range violation"
    g = K.range_guard(DEV)
    g.post()
    torch.cuda.synchronize()
    with pytest.raises(_lib.BenerfRangeError):
        g.poll()                                                  # the non-blocking look at the mirrored words sees it too
    with pytest.raises(_lib.BenerfRangeError):
        K.check_mlp_status(torch.device(DEV))
    K.adam_step(prm, torch.ones_like(prm), torch.zeros_like(prm), torch.zeros_like(prm), 1e-3, 1)
    assert not torch.equal(prm, before)
    # well-scaled weights: nothing reported
    p2 = O.xavier_params(rng, C)
    net2 = K.PackedMlp([p2[n + ".weight"].to(DEV) for n in K.LAYER_NAMES], [p2[n + ".bias"].to(DEV) for n in K.LAYER_NAMES], C)
    net2.pack()
    K.mlp_fwd(net2, ro, rd, vd, z, True, precision="split")
    K.check_mlp_status(torch.device(DEV))


def test_train_step_range_guard_skips_counts_and_raises():
    """TrainStep owns its guard words: a step that leaves the f16 range is skipped as a whole (networks, trajectory, Adam
    moments untouched), counted, and the next step starts clean; after MAX_SKIPPED_IN_A_ROW skipped steps in a row step()
    raises from the mirrored counters without a synchronisation of its own; check_range() raises at once.  Inference on the
    same device in between (AUTO fallback) does not disturb it."""
    from benerf_amd import _lib, engine, kernels as K, workloads as WL
    if K.get_mlp_precision() != "split":
        pytest.skip("range guard concerns the split-f16 mode")
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=16, Rr=2, n=5)
    args = WL.make_args(wl, optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    model, g = _graph(args, seed=9)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    step.GUARD_POST_EVERY = 1
    rng = np.random.default_rng(3)
    HW = cam["H"] * cam["W"]
    accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
    img = torch.from_numpy(rng.random((HW, 1)).astype(np.float32)).to(DEV)

    def one():
        return step.step(torch.tensor([0.2, 0.3], device=DEV), torch.tensor([0.0, 1.0], device=DEV),
                         torch.from_numpy(rng.permutation(HW)[:16]).to(DEV), torch.from_numpy(rng.permutation(HW)[:2]).to(DEV), accu, img)

    one()
    step.check_range()                                            # a healthy step: nothing to report
    good = step.flat_p.clone()
    with torch.no_grad():      # fine network: hidden weights x3, first layer x300 -> finite activations far past 65504 (as test_f16_range_guard)
        for li in range(8):
            step.net_f.views_w[li].mul_(300.0 if li == 0 else 3.0)
    step.net_f.packed.pack()
    poisoned = step.flat_p.clone()
    m_before, v_before = step.flat_m.clone(), step.flat_v.clone()
    one()
    torch.cuda.synchronize()
    assert torch.equal(step.flat_p, poisoned) and torch.equal(step.flat_m, m_before) and torch.equal(step.flat_v, v_before), \
        "a step outside the f16 range must leave parameters and Adam moments untouched"
    words = step.guard.words.cpu().tolist()
    assert words[_lib.ST_SKIPPED] == 1 and words[_lib.ST_CONSECUTIVE] == 1 and words[_lib.ST_STEPS] == 2 and words[_lib.ST_ACT] == 0
    one()
    torch.cuda.synchronize()
    with pytest.raises(_lib.BenerfRangeError):
        for _ in range(4):                                        # the third skipped step in a row trips the poll of a later step
            one()
            torch.cuda.synchronize()
    with pytest.raises(_lib.BenerfRangeError):
        step.check_range()
    # repaired parameters: training resumes, the consecutive counter falls back to zero
    with torch.no_grad():
        step.flat_p.copy_(good)
    K.params_changed()             # storage rewritten behind the Parameters' version counters: say so BEFORE re-packing (the weight-
    step.net_c.packed.pack()       # gradient launch refuses a network whose parameters changed since its last pack)
    step.net_f.packed.pack()
    step.guard.words.zero_()
    one()
    torch.cuda.synchronize()
    step.check_range()
    assert not torch.equal(step.flat_p, good)


def test_train_step_overlap_slot():
    """TrainStep.step(overlap=f): f runs exactly once per step, on the step's main stream, and work queued in it (here: the
    NEXT step's event image and pixel draws, written into the buffers the current step has already read) changes neither this
    step's result nor the next one's - two steps with the slot equal two steps with everything prepared up front, bit for bit."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=32, Rr=4, n=5)
    args = WL.make_args(wl, optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    HW = cam["H"] * cam["W"]
    rng = np.random.default_rng(31)
    accus = [torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV) for _ in range(2)]
    idxs = [(torch.from_numpy(rng.permutation(HW)[:32]).to(DEV), torch.from_numpy(rng.permutation(HW)[:4]).to(DEV)) for _ in range(2)]
    img = torch.from_numpy(rng.random((HW, 1)).astype(np.float32)).to(DEV)
    ets, rts = torch.tensor([0.2, 0.3], device=DEV), torch.tensor([0.0, 1.0], device=DEV)
    res = []
    for use_slot in (False, True, "prefetch"):
        _, g = _graph(args, seed=23)
        step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV), seed=5)
        calls = []
        if use_slot == "prefetch":
            # the slot returns the next step's inputs: TrainStep sets up their poses / rays / depths behind its own trajectory
            # update; used by the next call because it is handed the same tensor objects, ignored by the one after
            nxt = (ets.clone(), rts, idxs[1][0], idxs[1][1])
            l0 = step.step(ets, rts, idxs[0][0], idxs[0][1], accus[0], img, overlap=lambda: nxt)
            assert step._prefetched is not None
            l1 = step.step(nxt[0], nxt[1], nxt[2], nxt[3], accus[1], img)
            assert step._prefetched is None
            p_two = step.flat_p.clone()
            # a set-up made stale by a torch-side write to the trajectory parameters is not used: same result as a plain step
            stale = (ets.clone(), rts, idxs[0][0], idxs[0][1])
            step.step(ets, rts, idxs[1][0], idxs[1][1], accus[1], img, overlap=lambda: stale)
            ref_step = engine.TrainStep(_graph(args, seed=23)[1], args, cam_o, cam_o, torch.device(DEV), seed=5)
            with torch.no_grad():
                for buf in ("flat_p", "flat_m", "flat_v"):
                    getattr(ref_step, buf).copy_(getattr(step, buf))
                ref_step.global_step = step.global_step
                step.knots.mul_(1.5)
                ref_step.knots.mul_(1.5)
            ref_step.net_c.packed.pack()
            ref_step.net_f.packed.pack()
            a_ = step.step(stale[0], stale[1], stale[2], stale[3], accus[0], img)
            b_ = ref_step.step(ets, rts, idxs[0][0], idxs[0][1], accus[0], img)
            assert torch.equal(a_, b_) and torch.equal(step.flat_p, ref_step.flat_p)
            # ... and so is one made stale by a write THROUGH THE nn.Parameters (pose re-initialisation, load_state_dict): their
            # `.data` aliases the flat buffer, but set_data does not share version counters - the snapshot holds theirs too
            stale2 = (ets.clone(), rts, idxs[1][0], idxs[1][1])
            step.step(ets, rts, idxs[0][0], idxs[0][1], accus[0], img, overlap=lambda: stale2)
            assert step._prefetched is not None
            ref2 = engine.TrainStep(_graph(args, seed=23)[1], args, cam_o, cam_o, torch.device(DEV), seed=5)
            with torch.no_grad():
                for buf in ("flat_p", "flat_m", "flat_v"):
                    getattr(ref2, buf).copy_(getattr(step, buf))
                ref2.global_step = step.global_step
                g.evt_knot_pose_se3.params.weight.mul_(0.5)
                g.transform.params.weight.add_(0.01)
                ref2.knots.mul_(0.5)
                ref2.transform.add_(0.01)
            assert torch.equal(step.knots, ref2.knots) and torch.equal(step.transform, ref2.transform), "the Parameters alias the flat buffer"
            ref2.net_c.packed.pack()
            ref2.net_f.packed.pack()
            a_ = step.step(stale2[0], stale2[1], stale2[2], stale2[3], accus[1], img)
            b_ = ref2.step(ets, rts, idxs[1][0], idxs[1][1], accus[1], img)
            assert torch.equal(a_, b_) and torch.equal(step.flat_p, ref2.flat_p)
        elif not use_slot:
            l0 = step.step(ets, rts, idxs[0][0], idxs[0][1], accus[0], img)
            l1 = step.step(ets, rts, idxs[1][0], idxs[1][1], accus[1], img)
        else:
            accu = accus[0].clone()
            ie, ir = idxs[0][0].clone(), idxs[0][1].clone()

            def prepare_next():
                calls.append(torch.cuda.current_stream().cuda_stream)
                accu.copy_(accus[1])          # same buffers the running step was given: its reads are queued ahead
                ie.copy_(idxs[1][0])
                ir.copy_(idxs[1][1])
            main = torch.cuda.current_stream().cuda_stream
            l0 = step.step(ets, rts, ie, ir, accu, img, overlap=prepare_next)
            assert calls == [main]
            l1 = step.step(ets, rts, ie, ir, accu, img)
        step.check_range()
        res.append((l0.cpu().numpy(), l1.cpu().numpy(), (p_two if use_slot == "prefetch" else step.flat_p).cpu().numpy()))
    for other in (1, 2):
        for a, b, what in zip(res[0], res[other], ("first step's losses", "second step's losses", "parameters after two steps")):
            assert np.array_equal(a, b), what


def test_train_step_skips_a_nonfinite_loss_gradient():
    """Both arithmetic modes: a NaN parameter (here the fine network's colour bias; a NaN density is clamped away by the
    compositing kernel's relu, fmaxf(NaN, 0) = 0) makes raw, the loss and d_raw NaN - the
    compositing backward records NaN as +inf in the step's max |d_raw| word, the step gate turns that into the skip verdict:
    every other parameter and the Adam moments stay as they were, the step is counted, and training resumes once the
    parameter is repaired."""
    from benerf_amd import _lib, engine, kernels as K, workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=16, Rr=2, n=5)
    args = WL.make_args(wl, optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    model, g = _graph(args, seed=19)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    rng = np.random.default_rng(13)
    HW = cam["H"] * cam["W"]
    accu = torch.from_numpy(rng.integers(-3, 4, HW).astype(np.float32)).to(DEV)
    img = torch.from_numpy(rng.random((HW, 1)).astype(np.float32)).to(DEV)

    def one():
        return step.step(torch.tensor([0.2, 0.3], device=DEV), torch.tensor([0.0, 1.0], device=DEV),
                         torch.from_numpy(rng.permutation(HW)[:16]).to(DEV), torch.from_numpy(rng.permutation(HW)[:2]).to(DEV), accu, img)

    one()
    step.check_range()
    bias = step.net_f.views_b[_lib.L_RGB]
    keep = bias.clone()
    with torch.no_grad():
        bias.fill_(float("nan"))
    step.net_f.packed.pack()
    before, m_before = step.flat_p.clone(), step.flat_m.clone()
    losses = one()
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(losses[0]))
    same = torch.isnan(before) | (step.flat_p == before)
    assert bool(same.all()) and torch.equal(step.flat_m, m_before), "a step with a NaN loss gradient must not touch parameters or moments"
    words = step.guard.words.cpu().tolist()
    assert words[_lib.ST_SKIPPED] == 1 and words[_lib.ST_SKIP] == 1 and (words[_lib.ST_LAST_GRAD] & 0xffffffff) == 0x7f800000
    assert words[_lib.ST_STEP_SCRATCH] == 0 and words[_lib.ST_STEP_SCRATCH + 1] == 0      # cleared for the next step
    with pytest.raises(_lib.BenerfRangeError):
        step.check_range()
    with torch.no_grad():
        bias.copy_(keep)
    K.params_changed()
    step.net_f.packed.pack()
    one()
    torch.cuda.synchronize()
    step.check_range()
    assert bool(torch.isfinite(step.flat_p).all()) and not torch.equal(step.flat_p, before)


@pytest.mark.parametrize("which", ["rgb", "event", "both"])
def test_fused_step_with_crf(which):
    """optimize_rgb_crf / optimize_event_crf (train.py:180-192) in the fused TrainStep: losses, tone-mapper gradients and the
    gradients that flow back through the mappers into the networks and the trajectory, against the oracle's autograd."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=24, Rr=3, n=5)
    args = WL.make_args(wl, optimize_rgb_crf=which in ("rgb", "both"), optimize_event_crf=which in ("event", "both"), optimize_trans=True)
    cam = WL.CAMERAS[wl["cam"]]
    C, S, Ni, P, Re, Rr = 1, 16, 16, 5, 24, 3
    model, g = _graph(args, seed=5)
    rng = np.random.default_rng(77)
    with torch.no_grad():
        for m in (g.rgb_crf, g.event_crf):
            for prm in m.parameters():
                prm.add_(torch.from_numpy(rng.uniform(-0.05, 0.05, tuple(prm.shape)).astype(np.float32)).to(DEV))
    def crf_params(mod):
        return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    o_rgb = crf_params(g.rgb_crf.mlp_gray) if args.optimize_rgb_crf else None
    o_evt = crf_params(g.event_crf.mlp_luminance) if args.optimize_event_crf else None
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.nerf.state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.nerf_fine.state_dict().items()}
    ko = g.evt_knot_pose_se3.params.weight.detach().cpu().clone().requires_grad_(True)
    tro = g.transform.params.weight.detach().cpu().clone().requires_grad_(True)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    HW = cam["H"] * cam["W"]
    idx_e, idx_r = GI.pixel_indices(rng, cam, Re), GI.pixel_indices(rng, cam, Rr)
    d_e, d_r = GI.render_draws(rng, 2 * Re, S, Ni), GI.render_draws(rng, P * Rr, S, Ni)
    acc = torch.from_numpy(rng.integers(-3, 4, (HW,)).astype(np.float32))
    img = torch.from_numpy(rng.random((HW, C)).astype(np.float32))
    evt_ts, rgb_ts = torch.tensor([0.31, 0.41]), torch.tensor([0.0, 1.0])
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C, n_samples=S,
                       n_importance=Ni, n_poses=P, dataset=wl["dataset"], threshold=wl["threshold"])
    loss_o, _ = O.step_loss(cfg, pc, pf, ko, tro, evt_ts, rgb_ts, idx_e, idx_r, acc.double().reshape(-1, 1)[idx_e], img[idx_r], d_e, d_r,
                            exact_pdf=True, event_crf=o_evt, rgb_crf=o_rgb)
    loss_o.backward()

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
    before = {id(p_): p_.detach().clone() for p_ in step.crf_params}
    losses = step.step(evt_ts.to(DEV), rgb_ts.to(DEV), idx_e.to(DEV), idx_r.to(DEV), acc.to(DEV), img.to(DEV), dd(d_e), dd(d_r))
    report("crf(%s) loss" % which, losses[0:1], loss_o.detach().float().reshape(1), atol=1e-6, rtol=2e-5)
    for name, mod, op in (("rgb", g.rgb_crf.mlp_gray, o_rgb), ("event", g.event_crf.mlp_luminance, o_evt)):
        if op is None:
            continue
        for k, v in mod.named_parameters():
            r = op[k].grad
            report("crf(%s) d %s_crf.%s" % (which, name, k), v.grad, r, atol=2e-5 * float(r.abs().max()) + 1e-9, rtol=1e-3)
            assert not torch.equal(v.detach(), before[id(v)]), "the tone-mapper must have taken its Adam step"
    sc = float(ko.grad.abs().max())
    report("crf(%s) d knots" % which, step.g_knots, ko.grad, atol=2e-3 * sc, rtol=2e-3)
    rw = pc["pts_linears.7.weight"].grad
    report("crf(%s) d nerf.pts_linears.7.weight" % which, step.net_c.gviews_w[7], rw, atol=2e-3 * float(rw.abs().max()), rtol=2e-3)
    with pytest.raises(NotImplementedError):
        engine.TrainStep(_graph(WL.make_args(dict(wl, channels=3), optimize_rgb_crf=True), seed=1)[1],
                         WL.make_args(dict(wl, channels=3), optimize_rgb_crf=True), cam_o, cam_o, torch.device(DEV))


def test_barf_c2f_through_render_and_fused_step():
    """use_barf_c2f on the two paths that carry iter_step: Graph.render (RenderRays) and the fused TrainStep (its
    global_step), against the oracle with the same coarse-to-fine weights (model/nerf.py:16-26,78-89)."""
    from benerf_amd import engine, kernels as K, workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=24, Rr=3, n=5)
    args = WL.make_args(wl, use_barf_c2f=True, barf_c2f_start=0.1, barf_c2f_end=0.5, max_iter=100)
    cam = WL.CAMERAS[wl["cam"]]
    C, S, Ni, P, Re, Rr = 1, 16, 16, 5, 24, 3
    model, g = _graph(args, seed=6)
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.nerf.state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.nerf_fine.state_dict().items()}
    ko = g.evt_knot_pose_se3.params.weight.detach().cpu().clone().requires_grad_(True)
    rng = np.random.default_rng(78)
    Kmat = GI.cam_K(cam)
    idx_e, idx_r = GI.pixel_indices(rng, cam, Re), GI.pixel_indices(rng, cam, Rr)
    d_e, d_r = GI.render_draws(rng, 2 * Re, S, Ni), GI.render_draws(rng, P * Rr, S, Ni)
    it = 27                                                    # inside the window: progress 0.27
    barf = (it, 100, 0.1, 0.5)
    # Graph.render at iter_step = it, explicit draws through the torch generator replaced by Philox-free Draws: use the engine
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    poses = O.trajectory_poses(ko.detach(), None, (0.31, 0.41), 2, "spline")
    ref = O.render(pc, pf, poses, idx_e, cam["H"], cam["W"], Kmat, C, S, Ni, d_e, exact_pdf=True, barf=barf)
    net_c, net_f = g.nerf.packed(), g.nerf_fine.packed()
    net_c.pack_if_stale()
    net_f.pack_if_stale()
    net_c.pe_weights = net_f.pe_weights = K.barf_pe_weights(it, 100, 0.1, 0.5, torch.device(DEV))
    dd = engine.Draws(*(d_e[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
    out, _ = engine._render_forward(cam_o, True, S, Ni, dd, poses.to(DEV).contiguous(), idx_e.to(DEV), net_c, net_f, False)
    report("barf render rgb_map", out["rgb_map"], ref["rgb_map"], atol=1e-4)
    report("barf render rgb0", out["rgb0"], ref["rgb0"], atol=1e-4)
    # fused step at global_step = it
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    step.global_step = it
    HW = cam["H"] * cam["W"]
    acc = torch.from_numpy(rng.integers(-3, 4, (HW,)).astype(np.float32))
    img = torch.from_numpy(rng.random((HW, C)).astype(np.float32))
    evt_ts, rgb_ts = torch.tensor([0.31, 0.41]), torch.tensor([0.0, 1.0])
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C, n_samples=S,
                       n_importance=Ni, n_poses=P, dataset=wl["dataset"], threshold=wl["threshold"])
    loss_o, _ = O.step_loss(cfg, pc, pf, ko, torch.zeros(1, 6), evt_ts, rgb_ts, idx_e, idx_r, acc.double().reshape(-1, 1)[idx_e], img[idx_r],
                            d_e, d_r, exact_pdf=True, barf=barf)
    loss_o.backward()

    def dr(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
    losses = step.step(evt_ts.to(DEV), rgb_ts.to(DEV), idx_e.to(DEV), idx_r.to(DEV), acc.to(DEV), img.to(DEV), dr(d_e), dr(d_r))
    report("barf fused step loss", losses[0:1], loss_o.detach().float().reshape(1), atol=1e-6, rtol=2e-5)
    report("barf fused step d knots", step.g_knots, ko.grad, atol=2e-3 * float(ko.grad.abs().max()), rtol=2e-3)
    r0 = pc["pts_linears.0.weight"].grad
    report("barf fused step d nerf.pts_linears.0.weight", step.net_c.gviews_w[0], r0, atol=2e-3 * float(r0.abs().max()), rtol=2e-3)


def test_bench_two_ranks_end_to_end():
    """The multi-rank training path of bench.py, end to end, on the one-GPU box: `python bench.py --gpus 2 --oversubscribe` spawns
    its two ranks under torch.distributed.run (both on device 0, gloo transport - the same launch, sharding, bucketed exchange,
    guard verdict and Adam code an 8-GPU run takes; RCCL itself is exercised by test_step_over_rccl_communicator_of_one).  C4 is
    strong-scaled: the two ranks split ONE 8181-ray batch, so after the same number of steps from the same
    initialisation the loss agrees with the one-rank run of that batch (the ranks' Philox jitter streams differ from the
    one-rank run's: agreement to a percent, not bit for bit - test_sharded_step_equals_single_rank is the exact test)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}

    def run(extra):
        res = subprocess.run([sys.executable, bench, "--steps", "3", "--warmup", "1", "--workload", "C4", "--no-cpu-baseline"] + extra,
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])

    two = run(["--gpus", "2", "--oversubscribe"])
    assert two["n_gpus"] == 2 and two["rccl_ranks_seen"] == 2 and two["scaling"] == "strong" and two["allreduce_ms"] > 0
    # 215 blur pixels do not split over two ranks: rank 0 renders 108, rank 1 107 - nothing is dropped - and the 2048 event pixels are
    # dealt 1019 / 1029 so that both ranks render the same number of rays +- 1 (round 5: dist.balanced_shard_bounds)
    assert two["config"]["parallelism"] == "dp2" and two["config"]["rays_per_step_per_gpu"] == 2 * 1019 + 19 * 108
    assert two["config"]["rays_global"] == 2 * 2048 + 19 * 215 == 8181
    assert two["per_rank"]["rays_per_step_by_rank"] == [2 * 1019 + 19 * 108, 2 * 1029 + 19 * 107]
    assert two["per_rank"]["ms_per_step_max"] >= two["per_rank"]["ms_per_step_min"] > 0
    assert set(two["bucket_wait"]) >= {"fine_net", "coarse_net", "trajectory"}
    assert abs(two["value"] - 8181 / (two["ms_per_step"] * 1e-3)) / two["value"] < 2e-3
    assert np.isfinite(two["config"]["final_loss"]) and two["value"] > 0
    one = run(["--gpus", "1", "--primary-only"])
    assert one["n_gpus"] == 1 and np.isfinite(one["config"]["final_loss"])
    assert abs(two["config"]["final_loss"] - one["config"]["final_loss"]) <= 2e-2 * abs(one["config"]["final_loss"]), \
        (two["config"]["final_loss"], one["config"]["final_loss"])


def test_no_grad_renders_run_inference_launches(monkeypatch):
    """torch.no_grad() renders (render_video, render_*_test: model/nerf.py:353, run_nerf_helpers.py:117-171) must run INFERENCE
    launches - no saved activations, BENERF_MLP_AUTO with its exact-f32 fallback in the split mode.  An autograd Function's
    ctx.needs_input_grad says True for parameters under no_grad too: round 6 found render_video saving 7 KB per sample point for
    nobody (25 % slower).  Values must not depend on the launch kind."""
    from benerf_amd import kernels as K, workloads as WL
    args = WL.make_args("C2", N_samples=16, N_importance=16, chunk=64)
    model, g = _graph(args)
    Kt = torch.tensor([[30.0, 0, 10.0], [0, 30.0, 6.0], [0, 0, 1]])
    pose = g.get_pose_rgb(args, [0, 1], seg_num=3).detach()[1:2]
    seen = []
    orig = K.mlp_fwd

    def spy(net, ro, rd, vd, z, save_acts, *a, **k):
        seen.append(bool(save_acts))
        return orig(net, ro, rd, vd, z, save_acts, *a, **k)
    monkeypatch.setattr(K, "mlp_fwd", spy)
    idx = torch.arange(64, device=DEV)
    torch.manual_seed(3)
    with torch.no_grad():
        a = g.render(0, pose, idx, 12, 20, Kt, args, enable_crf=False, sensor_type=None, remap=np.array([]))
    assert seen == [False, False], seen
    seen.clear()
    torch.manual_seed(3)
    b = g.render(0, pose, idx, 12, 20, Kt, args, enable_crf=False, sensor_type=None, remap=np.array([]))
    assert seen == [True, True] and b["rgb_map"].requires_grad and not a["rgb_map"].requires_grad
    report("render under no_grad (inference launch) vs with grad (training launch)", a["rgb_map"], b["rgb_map"].detach(), atol=1e-6)
    seen.clear()
    g.render_video(0, pose, 12, 20, Kt, args, np.array([]), type="rgb")
    assert seen and not any(seen)
    seen.clear()
    with torch.no_grad():
        pts = torch.rand(8, 4, 3, device=DEV)
        vd = torch.nn.functional.normalize(torch.rand(8, 3, device=DEV), dim=-1)
        g.nerf(0, pts, vd, args)
    assert seen == [False]

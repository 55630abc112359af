"""GPU: the drop-in boundary.  After benerf_amd.dropin.install() a reference-style driver's imports
resolve to the HIP-backed mirrors, and a literal train.py-shaped loop (train.py:153-394: graph.forward, the loss lines,
backward, the five optimisers, the LR decay) runs on them.  Every random draw the loop makes is recorded and replayed
into the oracle, which runs the same iterations on the CPU: losses, gradients and parameters must agree.
Graph.forward's four event-window modes (model/nerf.py:162-191) are compared with the oracle's accumulation."""
import sys

import numpy as np
import pytest
import torch

import benerf_oracle as O
import golden_inputs as GI
from conftest import REPORT as REPORT_LINES, report

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("mlp_precision")]
DEV = "cuda:0"


@pytest.fixture
def installed():
    from benerf_amd import dropin
    before = {k: sys.modules.get(k) for k in dropin._ALIASES}
    names = dropin.install()
    yield names
    for k, v in before.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


class Recorder:
    """Records the draws of one Graph.forward: torch.randperm (pixels) and the four draws per render."""

    def __init__(self, monkeypatch):
        import benerf_amd.model.nerf as MN
        self.perms, self.draws = [], []
        orig_perm, orig_draw = torch.randperm, MN._draw

        def randperm(*a, **k):
            out = orig_perm(*a, **k)
            self.perms.append(out.detach().cpu())
            return out

        def draw(fn, shape, device, scale=None, **kw):
            out = orig_draw(fn, shape, device, scale, **kw)
            self.draws.append(out.detach().cpu())
            return out

        monkeypatch.setattr(torch, "randperm", randperm)
        monkeypatch.setattr(MN, "_draw", draw)

    def pop_step(self):
        assert len(self.perms) == 2 and len(self.draws) == 8, (len(self.perms), len(self.draws))
        d = self.draws
        out = (self.perms[0], self.perms[1], dict(t_rand=d[0], noise0=d[1], u=d[2], noise1=d[3]),
               dict(t_rand=d[4], noise0=d[5], u=d[6], noise1=d[7]))
        self.perms, self.draws = [], []
        return out


def _small_args(**over):
    from benerf_amd import workloads as WL
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=24, Rr=3, n=5)
    args = WL.make_args(wl, console_log_iter=1, **over)
    return args, WL.CAMERAS[wl["cam"]]


def test_reference_style_imports(installed):
    ns = {}
    exec("from model.nerf import *", ns)            # train.py:6 - and then uses np without importing it (train.py:66)
    for name in ("np", "torch", "nn", "F", "os", "Graph", "NeRF", "Model"):
        assert name in ns, name
    exec("import spline, bezier\n"
         "from model import optimize, embedder, component\n"
         "from run_nerf_helpers import init_nerf, render_image_test, render_video_test, get_rays, get_specific_rays, ndc_rays, sample_pdf\n"
         "from loss import imgloss\n"
         "from utils import math_utils, img_utils, event_utils\n", ns)
    sp = ns["spline"]
    for fn in ("cubic_spline_pose_unit_time", "linear_pose_unit_time", "se3_2_qt_parallel", "skew_symmetric", "taylor_B", "taylor_C",
               "exp_r2q_parallel", "log_q2r_parallel", "q_to_Q_parallel", "q_to_q_conj_parallel", "q_to_R_parallel"):
        assert callable(getattr(sp, fn)), fn
    assert callable(ns["bezier"].cubic_bezier_poses_unit_time)
    assert ns["optimize"].__name__ == "benerf_amd.model.optimize"


def test_train_py_shaped_loop(installed, monkeypatch):
    """train.py:153-394 with the reference's own names, three iterations, against the oracle on replayed draws."""
    ns = {}
    exec("from model.nerf import *\nfrom model import optimize\nfrom run_nerf_helpers import init_nerf\n"
         "from loss import imgloss\nfrom utils import img_utils\nfrom utils.math_utils import rgb2brightlog\n", ns)
    np_, optimize, init_nerf, imgloss, img_utils, rgb2brightlog = (ns[k] for k in ("np", "optimize", "init_nerf", "imgloss", "img_utils",
                                                                                 "rgb2brightlog"))
    args, cam = _small_args()
    H, W = cam["H"], cam["W"]
    K_rgb = np_.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np_.float32)
    rng = np_.random.default_rng(11)
    events = GI.synthetic_events(rng, cam, 60000)
    img = rng.random((1, H, W, args.channels))
    rgb_exp_ts = np_.array([0.0, 1.0])
    rec = Recorder(monkeypatch)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.set_default_tensor_type("torch.cuda.FloatTensor")      # train.py:472, literally
    try:
        torch.manual_seed(3)
        np_.random.seed(3)
        model = optimize.Model(args)
        graph = model.build_network(args)
        optimizer_nerf, optimizer_pose, optimizer_trans, optimizer_rgb_crf, optimizer_event_crf = model.setup_optimizer(args)
        mse_loss = imgloss.MSELoss()
        rgb2gray = img_utils.RGB2Gray()
        # the oracle's copy of the run
        o_state = None
        global_step = 0
        for i in range(3):
            if i == 0:
                init_nerf(graph.nerf)
                init_nerf(graph.nerf_fine)
                with torch.no_grad():   # open the density a little so that the gradients are not dominated by the far plane
                    graph.nerf.alpha_linear.bias += 1.0
                    graph.nerf_fine.alpha_linear.bias += 1.0
                p_c = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in graph.nerf.state_dict().items()}
                p_f = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in graph.nerf_fine.state_dict().items()}
                o_knots = graph.evt_knot_pose_se3.params.weight.detach().cpu().clone().requires_grad_(True)
                o_tr = graph.transform.params.weight.detach().cpu().clone().requires_grad_(True)
                o_params = list(p_c.values()) + list(p_f.values()) + [o_knots]      # optimize_trans is off (config default)
                o_state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in o_params]
            np_.random.seed(100 + i)
            ret_event, ret_rgb, ray_idx_event, ray_idx_rgb, events_accu = graph.forward(
                i, events, rgb_exp_ts, H, W, K_rgb, K_rgb, args, np_.array([]), np_.array([]))
            pixels_num = ray_idx_event.shape[0]
            ret_gray1 = {"rgb_map": ret_event["rgb_map"][:pixels_num], "rgb0": ret_event["rgb0"][:pixels_num]}
            ret_gray2 = {"rgb_map": ret_event["rgb_map"][pixels_num:], "rgb0": ret_event["rgb0"][pixels_num:]}
            ret_rgb = {"rgb_map": ret_rgb["rgb_map"], "rgb0": ret_rgb["rgb0"]}
            target_s = events_accu.reshape(-1, 1)[ray_idx_event]
            for opt in (optimizer_nerf, optimizer_pose, optimizer_trans, optimizer_rgb_crf, optimizer_event_crf):
                opt.zero_grad()
            loss = 0
            target_s *= torch.tensor(args.event_threshold)
            fine_bright2 = rgb2brightlog(ret_gray2["rgb_map"], args.dataset)
            fine_bright1 = rgb2brightlog(ret_gray1["rgb_map"], args.dataset)
            event_loss_fine = mse_loss((fine_bright2 - fine_bright1), target_s)
            event_loss_fine *= args.event_coeff_syn           # in place, as train.py:222 does it
            coarse_bright2 = rgb2brightlog(ret_gray2["rgb0"], args.dataset)
            coarse_bright1 = rgb2brightlog(ret_gray1["rgb0"], args.dataset)
            event_loss_coarse = mse_loss((coarse_bright2 - coarse_bright1), target_s)
            event_loss_coarse *= args.event_coeff_syn
            loss += event_loss_coarse + event_loss_fine
            image = torch.Tensor(img[0])
            target_rgb = image.reshape(-1, H * W, args.channels)[:, ray_idx_rgb].reshape(-1, args.channels)
            interval = target_rgb.shape[0]
            blur, blur0 = 0, 0
            for j in range(0, args.num_interpolated_pose):
                blur += ret_rgb["rgb_map"][j * interval:(j + 1) * interval]
                blur0 += ret_rgb["rgb0"][j * interval:(j + 1) * interval]
            blur, blur0 = blur / args.num_interpolated_pose, blur0 / args.num_interpolated_pose
            rgb_loss_fine = mse_loss(blur, target_rgb)
            rgb_loss_fine *= args.rgb_coeff                   # train.py:321
            rgb_loss_coarse = mse_loss(blur0, target_rgb)
            rgb_loss_coarse *= args.rgb_coeff
            loss += rgb_loss_fine + rgb_loss_coarse
            loss.backward()
            hip_knot_grad = graph.evt_knot_pose_se3.params.weight.grad.detach().cpu().clone()
            hip_w_grad = graph.nerf_fine.pts_linears[7].weight.grad.detach().cpu().clone()
            if args.optimize_nerf:
                optimizer_nerf.step()
            if args.optimize_pose:
                optimizer_pose.step()
            if args.optimize_trans:
                optimizer_trans.step()
            decay_steps = args.lrate_decay * 1000
            for opt, lr0, dr in ((optimizer_nerf, args.lrate, args.decay_rate), (optimizer_pose, args.pose_lrate, args.decay_rate_pose),
                                 (optimizer_trans, args.transform_lrate, args.decay_rate_transform)):
                for param_group in opt.param_groups:
                    param_group["lr"] = lr0 * (dr ** (global_step / decay_steps))
            global_step += 1

            # ---- the same iteration in the oracle, on the recorded draws -------------------------------------------------
            idx_e, idx_r, d_e, d_r = rec.pop_step()
            idx_e, idx_r = idx_e[:args.sampling_event_rays], idx_r[:args.sampling_rgb_rays // args.num_interpolated_pose]
            assert torch.equal(idx_e, ray_idx_event.cpu()) and torch.equal(idx_r, ray_idx_rgb.cpu())
            np_.random.seed(100 + i)
            low_t = float(np_.random.rand(1)[0] * (1 - args.accumulate_time_length))
            sel, up_t = O.event_window(events["ts"], low_t, args.accumulate_time_length)
            accu = O.accumulate_events(H, W, events["x"][sel], events["y"][sel], events["pol"][sel])
            assert np_.array_equal(events_accu.cpu().numpy(), accu.numpy())
            cfg = O.StepConfig(H=H, W=W, fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=args.channels,
                               n_samples=args.N_samples, n_importance=args.N_importance, n_poses=args.num_interpolated_pose,
                               dataset=args.dataset, threshold=args.event_threshold, window=args.accumulate_time_length)
            with torch.device("cpu"):
                o_loss, _ = O.step_loss(cfg, p_c, p_f, o_knots, o_tr, torch.tensor([low_t, up_t], dtype=torch.float32),
                                        torch.tensor([0.0, 1.0]), idx_e, idx_r, accu.reshape(-1, 1)[idx_e].double(),
                                        torch.from_numpy(img[0].reshape(H * W, args.channels)).float()[idx_r], d_e, d_r)
                for p in o_params + [o_tr]:
                    p.grad = None
                o_loss.backward()
                report("drop-in loop: loss, iteration %d" % i, loss.detach().float().cpu().reshape(1), o_loss.detach().float().reshape(1),
                       rtol=2e-4 if i == 0 else 5e-3)
                if i == 0:
                    report("drop-in loop: d knots, iteration 0", hip_knot_grad, o_knots.grad, atol=2e-3 * float(o_knots.grad.abs().max()), rtol=2e-3)
                    gw = p_f["pts_linears.7.weight"].grad
                    report("drop-in loop: d nerf_fine.pts_linears.7.weight, iteration 0", hip_w_grad, gw,
                           atol=2e-3 * float(gw.abs().max()), rtol=2e-3)
                k = max(i, 1) - 1
                lr = args.lrate * (args.decay_rate ** (k / decay_steps))
                with torch.no_grad():
                    for p, (m, v) in zip(o_params, o_state):
                        O.adam_update(p, p.grad, m, v, i + 1, lr)
        # after three Adam steps (each ~ lr in size) the two runs hold the same parameters up to a fraction of a step
        report("drop-in loop: knots after 3 iterations", graph.evt_knot_pose_se3.params.weight.detach().cpu(), o_knots.detach(),
               atol=0.25 * args.pose_lrate)
        w_hip, w_o = graph.nerf.pts_linears[3].weight.detach().cpu(), p_c["pts_linears.3.weight"].detach()
        frac = float(((w_hip - w_o).abs() > 0.25 * args.lrate).float().mean())
        print("drop-in loop: fraction of nerf.pts_linears.3.weight entries more than lr/4 apart after 3 steps: %.2e" % frac)
        assert frac < 0.02
        assert abs(optimizer_nerf.param_groups[0]["lr"] - args.lrate * args.decay_rate ** (2 / decay_steps)) < 1e-12
    finally:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.set_default_tensor_type(torch.FloatTensor)


@pytest.mark.parametrize("time_window,random_window", [(True, True), (True, False), (False, True), (False, False)])
def test_graph_forward_window_modes(installed, monkeypatch, time_window, random_window):
    """The four event-window modes of Graph.forward (model/nerf.py:162-191): accumulated image and the two window
    timestamps handed to get_pose_evt, against the reference's numpy lines restated here and the oracle's accumulation."""
    from model import optimize
    args, cam = _small_args(event_time_window=time_window, random_sampling_window=random_window, accumulate_time_length=0.13)
    H, W = cam["H"], cam["W"]
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
    events = GI.synthetic_events(np.random.default_rng(12), cam, 40000)
    torch.manual_seed(1)
    model = optimize.Model(args)
    model.graph.to(DEV)
    g = model.build_network(args)
    seen = {}
    orig = g.get_pose_evt

    def spy(a, events_ts, seg_num=None):
        seen["ts"] = events_ts.detach().cpu().numpy().copy()
        return orig(a, events_ts, seg_num) if seg_num is not None else orig(a, events_ts)

    monkeypatch.setattr(g, "get_pose_evt", spy)
    np.random.seed(21)
    _, _, idx_e, idx_r, accu = g.forward(0, events, np.array([0.0, 1.0]), H, W, K, K, args, np.array([]), np.array([]))
    # model/nerf.py:162-206 restated
    np.random.seed(21)
    wt = args.accumulate_time_length
    if time_window:
        if random_window:
            low_t = np.random.rand(1) * (1 - wt)
            upper_t = low_t + wt
        else:
            low_t = np.random.randint((1 - wt) // wt) * wt
            upper_t = np.min((low_t + wt, 1.0))
        sel = np.where((low_t <= events["ts"]) * (events["ts"] <= upper_t))
        ts_ref = np.stack((low_t, upper_t)).reshape(2)
        xs, ys, ps = events["x"][sel], events["y"][sel], events["pol"][sel]
    else:
        num = len(events["pol"])
        n_win = round(num * wt)
        lo = np.random.randint(num - n_win) if random_window else np.random.randint((num - n_win) // n_win) * n_win
        hi = int(lo + n_win)
        xs, ys, ps = events["x"][lo:hi], events["y"][lo:hi], events["pol"][lo:hi]
        ts_ref = events["ts"][lo:hi][np.array([0, int(n_win) - 1])]
    ref = O.accumulate_events(H, W, xs, ys, ps)
    assert accu.dtype == torch.float64 and np.array_equal(accu.cpu().numpy(), ref.numpy())
    report("window timestamps", seen["ts"], ts_ref.astype(np.float32), atol=0)
    assert idx_e.shape == (args.sampling_event_rays,) and idx_r.shape == (args.sampling_rgb_rays // args.num_interpolated_pose,)


def test_graph_forward_tum_vie(installed, monkeypatch):
    """dataset = TUM_VIE (model/nerf.py:194-196, 247-250): polarity 0 means -1 in the accumulated event image, and both
    renders take their pixel coordinates from the undistortion tables; with identity tables the renders equal the
    plain dataset's on the same draws."""
    from model import optimize
    cam = None
    rets = {}
    for dataset in ("BeNeRF_Unreal", "TUM_VIE"):
        args, cam = _small_args(dataset=dataset)
        H, W = cam["H"], cam["W"]
        K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
        events = GI.synthetic_events(np.random.default_rng(12), cam, 30000)
        if dataset == "TUM_VIE":
            events = dict(events, pol=np.where(events["pol"] < 0, 0.0, 1.0).astype(np.float32))     # TUM-VIE stores 0 / 1
        jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        ident = np.stack([ii, jj], -1).astype(np.float32)
        torch.manual_seed(1)
        model = optimize.Model(args)
        model.graph.to(DEV)
        g = model.build_network(args)
        np.random.seed(9)
        torch.manual_seed(2)
        remaps = (ident, ident) if dataset == "TUM_VIE" else (np.array([]), np.array([]))
        ret_e, ret_r, idx_e, idx_r, accu = g.forward(0, events, np.array([0.0, 1.0]), H, W, K, K, args, *remaps)
        rets[dataset] = (ret_e["rgb_map"].detach().cpu(), ret_r["rgb_map"].detach().cpu(), accu.cpu())
    a, b = rets["BeNeRF_Unreal"], rets["TUM_VIE"]
    assert torch.equal(a[2], b[2]), "polarity 0 must accumulate as -1"
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "identity undistortion tables must not change the renders"


def test_fused_graph_forward_equals_the_two_separate_renders(installed):
    """Graph.forward's fused pair of nodes (engine.SplinePosesPair + engine.RenderPair: one batched launch sequence, per-pose block
    outputs, parameter-independent work on a second stream) against the reference's own sequence of four calls (get_pose_evt,
    get_pose_rgb, render, render - what Graph.forward falls back to when a caller overrides one of them) on the same seeds: the
    rendered maps are bit-identical (a sample point's value does not depend on which other points share its launch), the loss
    gradients agree to round-off (the weight-gradient sums run over another partition of the points)."""
    from model import optimize
    args, cam = _small_args()
    H, W = cam["H"], cam["W"]
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
    events = GI.synthetic_events(np.random.default_rng(12), cam, 30000)
    outs = {}
    for which in ("fused", "separate"):
        torch.manual_seed(1)
        model = optimize.Model(args)
        model.graph.to(DEV)
        g = model.build_network(args)
        with torch.no_grad():
            g.nerf.alpha_linear.bias += 1.0
            g.nerf_fine.alpha_linear.bias += 1.0
        if which == "separate":
            orig = g.get_pose_evt
            g.get_pose_evt = lambda a, ts, seg_num=None: orig(a, ts, seg_num) if seg_num is not None else orig(a, ts)   # an instance override
        assert g._stock_queries() == (which == "fused")
        np.random.seed(9)
        torch.manual_seed(2)
        ret_e, ret_r, idx_e, idx_r, accu = g.forward(0, events, np.array([0.0, 1.0]), H, W, K, K, args, np.array([]), np.array([]))
        n = idx_e.shape[0]
        blur = 0
        R = idx_r.shape[0]
        for j in range(args.num_interpolated_pose):
            blur = blur + ret_r["rgb_map"][j * R:(j + 1) * R] + 0.5 * ret_r["rgb0"][j * R:(j + 1) * R]
        loss = ((ret_e["rgb_map"][n:] - ret_e["rgb_map"][:n]) ** 2).mean() + (ret_e["rgb0"] ** 2).mean() + (blur ** 2).mean() \
            + ret_r["acc_map"].mean() + 0.1 * ret_r["rgb_map"].sum()       # blocks AND whole maps, colours and opacity
        loss.backward()
        outs[which] = ({k: v.detach().cpu().clone() for k, v in list(ret_e.items()) + [("r_" + k2, v2) for k2, v2 in ret_r.items()]},
                       g.evt_knot_pose_se3.params.weight.grad.cpu().clone(), g.transform.params.weight.grad.cpu().clone(),
                       g.nerf.pts_linears[2].weight.grad.cpu().clone(), g.nerf_fine.views_linears[0].weight.grad.cpu().clone(),
                       idx_e.cpu(), idx_r.cpu(), float(loss))
    a, b = outs["fused"], outs["separate"]
    assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6]), "same pixel draws"
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k
    assert a[7] == b[7]
    for i, name in ((1, "knots"), (2, "transform"), (3, "nerf.pts_linears.2.weight"), (4, "nerf_fine.views_linears.0.weight")):
        report("fused Graph.forward vs two renders: d " + name, a[i], b[i], atol=2e-5 * float(b[i].abs().max()), rtol=1e-4)


def test_flat_adam_is_torch_adam(installed):
    """benerf_amd.optim.FlatAdam (what Model.setup_optimizer returns) against torch.optim.Adam on the same gradients: parameters and
    state after 5 steps with the learning rate rewritten between them as train.py:355-394 does, a parameter that gets no gradient in
    one step (torch leaves it and its counters alone), a gradient the caller replaced by a tensor of its own, gradients accumulated over
    two backward passes without zero_grad, state_dict() in the reference's checkpoint format and load_state_dict() into a fresh pair."""
    from benerf_amd.optim import FlatAdam
    torch.manual_seed(0)

    def nets():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Linear(9, 33), torch.nn.ReLU(), torch.nn.Linear(33, 4)).to(DEV)
    na, nb = nets(), nets()
    oa, ob = FlatAdam(na.parameters(), lr=5e-4), torch.optim.Adam(nb.parameters(), lr=5e-4, foreach=False)
    x = torch.randn(64, 9, device=DEV)

    def steps(na, nb, oa, ob, n0, n1):
        for it in range(n0, n1):
            for net, opt in ((na, oa), (nb, ob)):
                if it != 3:
                    opt.zero_grad()              # it == 3: accumulate on top of step 2's gradients
                loss = (net(x) ** 2).mean() if it != 1 else (net[0](x) ** 2).mean()      # it == 1: the last layer gets no gradient
                loss.backward()
                if it == 2:
                    p0 = next(net.parameters())
                    p0.grad = p0.grad * 2.0       # a caller-owned gradient tensor
                opt.step()
                for gr in opt.param_groups:
                    gr["lr"] = 5e-4 * 0.1 ** (it / 7.0)
    steps(na, nb, oa, ob, 0, 5)
    for (n_, pa), pb in zip(na.named_parameters(), nb.parameters()):
        report("FlatAdam vs torch Adam: " + n_, pa, pb, atol=2e-7, rtol=1e-6)
    import copy
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())      # as a checkpoint file would hold them (state_dict() hands out live tensors)
    assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"] and sorted(sa["state"]) == sorted(sb["state"])
    for k in sa["state"]:
        assert sorted(sa["state"][k]) == ["exp_avg", "exp_avg_sq", "step"]
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]), k
        report("FlatAdam state exp_avg_sq %d" % k, sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], atol=1e-12, rtol=1e-5)
    # the torch optimiser's checkpoint into a fresh FlatAdam (and vice versa), then two more steps
    nc, nd = nets(), nets()
    nc.load_state_dict(nb.state_dict())
    nd.load_state_dict(na.state_dict())
    oc, od = FlatAdam(nc.parameters(), lr=5e-4), torch.optim.Adam(nd.parameters(), lr=5e-4, foreach=False)
    oc.load_state_dict(sb)
    od.load_state_dict(sa)
    steps(nc, nd, oc, od, 5, 7)
    steps(na, nb, oa, ob, 5, 7)
    for (n_, pc), pa_ in zip(nc.named_parameters(), na.parameters()):
        report("FlatAdam resumed from torch's state_dict: " + n_, pc, pa_, atol=2e-7, rtol=1e-6)
    for (n_, pd), pb_ in zip(nd.named_parameters(), nb.parameters()):
        report("torch Adam resumed from FlatAdam's state_dict: " + n_, pd, pb_, atol=2e-7, rtol=1e-6)


def test_reference_shaped_loop_equals_train_step(installed, monkeypatch):
    """The two ways through the same kernels: tools/dropin_driver.py (train.py:153-394 on the drop-in modules: Graph.forward's
    fused nodes, the loss lines in torch, FlatAdam) against engine.TrainStep (K6 losses, flat Adam) on IDENTICAL inputs - the pixel
    draws, the eight sampling draws and the event window of every iteration of the loop are recorded and handed to the fused step.
    Six iterations at a C1-like size: the losses agree to round-off at every iteration (so every update in between did), and the
    parameters at the end differ by a fraction of a step on a handful of entries only (Adam's first updates are lr * sign(g): an
    entry whose gradient is round-off noise around zero moves by +-lr in either run)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dropin_driver as DD
    from benerf_amd import engine, kernels as K, workloads as WL
    from benerf_amd.model import optimize
    wl = dict(WL.WORKLOADS["C1"], S=16, Ni=16, Re=64, Rr=6, n=5)
    cam = WL.CAMERAS[wl["cam"]]
    args = WL.make_args(wl, optimize_trans=True)
    H, W, C = cam["H"], cam["W"], wl["channels"]
    events, img = DD.synthetic_scene(wl, cam, 60000, 5)
    rec = Recorder(monkeypatch)
    DD.cuda_default_tensor_type(True)
    try:
        drv = DD.Driver(args, cam, events, img, seed=4)
        snaps = []
        orig_init = drv.ns["init_nerf"]

        def init_and_snapshot(net):
            orig_init(net)
            with torch.no_grad():
                net.alpha_linear.bias += 1.0
            snaps.append({k: v.detach().clone() for k, v in net.state_dict().items()})
        drv.ns["init_nerf"] = init_and_snapshot
        knots0 = drv.graph.evt_knot_pose_se3.params.weight.detach().clone()
        tr0 = drv.graph.transform.params.weight.detach().clone()
        loop_losses, inputs = [], []
        for it in range(6):
            np.random.seed(300 + it)
            drv.iterate(1)
            loop_losses.append(float(drv.last_loss))
            idx_e, idx_r, d_e, d_r = rec.pop_step()
            np.random.seed(300 + it)
            low_t = float(np.random.rand(1)[0] * (1 - args.accumulate_time_length))
            inputs.append((low_t, low_t + args.accumulate_time_length, idx_e[:args.sampling_event_rays],
                           idx_r[:args.sampling_rgb_rays // args.num_interpolated_pose], d_e, d_r))
        loop_params = {k: v.detach().clone() for k, v in drv.graph.state_dict().items()}
    finally:
        DD.cuda_default_tensor_type(False)
    # the fused step on a second graph with the loop's initial parameters
    torch.manual_seed(0)
    model = optimize.Model(args)
    model.graph.to(DEV)
    g = model.build_network(args)
    g.nerf.load_state_dict(snaps[0])
    g.nerf_fine.load_state_dict(snaps[1])
    with torch.no_grad():
        g.evt_knot_pose_se3.params.weight.copy_(knots0)
        g.transform.params.weight.copy_(tr0)
    cam_o = engine.Camera(H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    ev = drv.graph._events_on_device(events)
    image = torch.from_numpy(img[0].reshape(H * W, C)).to(DEV)
    for it, (lo, up, idx_e, idx_r, d_e, d_r) in enumerate(inputs):
        accu = K.event_window_accumulate(ev["x"], ev["y"], ev["p"], ev["ts"], lo, up, H, W).view(-1)
        ts = torch.tensor([lo, up], dtype=torch.float32, device=DEV)
        dd = lambda d: engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))   # noqa: E731
        losses = step.step(ts, torch.tensor([0.0, 1.0], device=DEV), idx_e.to(DEV), idx_r.to(DEV), accu, image, dd(d_e), dd(d_r))
        report("reference-shaped loop vs TrainStep: loss, iteration %d" % it, np.array(loop_losses[it]), losses[0:1].cpu().numpy().reshape(()),
               rtol=2e-5 if it == 0 else 2e-3)
    step.check_range()
    worst = 0.0
    for k, v in g.state_dict().items():
        if k.startswith(("rgb_crf", "event_crf", "rgb_knot")):
            continue
        lr = args.pose_lrate if "knot" in k or "transform" in k else args.lrate
        frac = float(((v - loop_params[k]).abs() > 0.25 * lr).float().mean())
        worst = max(worst, frac)
        # the coarse network's first layers - behind the 2^9 x frequencies of the positional encoding: many entries whose gradient is
        # round-off noise around zero, each moved by lr * sign - scatter most (exact-f32 mode: 7.3 % of nerf.pts_linears.0.weight, 5.4 %
        # of pts_linears.1.weight, 2 % of pts_linears.0.bias; everything else below 1e-4): one generous limit, the losses are the test
        lim = 0.15
        REPORT_LINES.append("reference-shaped loop vs TrainStep, %-40s entries more than lr/4 apart after 6 iterations: %.4f" % (k, frac))
        assert frac < lim, "%s: %.3f of the entries more than lr / 4 apart after 6 iterations" % (k, frac)
    print("reference-shaped loop vs TrainStep: largest fraction of a tensor's entries more than lr/4 apart after 6 iterations: %.2e" % worst)

"""G11: short training-curve + PSNR parity on the procedural stand-in scene (oracle/curve_scene.py:
C1-shaped steps - 503 rays, 32 + 64 samples, 19 virtual poses).

tests/golden/g11_curve.npz holds, for 300 iterations: the loss curve and final PSNR of the UNMODIFIED reference on the
first input stream and of the oracle on three input streams (oracle/gen_golden.py g11).  Here the fused HIP TrainStep runs
the same iterations with identical inputs and draws, on all three streams, in both arithmetic modes.

Single steps agree to round-off (first steps below); over 300 Adam steps round-off is amplified, but on steps of this size
the trajectories stay together: the reference and the oracle end 0.08 dB apart on the shared stream.  Held here, in BOTH
arithmetic modes: final PSNR within 0.1 dB of the REFERENCE on its stream (the north-star criterion) and within 0.1 dB
of the oracle on the two other streams; the late loss matches the oracle's on every stream.  (On a 10x smaller scene
the same runs are chaotic - see tools/experiments/psnr_spread.py for the distribution over eight streams per mode.)"""
import numpy as np
import pytest
import torch

import curve_scene as CS
from conftest import report

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("mlp_precision")]
DEV = "cuda:0"


def _run(g11, frames, blurry, stream_seed):
    from benerf_amd import engine, kernels as K, workloads as WL
    from benerf_amd.model import optimize
    cam = CS.camera()
    wl = dict(cam="_g11", channels=CS.C, dataset="BeNeRF_Unreal", threshold=CS.THRESHOLD, window=0.125, n=CS.P, S=CS.S, Ni=CS.NI,
              Re=CS.RE, Rr=CS.RR)
    WL.CAMERAS["_g11"] = cam
    args = WL.make_args(wl)
    torch.manual_seed(0)
    model = optimize.Model(args)
    model.graph.to(DEV)
    g = model.build_network(args)
    pc, pf, knots = CS.student_init()
    with torch.no_grad():
        for net, p in ((g.nerf, pc), (g.nerf_fine, pf)):
            for name in K.LAYER_NAMES:
                lin = engine.getattr_path(net, name)
                lin.weight.copy_(p[name + ".weight"])
                lin.bias.copy_(p[name + ".bias"])
        g.evt_knot_pose_se3.params.weight.copy_(knots)
    cam_o = engine.Camera(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    step = engine.TrainStep(g, args, cam_o, cam_o, torch.device(DEV))
    rng = np.random.default_rng(stream_seed)
    rgb_ts = torch.tensor([0.0, 1.0], device=DEV)
    curve = []

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    for it in range(CS.N_STEPS):
        (t0, t1), accu, idx_e, idx_r, d_e, d_r = CS.step_inputs(rng, frames)
        losses = step.step(torch.tensor([t0, t1], dtype=torch.float32, device=DEV), rgb_ts, idx_e.to(DEV), idx_r.to(DEV),
                           accu.float().to(DEV).contiguous(), blurry, dd(d_e), dd(d_r))
        curve.append(losses[0:1])
    step.check_range()
    # final PSNR of the mid-exposure render (noise-free deterministic draws, as oracle/curve_scene.eval_psnr)
    n = CS.H * CS.W
    pose = K.spline_poses_fwd(step.knots, None, torch.tensor([0.5, 0.5], device=DEV), 1, 0)
    draws = engine.Draws(torch.full((n, CS.S), 0.5, device=DEV), None,
                         torch.linspace(0.02, 0.98, CS.NI, device=DEV).expand(n, CS.NI).contiguous(), None, noise_std=0.0)
    out, _ = engine._render_forward(cam_o, True, CS.S, CS.NI, draws, pose, torch.arange(n, device=DEV), step.net_c.packed,
                                    step.net_f.packed, False)
    return torch.cat(curve).cpu().numpy(), engine.psnr(out["rgb_map"].cpu(), frames[(CS.GRID - 1) // 2])


def test_training_curve_and_psnr(golden, mlp_precision):
    g11 = golden("g11_curve")
    frames = torch.from_numpy(g11["frames"])
    assert frames.shape[1] == CS.H * CS.W, "tests/golden/g11_curve.npz predates oracle/curve_scene.py: re-run oracle/gen_golden.py g11"
    blurry = frames.mean(0).to(DEV).contiguous()
    seeds = [int(s) for s in g11["stream_seeds"]]
    ref, ref_psnr = g11["ref_losses"], float(g11["ref_psnr"])
    ora, ora_psnr = g11["oracle_losses"], [float(p) for p in g11["oracle_psnr"]]
    hip_psnr = []
    for si, seed in enumerate(seeds):
        curve, ps = _run(g11, frames, blurry, seed)
        hip_psnr.append(ps)
        if si == 0:
            report("G11 loss curve vs reference, first 3 steps", curve[:3], ref[:3], atol=1e-6, rtol=1e-3)
            report("G11 loss curve vs reference, first 20 steps", curve[:20], ref[:20], atol=5e-5, rtol=5e-2)
        report("G11 loss curve vs oracle, first 3 steps, stream %d" % seed, curve[:3], ora[si][:3], atol=1e-6, rtol=1e-3)
        late_h, late_o = float(np.mean(curve[-50:])), float(np.mean(ora[si][-50:]))
        print("G11 [%s] stream %d: PSNR hip %.3f dB, oracle %.3f dB%s; mean loss of the last 50 steps hip %.3e, oracle %.3e"
              % (mlp_precision, seed, ps, ora_psnr[si], (", reference %.3f dB" % ref_psnr) if si == 0 else "", late_h, late_o))
        if si == 0:
            report("G11 final PSNR (dB) vs the reference, stream %d" % seed, np.array(ps), np.array(ref_psnr), atol=0.1)
        else:
            report("G11 final PSNR (dB) vs the oracle, stream %d" % seed, np.array(ps), np.array(ora_psnr[si]), atol=0.1)
        report("G11 mean loss of the last 50 steps vs the oracle, stream %d" % seed, np.array(late_h), np.array(late_o), rtol=0.05)
    report("G11 mean final PSNR over the three streams vs the oracle's", np.array(np.mean(hip_psnr)), np.array(np.mean(ora_psnr)), atol=0.1)

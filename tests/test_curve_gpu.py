"""G11: short training-curve + PSNR parity on the procedural stand-in scene.

tests/golden/g11_curve.npz holds the loss curve and final PSNR of the UNMODIFIED reference and of the
oracle for the same 300 iterations (oracle/gen_golden.py g11 + oracle/curve_scene.py).  Here the fused
HIP TrainStep runs those iterations with identical inputs and draws.  Single steps are identical up to
f32 round-off; over 300 noisy SGD steps two f32 implementations drift apart statistically, so late-curve
and PSNR are compared within the run-to-run band the reference/oracle pair itself shows (+0.1 dB)."""
import numpy as np
import pytest
import torch

import curve_scene as CS
from conftest import report

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("mlp_precision")]
DEV = "cuda:0"


def test_training_curve_and_psnr(golden):
    from benerf_amd import engine, kernels as K, workloads as WL
    from benerf_amd.model import optimize
    g11 = golden("g11_curve")
    frames = torch.from_numpy(g11["frames"])
    blurry = frames.mean(0).to(DEV).contiguous()
    cam = CS.camera()
    wl = dict(cam=None, channels=CS.C, dataset="BeNeRF_Unreal", threshold=CS.THRESHOLD, window=0.125, n=CS.P, S=CS.S, Ni=CS.NI,
              Re=CS.RE, Rr=CS.RR)
    WL.CAMERAS["_g11"] = cam
    wl["cam"] = "_g11"
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
    rng = np.random.default_rng(4242)
    rgb_ts = torch.tensor([0.0, 1.0], device=DEV)
    curve = []

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    for it in range(CS.N_STEPS):
        (t0, t1), accu, idx_e, idx_r, d_e, d_r = CS.step_inputs(rng, frames)
        losses = step.step(torch.tensor([t0, t1], dtype=torch.float32, device=DEV), rgb_ts, idx_e.to(DEV), idx_r.to(DEV),
                           accu.float().to(DEV).contiguous(), blurry, dd(d_e), dd(d_r))
        curve.append(losses[0:1])
    curve = torch.cat(curve).cpu().numpy()
    ref, ora = g11["ref_losses"], g11["oracle_losses"]
    report("G11 loss curve vs reference, first 3 steps", curve[:3], ref[:3], atol=1e-6, rtol=1e-3)
    report("G11 loss curve vs reference, first 20 steps", curve[:20], ref[:20], atol=5e-5, rtol=5e-2)
    band = float(np.median(np.abs(ref[-50:] - ora[-50:]) / ref[-50:]))
    drift = float(np.median(np.abs(curve[-50:] - ref[-50:]) / ref[-50:]))
    print("G11 late-curve median relative drift: hip-vs-ref %.3f, oracle-vs-ref %.3f" % (drift, band))
    assert drift <= max(0.2, 2.0 * band), "late loss curve outside the run-to-run band"
    assert float(np.mean(curve[-50:])) <= 1.5 * float(np.mean(ref[-50:])), "training must converge like the reference"

    # final PSNR of the mid-exposure render (noise-free deterministic draws, as oracle/curve_scene.eval_psnr)
    n = CS.H * CS.W
    pose = K.spline_poses_fwd(step.knots, None, torch.tensor([0.5, 0.5], device=DEV), 1, 0)
    draws = engine.Draws(torch.full((n, CS.S), 0.5, device=DEV), None,
                         torch.linspace(0.02, 0.98, CS.NI, device=DEV).expand(n, CS.NI).contiguous(), None, noise_std=0.0)
    out, _ = engine._render_forward(cam_o, True, CS.S, CS.NI, draws, pose, torch.arange(n, device=DEV), step.net_c.packed,
                                    step.net_f.packed, False)
    hip_psnr = engine.psnr(out["rgb_map"].cpu(), frames[(CS.GRID - 1) // 2])
    ref_psnr, ora_psnr = float(g11["ref_psnr"]), float(g11["oracle_psnr"])
    print("G11 PSNR: hip %.3f dB, reference %.3f dB, oracle %.3f dB" % (hip_psnr, ref_psnr, ora_psnr))
    # 300 noisy SGD steps amplify round-off: the unmodified reference and its op-for-op restatement (the oracle) already
    # end 0.16 dB apart, and our two MFMA modes land 0.2 dB apart on either side.  Both reference-grade results are
    # equally valid ground truth, so the HIP result is held to 0.1 dB + that band around their midpoint.
    mid = 0.5 * (ref_psnr + ora_psnr)
    report("G11 final PSNR (dB) vs reference / oracle midpoint", np.array(hip_psnr), np.array(mid), atol=0.1 + abs(ref_psnr - ora_psnr))
    assert hip_psnr >= min(ref_psnr, ora_psnr) - 0.1, "training must reach the reference's quality"

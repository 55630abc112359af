"""GPU experiment: run-to-run spread of the G11 training curve (oracle/curve_scene.py scene) on the HIP path.
300 Adam steps on a tiny scene are chaotic - two f32 implementations of the reference already end 0.16 dB apart - so
the question for an arithmetic mode is whether its final PSNR DISTRIBUTION over input streams matches, not one run.
Prints final PSNR per (mode, stream seed) and the per-mode mean / std."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import curve_scene as CS  # noqa: E402
from benerf_amd import engine, kernels as K, workloads as WL  # noqa: E402
from benerf_amd.model import optimize  # noqa: E402

DEV = "cuda:0"


def run(mode, seed, frames, n_steps):
    K.set_mlp_precision(mode)
    blurry = frames.mean(0).to(DEV).contiguous()
    cam = CS.camera()
    wl = dict(cam="_g11", channels=CS.C, dataset="BeNeRF_Unreal", threshold=CS.THRESHOLD, window=0.125, n=CS.P, S=CS.S, Ni=CS.NI,
              Re=CS.RE, Rr=CS.RR)
    WL.CAMERAS["_g11"] = cam
    args = WL.make_args(wl)
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
    rng = np.random.default_rng(seed)
    rgb_ts = torch.tensor([0.0, 1.0], device=DEV)

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))

    last = []
    for it in range(n_steps):
        (t0, t1), accu, idx_e, idx_r, d_e, d_r = CS.step_inputs(rng, frames)
        losses = step.step(torch.tensor([t0, t1], dtype=torch.float32, device=DEV), rgb_ts, idx_e.to(DEV), idx_r.to(DEV),
                           accu.float().to(DEV).contiguous(), blurry, dd(d_e), dd(d_r))
        if it >= n_steps - 50:
            last.append(losses[0:1])
    n = CS.H * CS.W
    pose = K.spline_poses_fwd(step.knots, None, torch.tensor([0.5, 0.5], device=DEV), 1, 0)
    draws = engine.Draws(torch.full((n, CS.S), 0.5, device=DEV), None,
                         torch.linspace(0.02, 0.98, CS.NI, device=DEV).expand(n, CS.NI).contiguous(), None, noise_std=0.0)
    out, _ = engine._render_forward(cam_o, True, CS.S, CS.NI, draws, pose, torch.arange(n, device=DEV), step.net_c.packed,
                                    step.net_f.packed, False)
    return engine.psnr(out["rgb_map"].cpu(), frames[(CS.GRID - 1) // 2]), float(torch.cat(last).mean())


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else CS.N_STEPS
    frames = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "g11_curve.npz"))["frames"])
    res = {}
    for mode in ("f32", "split"):
        for s in range(n_seeds):
            ps, ls = run(mode, 4242 + s, frames, n_steps)
            res.setdefault(mode, []).append(ps)
            print("mode %-5s seed %d: PSNR %.3f dB, mean loss of the last 50 steps %.5f" % (mode, 4242 + s, ps, ls), flush=True)
    for mode, v in res.items():
        v = np.array(v)
        print("mode %-5s: mean %.3f dB, std %.3f dB, min %.3f, max %.3f  (n = %d)" % (mode, v.mean(), v.std(ddof=1), v.min(), v.max(), len(v)))


if __name__ == "__main__":
    main()

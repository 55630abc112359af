"""GPU + CPU experiment: every gradient of one training step against a float64 evaluation of the same step.

Columns per gradient (max-entry error / max|truth|, | ||x|| / ||truth|| - 1 |, ||x - truth|| / ||truth||):
  o32   the oracle in float32 (the reference's arithmetic, torch CPU)
  f32   HIP path, exact-f32 MFMA mode
  split HIP path, default mode (f16 MFMA operands, f32 accumulation)
All four evaluate the same coarse / fine depths (the float32 oracle's, oracle/f64_truth.py).  Table 1: the float64 evaluation
also takes the float32 VALUES of the network inputs (pts = o + d z, viewdirs; straight-through gradient) - what is left is the
arithmetic behind the inputs.  Table 2: float64 end to end (the float32 rounding of pts, amplified 2^9 times by the positional
encoding, is then part of every float32 implementation's error).  Table 3: the HIP modes against the float32 oracle itself.
usage: python tools/experiments/f64_truth.py [case ...]      cases: g8_0 g8_1 g8_2 C2_eighth C2 (default: all but C2)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import benerf_oracle as O  # noqa: E402
import f64_truth as T  # noqa: E402
import golden_inputs as GI  # noqa: E402
from benerf_amd import engine, kernels as K, workloads as WL  # noqa: E402
from benerf_amd.model import optimize  # noqa: E402

DEV = "cuda:0"
G8_SPECS = [("unreal_C1", "unreal", 1, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3), ("unreal_C3", "unreal", 3, "BeNeRF_Unreal", 0.1, 19, 16, 16, 24, 3),
            ("e2syn_C3", "e2nerf_syn", 3, "E2NeRF_Synthetic", 0.2, 7, 32, 32, 16, 5)]


def case_inputs(name):
    if name.startswith("g8_"):
        si = int(name[3:])
        tag, cname, C, dataset, thr, P, S, Ni, Re, Rr = G8_SPECS[si]
        rng = np.random.default_rng(808 + si)
        cam = GI.CAMERAS[cname]
        window, chunks = (0.1 if "unreal" in tag else 0.25), 1
    else:
        wl = WL.WORKLOADS["C2"]
        frac = 8 if name.endswith("eighth") else 1
        cname, C, dataset, thr, P, S, Ni = wl["cam"], wl["channels"], wl["dataset"], wl["threshold"], wl["n"], wl["S"], wl["Ni"]
        Re, Rr = wl["Re"] // frac, max(wl["Rr"] // frac, 1)
        rng = np.random.default_rng(2024)
        cam = WL.CAMERAS[cname]
        window, chunks = wl["window"], (2 if frac == 8 else 12)
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


def hip_step(x, mode, z_forced):
    K.set_mlp_precision(mode)
    cam = x["cam"]
    wl = dict(cam="_t", channels=x["C"], dataset=x["dataset"], threshold=x["thr"], window=0.1, n=x["P"], S=x["S"], Ni=x["Ni"], Re=x["Re"],
              Rr=x["Rr"])
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

    def dd(d):
        return engine.Draws(*(d[k].to(DEV) for k in ("t_rand", "noise0", "u", "noise1")))
    zf = torch.cat([z_forced["evt"][1], z_forced["rgb"][1]]).to(DEV)
    losses = step.step(x["evt_ts"].to(DEV), torch.tensor([0.0, 1.0], device=DEV), x["idx_e"].to(DEV), x["idx_r"].to(DEV), x["accu"].to(DEV),
                       x["img"].to(DEV), dd(x["d_e"]), dd(x["d_r"]), z_fine_forced=zf)
    step.check_range()
    grads = {"knots": step.g_knots.cpu().clone(), "transform": step.g_transform.cpu().clone()}
    for nn_, fn in (("nerf", step.net_c), ("nerf_fine", step.net_f)):
        for i, name in enumerate(K.LAYER_NAMES):
            grads["%s.%s.weight" % (nn_, name)] = fn.gviews_w[i].cpu().clone()
            grads["%s.%s.bias" % (nn_, name)] = fn.gviews_b[i].cpu().clone()
    return float(losses[0]), grads


def run_case(name):
    x = case_inputs(name)
    cam = x["cam"]
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=x["C"], n_samples=x["S"],
                       n_importance=x["Ni"], n_poses=x["P"], dataset=x["dataset"], threshold=x["thr"])
    tacc = x["accu"].double().reshape(-1, 1)[x["idx_e"]]
    trgb = x["img"][x["idx_r"]]
    rts = torch.tensor([0.0, 1.0])
    a = (cfg, x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], rts, x["idx_e"], x["idx_r"], tacc, trgb, x["d_e"], x["d_r"])
    t0 = time.time()
    o32 = T.step_grads(*a, dtype=torch.float32, z_forced=None, n_chunks=x["chunks"])
    t1 = time.time()
    o64 = T.step_grads(*a, dtype=torch.float64, z_forced=o32["z"], n_chunks=x["chunks"])
    t2 = time.time()
    o64e = T.step_grads(*a, dtype=torch.float64, z_forced=o32["z"], n_chunks=x["chunks"], force_inputs=False)
    cands = {"o32": o32["grads"]}
    losses = {"o32": o32["loss"], "o64": o64["loss"]}
    for mode in ("f32", "split"):
        losses[mode], cands[mode] = hip_step(x, mode, o32["z"])
    K.set_mlp_precision("split")
    print("==== case %s: %d + %d rays, %d+%d samples, C=%d   (oracle f32 %.1f s, f64 %.1f s, %d torch threads)"
          % (name, 2 * x["Re"], x["P"] * x["Rr"], x["S"], x["Ni"], x["C"], t1 - t0, t2 - t1, torch.get_num_threads()))
    print("loss: " + "  ".join("%s %.9f" % kv for kv in losses.items()))
    print("-- table 2: against float64 end to end")
    print(T.format_table(T.error_table(o64e["grads"], cands), ["o32", "f32", "split"]))
    print("-- table 3: against the float32 oracle")
    print(T.format_table(T.error_table(o32["grads"], {k: v for k, v in cands.items() if k != "o32"}), ["f32", "split"]))
    print("-- table 1: against float64 behind float32 network inputs")
    tab = T.error_table(o64["grads"], cands)
    print(T.format_table(tab, ["o32", "f32", "split"]))
    worst = {}
    for lb in ("f32", "split"):
        for j, what in ((0, "max"), (1, "norm"), (2, "L2")):
            r = max(((row[lb][j] / max(row["o32"][j], 1e-12), row[lb][j], k) for k, row in tab.items()), key=lambda t: t[1])
            rr = max(((row[lb][j] / (row["o32"][j] + 2e-5), row[lb][j], k) for k, row in tab.items()), key=lambda t: t[0])
            worst[(lb, what)] = r
            print("%s %s error: largest %.2e (%s; %.1f x the float32 oracle's); largest ratio to (oracle's + 2e-5): %.2f (%s, %.2e)"
                  % (lb, what, r[1], r[2], r[0], rr[0], rr[2], rr[1]))
    sys.stdout.flush()


if __name__ == "__main__":
    names = sys.argv[1:] or ["g8_0", "g8_1", "g8_2", "C2_eighth"]
    for n in names:
        run_case(n)

"""CPU: the oracle (oracle/benerf_oracle.py) re-checked against the committed golden vectors
of the unmodified reference (tests/golden/*.npz, written by oracle/gen_golden.py)."""
import numpy as np
import torch

import benerf_oracle as O
import golden_inputs as GI
from conftest import report

T = torch.from_numpy


def test_g1_spline(golden):
    g = golden("g1_spline")
    for ci in range(int(g["n_cases"])):
        for traj in ("spline", "linear"):
            tag = "c%02d_%s" % (ci, traj)
            k = T(g[tag + "_knots"]).requires_grad_(True)
            tr = T(g[tag + "_transform"]).requires_grad_(True)
            P = g[tag + "_poses"].shape[0]
            poses = O.trajectory_poses(k, tr, tuple(g[tag + "_ts"]), P, traj)
            (poses * T(g[tag + "_G"])).sum().backward()
            report("G1 poses " + tag, poses, g[tag + "_poses"], atol=1e-7)
            report("G1 dknots " + tag, k.grad, g[tag + "_dknots"], atol=1e-5, rtol=1e-4)
            report("G1 dtransform " + tag, tr.grad, g[tag + "_dtransform"], atol=1e-5, rtol=1e-4)


def test_g2_rays(golden):
    g = golden("g2_rays")
    for cname, cam in GI.CAMERAS.items():
        poses, idx = T(g[cname + "_poses"]), T(g[cname + "_idx"])
        K = GI.cam_K(cam)
        o, d, v = O.make_rays(poses, idx, cam["H"], cam["W"], K, True)
        report("G2 ndc_o " + cname, o, g[cname + "_ndc_o"], atol=1e-7)
        report("G2 ndc_d " + cname, d, g[cname + "_ndc_d"], atol=1e-7)
        report("G2 viewdirs " + cname, v, g[cname + "_viewdirs"], atol=1e-7)


def test_g3_posenc(golden):
    g = golden("g3_posenc")
    report("G3 pe", O.posenc(T(g["pts"]), 10), g["pe"], atol=1e-7)
    report("G3 ped", O.posenc(T(g["dirs"]), 4), g["ped"], atol=1e-7)


def test_g5_composite(golden):
    g = golden("g5_composite")
    for C in (1, 3):
        raw, z, rd, noise = (T(g["C%d_%s" % (C, k)]) for k in ("raw", "z", "rays_d", "noise"))
        for noisy in (True, False):
            tag = "C%d_%s" % (C, "noise" if noisy else "clean")
            out = O.composite(raw, z, rd, noise if noisy else None, C)
            for nm, v in zip(("rgb_map", "disp", "acc", "weights", "depth", "sigma"), out):
                report("G5 %s %s" % (nm, tag), v, g[tag + "_" + nm], atol=1e-7, rtol=1e-6)
    assert np.isnan(g["C1_noise_disp"][4]), "all-zero-alpha ray must give disp = NaN like the reference"


def test_g6_sample_pdf(golden):
    g = golden("g6_sample_pdf")
    for kind in ("flat", "peaky", "zero"):
        for (S, Ni) in ((64, 64), (32, 32), (64, 128)):
            tag = "%s_S%d_N%d" % (kind, S, Ni)
            t_rand = T(g[tag + "_t_rand"])
            z = O.stratified_z(t_rand.shape[0], S, t_rand)
            bins = 0.5 * (z[..., 1:] + z[..., :-1])
            s, inds = O.sample_pdf_torch(bins, T(g[tag + "_w"]), T(g[tag + "_u"]))
            assert np.array_equal(inds.numpy(), g[tag + "_inds"]), tag
            report("G6 samples " + tag, s, g[tag + "_samples"], atol=1e-7)
            se, ie, _ = O.sample_pdf_exact(bins.numpy(), g[tag + "_w"], g[tag + "_u"])
            assert np.array_equal(ie, g[tag + "_inds_exact"]) and np.array_equal(se, g[tag + "_samples_exact"]), tag
            assert int((ie != g[tag + "_inds"]).sum()) <= 2, "exact restatement may differ from torch only at cdf ties"


def test_g7_render(golden):
    g = golden("g7_render")
    cam = GI.CAMERAS["unreal"]
    K = GI.cam_K(cam)
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
                GI.knots_init(rng)
                GI.transform_small(rng)
                idx = GI.pixel_indices(rng, cam, Rn)
                draws = GI.render_draws(rng, P * Rn, S, Ni)
                ret = O.render(pc, pf, T(g[tag + "_poses"]), idx, cam["H"], cam["W"], K, C, S, Ni, draws)
                for k in ("rgb_map", "rgb0", "acc_map", "sigma"):
                    report("G7 %s %s" % (k, tag), ret[k], g[tag + "_" + k], atol=2e-6, rtol=1e-5)


def test_g9_events(golden):
    g = golden("g9_events")
    rng = np.random.default_rng(909)
    cam = GI.CAMERAS["e2nerf_real"]
    ev = GI.synthetic_events(rng, cam, 100000)
    ev["x"][:5000] = ev["x"][0]
    ev["y"][:5000] = ev["y"][0]
    acc = O.accumulate_events(cam["H"], cam["W"], ev["x"], ev["y"], ev["pol"])
    assert np.array_equal(acc.numpy().astype(np.int16), g["accu"])


def test_g10_adam(golden):
    g = golden("g10_adam")
    p = T(g["p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr = 5e-4
    for step in range(5):
        O.adam_update(p, T(g["g_%d" % step]), m, v, step + 1, lr)
        lr = O.decayed_lr(5e-4, 0.1, step)
        report("G10 adam %d" % step, p, g["p_after_%d" % step], atol=1e-7, rtol=1e-6)
    for s in (0, 1, 1000, 80000):
        assert float(g["lr_%d" % s]) == O.decayed_lr(5e-4, 0.1, s)


def test_g12_spline_helpers_and_degenerate_knots(golden):
    """Oracle helpers vs the reference's single-step spline helpers, incl. the rows that reach their special branches
    (theta == 0, |w| < 1e-10), and whole trajectories with zero / 1e-12 / identical knots."""
    g = golden("g12_spline_ops")
    fns = {"exp_r2q": O.rotvec_to_quat, "log_q2r": O.quat_to_rotvec, "q_to_R": O.quat_to_rot, "q_to_Q": O.quat_left_matrix,
           "q_to_q_conj": O.quat_conj, "skew_symmetric": O.hat, "taylor_B": lambda x: O._taylor(x, 1),
           "taylor_C": lambda x: O._taylor(x, 2), "se3_2_qt": lambda x: torch.cat(O.se3_to_quat_trans(x), -1)}
    for name, fn in fns.items():
        report("G12 " + name, fn(T(g[name + "_in"])), g[name + "_out"], atol=1e-7)
    for tag in ("zero", "tiny", "equal"):
        for traj in ("spline", "linear"):
            key = "traj_%s_%s" % (tag, traj)
            report("G12 poses " + key, O.trajectory_poses(T(g[key + "_knots"]), None, (0.0, 1.0), 5, traj), g[key + "_poses"], atol=1e-7)
    # the reference's backward is NaN exactly where an unselected torch.where branch divides 0 by 0
    assert np.isnan(g["exp_r2q_din"][0, 0]).all() and np.isfinite(g["exp_r2q_din"][0, 1:]).all()
    assert np.isnan(g["traj_zero_spline_dknots"]).any()


def test_bezier_restatement_properties():
    """bezier.py of the reference cannot run (IndexError), so its evident intent is pinned by identities: the curve
    starts at control pose 0 and ends at control pose 3, its translation is the Bernstein combination, and with four
    identical control poses it is constant."""
    rng = np.random.default_rng(31)
    knots = GI.f32(rng.uniform(-0.4, 0.4, (4, 6)))
    ts = torch.tensor([0.0, 0.25, 0.5, 1.0])
    poses = O.bezier_poses(knots, ts)
    ends = [torch.cat([O.quat_to_rot(q), t.unsqueeze(-1)], -1).reshape(3, 4)
            for q, t in (O.se3_to_quat_trans(knots[k].reshape(1, 1, 6)) for k in (0, 3))]
    report("bezier start = control pose 0", poses[0], ends[0], atol=2e-5)
    report("bezier end = control pose 3", poses[3], ends[1], atol=2e-5)
    tk = torch.stack([O.se3_to_quat_trans(knots[k].reshape(1, 1, 6))[1].reshape(3) for k in range(4)])
    b = torch.tensor([0.125, 0.375, 0.375, 0.125])
    report("bezier translation at u = 1/2", poses[2, :, 3], b @ tk, atol=1e-6)
    same = O.bezier_poses(knots[:1].expand(4, 6), ts)
    report("bezier of identical control poses is constant", same, ends[0].expand(4, 3, 4), atol=1e-6)


def test_g13_tone_mappers(golden):
    """oracle.tone_map against the reference's ColorToneMapper / LuminanceToneMapper (model/component.py:38-149)."""
    g = golden("g13_crf")
    for name in ("rgb", "event"):
        p = {k: T(g["%s_p_%s" % (name, k)]).requires_grad_(True) for k in ("0.weight", "0.bias", "2.weight", "2.bias")}
        x = T(g[name + "_x"]).requires_grad_(True)
        y = O.tone_map(p, x)
        (y * T(g[name + "_G"])).sum().backward()
        report("G13 %s tone-mapper" % name, y, g[name + "_y"], atol=1e-7)
        report("G13 %s tone-mapper d input" % name, x.grad, g[name + "_dx"], atol=1e-7, rtol=1e-5)
        for k in p:
            report("G13 %s tone-mapper d %s" % (name, k), p[k].grad, g["%s_dp_%s" % (name, k)], atol=1e-7, rtol=1e-5)


def test_g14_barf_c2f(golden):
    """oracle.mlp_forward(barf=...) against the reference's NeRF.forward with use_barf_c2f (model/nerf.py:16-26,78-89)."""
    g = golden("g14_barf")
    rng = np.random.default_rng(1414)
    p = O.xavier_params(rng, 1)
    p["alpha_linear.bias"] += 1.0
    for it in (0, 12000, 23000, 60000):
        raw = O.mlp_forward(p, T(g["pts"]), T(g["viewdirs"]), barf=(it, 80000, 0.1, 0.5))
        report("G14 raw it=%d" % it, raw, g["it%d_raw" % it], atol=1e-7)


def _binned_case(rng, B, C, dataset, thr, P=5, S=8, Ni=8, Re=6, Rr=3):
    cam = GI.CAMERAS["e2nerf_real"]
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    pc["alpha_linear.bias"] += 1.0
    pf["alpha_linear.bias"] += 1.0
    cfg = O.StepConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C, n_samples=S,
                       n_importance=Ni, n_poses=P, dataset=dataset, threshold=thr)
    x = dict(cfg=cfg, pc=pc, pf=pf, knots=GI.knots_init(rng) * 3, tr=GI.transform_small(rng) * 0.1,
             idx_e=GI.pixel_indices(rng, cam, Re), idx_r=GI.pixel_indices(rng, cam, Rr),
             d_e=GI.render_draws(rng, (B + 1) * Re, S, Ni), d_r=GI.render_draws(rng, P * Rr, S, Ni),
             tacc=[T(rng.integers(-3, 4, (Re, 1)).astype(np.float64)) for _ in range(B)],
             timg=T(rng.random((Rr, C)).astype(np.float32)), evt_ts=torch.tensor([0.2, 0.6]), rgb_ts=torch.tensor([0.0, 1.0]))
    return x


def test_binned_oracle_is_the_sum_of_single_window_steps():
    """Dense event bins (an extension: the reference has one bin per step): oracle.step_loss_binned must equal the sum
    over bins of the single-window step_loss's EVENT part (pinned by G8) on that bin's pose pair and draws, plus one blur part;
    at B = 1 it is step_loss itself."""
    for si, (C, dataset, thr) in enumerate(((1, "BeNeRF_Unreal", 0.1), (3, "E2NeRF_Real", -1.0))):
        for B in (1, 3):
            rng = np.random.default_rng(500 + 10 * si + B)
            x = _binned_case(rng, B, C, dataset, thr)
            cfg, Re = x["cfg"], x["idx_e"].shape[0]
            kn = x["knots"].clone().requires_grad_(True)
            lb, parts = O.step_loss_binned(cfg, x["pc"], x["pf"], kn, x["tr"], x["evt_ts"], B, x["rgb_ts"], x["idx_e"], x["idx_r"],
                                           x["tacc"], x["timg"], x["d_e"], x["d_r"])
            lb.backward()
            ts = torch.linspace(float(x["evt_ts"][0]), float(x["evt_ts"][1]), B + 1, dtype=torch.float32)
            kn2 = x["knots"].clone().requires_grad_(True)
            tot = 0.0
            for b in range(B):
                rows = slice(b * Re, (b + 2) * Re)
                d_b = {k: v[rows] for k, v in x["d_e"].items()}
                l1, p1 = O.step_loss(cfg, x["pc"], x["pf"], kn2, x["tr"], ts[b:b + 2], x["rgb_ts"], x["idx_e"], x["idx_r"], x["tacc"][b],
                                     x["timg"], d_b, x["d_r"])
                tot = tot + p1["event"]
                if b == 0:
                    tot = tot + p1["rgb"]
                    if B == 1:
                        assert float(l1.detach()) == float(lb.detach()), "B = 1 must be step_loss itself"
            tot.backward()
            report("binned oracle loss B=%d %s" % (B, dataset), lb.detach().reshape(1), tot.detach().reshape(1), atol=1e-7, rtol=1e-6)
            report("binned oracle dknots B=%d %s" % (B, dataset), kn.grad, kn2.grad, atol=1e-6 * float(kn2.grad.abs().max()), rtol=1e-5)


def test_binned_vjp_oracle_matches_autograd():
    """f64_truth.step_grads_vjp(event_bins = B) (the chunked evaluation the full-size GPU tests use) against plain autograd
    through step_loss_binned, L2-normalised loss included."""
    import f64_truth as FT
    rng = np.random.default_rng(77)
    B = 3
    x = _binned_case(rng, B, 3, "E2NeRF_Real", -1.0)
    cfg = x["cfg"]
    qc = {k: v.clone().requires_grad_(True) for k, v in x["pc"].items()}
    kn = x["knots"].clone().requires_grad_(True)
    loss, _ = O.step_loss_binned(cfg, qc, x["pf"], kn, x["tr"], x["evt_ts"], B, x["rgb_ts"], x["idx_e"], x["idx_r"], x["tacc"], x["timg"],
                                 x["d_e"], x["d_r"])
    loss.backward()
    got = FT.step_grads_vjp(cfg, x["pc"], x["pf"], x["knots"], x["tr"], x["evt_ts"], x["rgb_ts"], x["idx_e"], x["idx_r"], x["tacc"],
                            x["timg"], x["d_e"], x["d_r"], n_chunks=2, event_bins=B)
    lv = float(loss.detach())
    assert abs(got["loss"] - lv) <= 1e-6 * max(1.0, abs(lv))
    report("binned vjp dknots", got["grads"]["knots"].float(), kn.grad, atol=2e-5 * float(kn.grad.abs().max()), rtol=1e-3)
    w = "pts_linears.3.weight"
    report("binned vjp d" + w, got["grads"]["nerf." + w].float(), qc[w].grad, atol=2e-5 * float(qc[w].grad.abs().max()), rtol=1e-3)

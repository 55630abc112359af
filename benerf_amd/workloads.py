"""Benchmark / parity workloads of BASELINE.json restated in the reference's own terms
(SURVEY.md section 8 table): camera, channels, dataset branch, samples, batch split.

"X rays" => sampling_event_rays = X/4 pixels (2 poses -> X/2 rays) and R_rgb = floor(X/2/n)
pixels (n poses -> ~X/2 rays): the reference's default 1:1 split (model/nerf.py:214,224).
"""
import types

CAMERAS = {
    "unreal": dict(H=480, W=768, fx=548.409, fy=548.409, cx=384.0, cy=240.0),                  # configs/benerf_unreal/*.txt:12-24
    "e2nerf_syn": dict(H=800, W=800, fx=1111.1110311937682, fy=1111.1110311937682, cx=400.0, cy=400.0),
    "e2nerf_real": dict(H=260, W=346, fx=653.98456, fy=653.98456, cx=173.0, cy=130.0),
}

WORKLOADS = {
    # id: camera, channels, dataset, threshold, window, n poses, S, Ni, event pixels, rgb pixels (per GPU)
    "C1": dict(cam="unreal", channels=1, dataset="BeNeRF_Unreal", threshold=0.1, window=0.1, n=19, S=32, Ni=32, Re=128, Rr=13,
               name="benerf_unreal/livingroom_gray, 512 rays, 32+64 samples"),
    "C2": dict(cam="unreal", channels=1, dataset="BeNeRF_Unreal", threshold=0.1, window=0.1, n=19, S=64, Ni=64, Re=1024, Rr=107,
               name="benerf_unreal/livingroom_gray, 4096 rays, 64+128 samples, 19 virtual poses"),
    "C3": dict(cam="unreal", channels=3, dataset="BeNeRF_Unreal", threshold=0.1, window=0.1, n=19, S=64, Ni=64, Re=1024, Rr=107,
               name="benerf_unreal/whiteroom color, 4096 rays, 64+128 samples"),
    "C4": dict(cam="e2nerf_syn", channels=3, dataset="E2NeRF_Synthetic", threshold=0.2, window=0.25, n=19, S=64, Ni=64, Re=2048,
               Rr=215, name="e2nerf_synthetic/lego, 8192 rays, 64+128 samples"),
    "C5": dict(cam="e2nerf_real", channels=3, dataset="E2NeRF_Real", threshold=-1.0, window=0.25, n=31, S=64, Ni=128, Re=2048,
               Rr=132, name="e2nerf_real/letter, 8192 rays, 64+192 samples, 31 virtual poses"),
}


def make_args(wl, **over):
    """Namespace with the reference's flag names (config.py) for the ~35 flags the path reads."""
    w = WORKLOADS[wl] if isinstance(wl, str) else wl
    cam = CAMERAS[w["cam"]]
    d = dict(
        channels=w["channels"], N_samples=w["S"], N_importance=w["Ni"], use_viewdirs=True, multires=10,
        multires_views=4, i_embed=0, use_barf_c2f=False, ndc=True, dataset=w["dataset"], traj="spline",
        num_interpolated_pose=w["n"], rgb_crf_net_hidden=0, rgb_crf_net_width=128, event_crf_net_hidden=0,
        event_crf_net_width=128, lrate=5e-4, pose_lrate=5e-4, transform_lrate=5e-4, rgb_crf_lrate=5e-4,
        event_crf_lrate=5e-4, chunk=4096, max_iter=80000, event_time_window=True, random_sampling_window=True,
        accumulate_time_length=w["window"], event_height=cam["H"], event_width=cam["W"],
        sampling_event_rays=w["Re"], sampling_rgb_rays=w["Rr"] * w["n"], event_threshold=w["threshold"],
        event_coeff_syn=0.1, event_coeff_real=2.0, rgb_coeff=1.0, rgb_loss=True, event_loss=True,
        optimize_nerf=True, optimize_pose=True, optimize_trans=False, optimize_rgb_crf=False,
        optimize_event_crf=False, decay_rate=0.1, decay_rate_pose=0.1, decay_rate_transform=0.1,
        decay_rate_rgb_crf=0.1, decay_rate_event_crf=0.1, lrate_decay=200)
    d.update(over)
    return types.SimpleNamespace(**d)


def rays_per_step(wl):
    """rays one step renders: (event bins + 1) event poses x Re pixels + n blur poses x Rr pixels (one bin = the reference's step)"""
    w = WORKLOADS[wl] if isinstance(wl, str) else wl
    return (w.get("bins", 1) + 1) * w["Re"] + w["n"] * w["Rr"]


def mlp_flops_per_point(channels):
    """2 x MACs of NeRF.forward per sample point: 1 186 304 (C=1), 1 186 816 (C=3)."""
    macs = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 + 256 * 256 + 283 * 128 + 128 * channels
    return 2 * macs

"""Seeded synthetic inputs shared by oracle/gen_golden.py, tests/ and bench.py.

TEST INFRASTRUCTURE (lives under oracle/).  Pure numpy `default_rng` so every
consumer regenerates bit-identical inputs without torch's RNG and the golden
files mostly carry OUTPUTS.
"""
import numpy as np
import torch

# camera rows of SURVEY.md section 8 (cfg files lines 12-24 of the reference configs)
CAMERAS = {
    "unreal": dict(H=480, W=768, fx=548.409, fy=548.409, cx=384.0, cy=240.0),      # configs/benerf_unreal/*.txt
    "blender": dict(H=400, W=600, fx=541.850232, fy=541.850232, cx=300.0, cy=200.0),  # configs/benerf_blender/*.txt
    "e2nerf_syn": dict(H=800, W=800, fx=1111.1110311937682, fy=1111.1110311937682, cx=400.0, cy=400.0),
    "e2nerf_real": dict(H=260, W=346, fx=653.98456, fy=653.98456, cx=173.0, cy=130.0),
}


def cam_K(cam):
    return torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=torch.float32)


def f32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def knots_init(rng):
    """Reference init range rand*0.01 (model/optimize.py:22-24)."""
    return f32(rng.uniform(0.0, 0.01, (4, 6)))


def knots_stress(rng):
    return f32(rng.uniform(-0.5, 0.5, (4, 6)))


def transform_small(rng):
    return f32(rng.uniform(-0.05, 0.05, (1, 6)))


def pixel_indices(rng, cam, n):
    return torch.from_numpy(rng.permutation(cam["H"] * cam["W"])[:n].astype(np.int64))


def render_draws(rng, n_rays, n_samples, n_importance, noise=True):
    """The four RNG draws of one Graph.render call in reference order."""
    d = {"t_rand": f32(rng.random((n_rays, n_samples)))}
    d["noise0"] = f32(rng.standard_normal((n_rays, n_samples))) if noise else None
    d["u"] = f32(rng.random((n_rays, n_importance)))
    d["noise1"] = f32(rng.standard_normal((n_rays, n_samples + n_importance))) if noise else None
    return d


def synthetic_events(rng, cam, n_events):
    """x~U{0..W-1}, y~U{0..H-1}, ts sorted U[0,1), pol in {-1,+1}."""
    xs = rng.integers(0, cam["W"], n_events).astype(np.int64)
    ys = rng.integers(0, cam["H"], n_events).astype(np.int64)
    ts = np.sort(rng.random(n_events))
    ps = (rng.integers(0, 2, n_events) * 2 - 1).astype(np.float32)
    return {"x": xs, "y": ys, "ts": ts, "pol": ps}

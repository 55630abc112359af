"""G11: procedurally generated stand-in scene + short training run.

TEST INFRASTRUCTURE (oracle/).  The real BeNeRF datasets are not available, so PSNR parity is
shown on a self-consistent synthetic scene: a fixed random "teacher" NeRF and a teacher camera
trajectory render (noise-free) the sharp frames on a time grid; their average is the blurry image
and differences of their log-brightness are the event targets.  A "student" (different init) is
then trained for N steps with the reference's recipe.  Both the oracle and the HIP path run the
SAME step sequence (pixels, windows and all four RNG draws per render come from one numpy stream),
so their loss curves and final PSNR can be compared directly.

    python oracle/gen_golden.py g11         # reference + oracle runs -> tests/golden/g11_curve.npz (about 15 min on 8 cores)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import benerf_oracle as O  # noqa: E402
import golden_inputs as GI  # noqa: E402

# C1-shaped steps (503 rays, 32 + 64 samples, 19 virtual poses) on a 48 x 64 image
H, W, FOCAL = 48, 64, 80.0
C, S, NI, P = 1, 32, 64, 19
RE, RR = 128, 13          # event pixels / blur pixels per step: 2*128 + 19*13 = 503 rays
N_STEPS = 300
STREAM_SEEDS = (4242, 4243, 4244)   # input streams (pixels, windows, the four draws per render) of the golden runs
GRID = 33                 # sharp teacher frames at t = k / (GRID-1)
WINDOW = 4                # event window = 4 grid steps (0.125)
THRESHOLD = 0.1
LR = 5e-4


def camera():
    return dict(H=H, W=W, fx=FOCAL, fy=FOCAL, cx=W / 2.0, cy=H / 2.0)


def teacher(seed=1234):
    rng = np.random.default_rng(seed)
    pc, pf = O.xavier_params(rng, C), O.xavier_params(rng, C)
    for p in (pc, pf):
        p["alpha_linear.bias"] += 1.5
        p["rgb_linear.weight"] *= 4.0          # some contrast
    knots = GI.f32(rng.uniform(-0.04, 0.04, (4, 6)))
    return pc, pf, knots


def teacher_frames():
    """Noise-free sharp renders of the whole image on the time grid -> [GRID, H*W, C]."""
    pc, pf, knots = teacher()
    cam = camera()
    K = GI.cam_K(cam)
    idx = torch.arange(H * W)
    frames = []
    with torch.no_grad():
        for k in range(GRID):
            t = k / (GRID - 1)
            pose = O.trajectory_poses(knots, None, (t, t), 1, "spline")
            n = H * W
            draws = {"t_rand": torch.full((n, S), 0.5), "noise0": None, "u": torch.linspace(0.02, 0.98, NI).expand(n, NI).contiguous(),
                     "noise1": None}
            ret = O.render(pc, pf, pose, idx, H, W, K, C, S, NI, draws, exact_pdf=True)
            frames.append(ret["rgb_map"].clamp(1e-3, 1.0))
    return torch.stack(frames)


def step_inputs(rng, frames):
    """One training step's inputs from the shared numpy stream."""
    k0 = int(rng.integers(0, GRID - WINDOW))
    t0, t1 = k0 / (GRID - 1), (k0 + WINDOW) / (GRID - 1)
    accu = ((torch.log(frames[k0 + WINDOW] + 1e-9) - torch.log(frames[k0] + 1e-9)) / THRESHOLD).reshape(-1)   # [H*W]
    idx_e = torch.from_numpy(rng.permutation(H * W)[:RE].astype(np.int64))
    idx_r = torch.from_numpy(rng.permutation(H * W)[:RR].astype(np.int64))
    d_e = GI.render_draws(rng, 2 * RE, S, NI)
    d_r = GI.render_draws(rng, P * RR, S, NI)
    return (t0, t1), accu, idx_e, idx_r, d_e, d_r


def student_init(seed=77):
    rng = np.random.default_rng(seed)
    return O.xavier_params(rng, C), O.xavier_params(rng, C), GI.knots_init(rng)


def eval_psnr(pc, pf, knots, frames):
    """PSNR of the student's mid-exposure render (its own trajectory, t = 0.5) vs the teacher's sharp frame."""
    cam = camera()
    n = H * W
    with torch.no_grad():
        pose = O.trajectory_poses(knots, None, (0.5, 0.5), 1, "spline")
        draws = {"t_rand": torch.full((n, S), 0.5), "noise0": None, "u": torch.linspace(0.02, 0.98, NI).expand(n, NI).contiguous(),
                 "noise1": None}
        ret = O.render(pc, pf, pose, torch.arange(n), H, W, GI.cam_K(cam), C, S, NI, draws, exact_pdf=True)
    return O.psnr(ret["rgb_map"], frames[(GRID - 1) // 2]), ret["rgb_map"]


def run_oracle(n_steps=N_STEPS, log=None, stream_seed=STREAM_SEEDS[0], frames=None):
    torch.set_num_threads(8)
    if frames is None:
        frames = teacher_frames()
    blurry = frames.mean(0)                                  # [H*W, C]
    cam = camera()
    cfg = O.StepConfig(H=H, W=W, fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], channels=C, n_samples=S,
                       n_importance=NI, n_poses=P, dataset="BeNeRF_Unreal", threshold=THRESHOLD)
    pc, pf, knots = student_init()
    for p in list(pc.values()) + list(pf.values()) + [knots]:
        p.requires_grad_(True)
    tr = torch.zeros(1, 6)
    params = list(pc.values()) + list(pf.values()) + [knots]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    rng = np.random.default_rng(stream_seed)
    losses = []
    for it in range(n_steps):
        (t0, t1), accu, idx_e, idx_r, d_e, d_r = step_inputs(rng, frames)
        loss, _ = O.step_loss(cfg, pc, pf, knots, tr, torch.tensor([t0, t1]), torch.tensor([0.0, 1.0]), idx_e, idx_r,
                              (accu * THRESHOLD / THRESHOLD).double().reshape(-1, 1)[idx_e], blurry[idx_r], d_e, d_r, exact_pdf=True)
        for p in params:
            p.grad = None
        loss.backward()
        k = max(it, 1) - 1                                   # reference LR schedule (train.py:355-394)
        lr = LR * (0.1 ** (k / (200 * 1000)))
        with torch.no_grad():
            for p, (m, v) in zip(params, state):
                O.adam_update(p, p.grad, m, v, it + 1, lr)
        losses.append(float(loss))
        if log and it % 25 == 0:
            log("step %d loss %.6f" % (it, losses[-1]))
    ps, img = eval_psnr(pc, pf, knots, frames)
    return np.array(losses), ps, img.numpy(), frames


if __name__ == "__main__":
    losses, ps, img, frames = run_oracle(log=print)
    print("final PSNR %.3f dB" % ps)

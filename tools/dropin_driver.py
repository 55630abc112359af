"""A reference-shaped training driver for the drop-in modules (bench.py's `dropin` leg).

What `train.py:153-394` does per iteration, with the reference's own names and in its order, on modules imported THE WAY
train.py IMPORTS THEM (`from model.nerf import *`, `from model import optimize`, ...) after `benerf_amd.dropin.install()`:
`graph.forward` -> the event-loss lines on torch tensors (`rgb2brightlog`, `rgb2gray`, `mse_loss`, the L2-normalised form for the
real datasets) -> the blur-average loop over `num_interpolated_pose` slices -> every `logger.write(name, x.item())` of the
reference (six host synchronisations per iteration, a seventh in the image upload) -> `loss.backward()` -> the `optimizer_*.step()` calls -> the five
learning-rate updates.  Nothing here calls benerf_amd directly: this file stands for the user's unchanged train.py, and the
time it measures is the time such a user sees.  The oracle comparison of this very loop is tests/test_dropin_gpu.py.
"""
import time
import warnings

import numpy as np
import torch


class ItemLogger:
    """logger.write of the reference (logger/wandb_logger.py): keeps the scalar; the `.item()` at the call site is the cost."""

    def __init__(self):
        self.last = {}

    def write(self, name, value):
        self.last[name] = value

    def update_buffer(self):
        pass


def import_like_train_py():
    ns = {}
    exec("from model.nerf import *\n"                       # train.py:6 (np / torch / nn / F / os come with it)
         "from model import optimize\n"
         "from run_nerf_helpers import init_nerf, render_image_test, render_video_test\n"
         "from loss import imgloss\n"
         "from utils import img_utils\n"
         "from utils.math_utils import rgb2brightlog\n", ns)
    return ns


def cuda_default_tensor_type(on):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.set_default_tensor_type("torch.cuda.FloatTensor" if on else torch.FloatTensor)      # train.py:472


class Driver:
    """Everything train.py builds in front of its loop (train.py:110-152), then `iterate(n)`."""

    def __init__(self, args, cam, events, img, seed=0, logger=None):
        from benerf_amd import dropin
        dropin.install()
        ns = import_like_train_py()
        self.ns, self.args, self.events, self.img = ns, args, events, img
        self.H, self.W = cam["H"], cam["W"]
        self.K_rgb = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], dtype=np.float32)
        self.K_event = self.K_rgb.copy()
        self.img_xy_remap, self.evt_xy_remap = np.array([]), np.array([])
        self.rgb_exp_ts = np.array([0.0, 1.0])
        self.logger = logger or ItemLogger()
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.model = ns["optimize"].Model(args)
        self.graph = self.model.build_network(args, poses=None, event_poses=None)
        self.optimizers = self.model.setup_optimizer(args)
        self.optimizers[0].zero_grad()
        self.mse_loss = ns["imgloss"].MSELoss()
        self.rgb2gray = ns["img_utils"].RGB2Gray()
        self.i = 0
        self.global_step = 0
        self.last_loss = None
        self.host_times = None      # {"section": seconds} summed over iterations when a dict (run(host_times=True))

    def iterate(self, n_iters):
        """n_iters iterations of train.py:153-394.  Returns wall seconds (host clock; the caller synchronises around it)."""
        ns, args, logger, graph = self.ns, self.args, self.logger, self.graph
        rgb2brightlog, init_nerf, mse_loss, rgb2gray = ns["rgb2brightlog"], ns["init_nerf"], self.mse_loss, self.rgb2gray
        optimizer_nerf, optimizer_pose, optimizer_trans, optimizer_rgb_crf, optimizer_event_crf = self.optimizers
        H, W, img, events = self.H, self.W, self.img, self.events
        t0 = time.perf_counter()
        ht = self.host_times

        def lap(name, since):
            now = time.perf_counter()
            if ht is not None:
                ht[name] = ht.get(name, 0.0) + (now - since)
            return now

        for _ in range(n_iters):
            i = self.i
            tl = time.perf_counter()
            if i == 0:
                init_nerf(graph.nerf)
                init_nerf(graph.nerf_fine)
            ret_event, ret_rgb, ray_idx_event, ray_idx_rgb, events_accu = graph.forward(
                i, events, self.rgb_exp_ts, H, W, self.K_rgb, self.K_event, args, self.img_xy_remap, self.evt_xy_remap)
            tl = lap("graph.forward (host: queues the render)", tl)
            pixels_num = ray_idx_event.shape[0]
            ret_gray1 = {"rgb_map": ret_event["rgb_map"][:pixels_num], "rgb0": ret_event["rgb0"][:pixels_num]}
            ret_gray2 = {"rgb_map": ret_event["rgb_map"][pixels_num:], "rgb0": ret_event["rgb0"][pixels_num:]}
            ret_rgb = {"rgb_map": ret_rgb["rgb_map"], "rgb0": ret_rgb["rgb0"]}
            target_s = events_accu.reshape(-1, 1)[ray_idx_event]
            if args.optimize_event_crf:
                ret_gray1 = {k: graph.event_crf.forward(v) for k, v in ret_gray1.items()}
                ret_gray2 = {k: graph.event_crf.forward(v) for k, v in ret_gray2.items()}
            if args.optimize_rgb_crf:
                ret_rgb = {k: graph.rgb_crf.forward(v) for k, v in ret_rgb.items()}
            for opt in (optimizer_nerf, optimizer_pose, optimizer_trans, optimizer_rgb_crf, optimizer_event_crf):
                opt.zero_grad()
            loss = 0
            if args.event_loss:
                gray = rgb2gray if args.channels == 3 else (lambda t: t)
                syn = args.event_threshold > 0
                if syn:
                    target_s *= torch.tensor(args.event_threshold)
                terms = {}
                for which, key in (("fine", "rgb_map"), ("coarse", "rgb0")):      # train.py:207-292: fine first, then "rgb0"
                    bright2 = rgb2brightlog(gray(ret_gray2[key]), args.dataset)
                    bright1 = rgb2brightlog(gray(ret_gray1[key]), args.dataset)
                    if syn:
                        term = mse_loss((bright2 - bright1), target_s)
                        term *= args.event_coeff_syn
                    else:
                        render_brightness_diff = bright2 - bright1
                        render_norm = render_brightness_diff / (torch.linalg.norm(render_brightness_diff, dim=0, keepdim=True) + 1e-9)
                        target_s_norm = target_s / (torch.linalg.norm(target_s, dim=0, keepdim=True) + 1e-9)
                        term = mse_loss(render_norm, target_s_norm)
                        term *= args.event_coeff_real
                    logger.write("train_event_loss_" + which, term.item())
                    terms[which] = term
                event_loss = terms["coarse"] + terms["fine"]
                logger.write("train_event_loss", event_loss.item())
                loss += event_loss
            if args.rgb_loss:
                image = torch.Tensor(img[0])
                target_s = image.reshape(-1, H * W, args.channels)
                target_s = target_s[:, ray_idx_rgb].reshape(-1, args.channels)
                interval = target_s.shape[0]
                synthesized_blur_rgb = 0
                synthesized_blur_rgb0 = 0
                for j in range(0, args.num_interpolated_pose):
                    synthesized_blur_rgb += ret_rgb["rgb_map"][j * interval:(j + 1) * interval]
                    synthesized_blur_rgb0 += ret_rgb["rgb0"][j * interval:(j + 1) * interval]
                    if (j + 1) % args.num_interpolated_pose == 0:
                        synthesized_blur_rgb = synthesized_blur_rgb / args.num_interpolated_pose
                        synthesized_blur_rgb0 = synthesized_blur_rgb0 / args.num_interpolated_pose
                rgb_loss_fine = mse_loss(synthesized_blur_rgb, target_s)
                rgb_loss_fine *= args.rgb_coeff
                logger.write("train_rgb_loss_fine", rgb_loss_fine.item())
                rgb_loss_coarse = mse_loss(synthesized_blur_rgb0, target_s)
                rgb_loss_coarse *= args.rgb_coeff
                logger.write("train_rgb_loss_coarse", rgb_loss_coarse.item())
                rgb_loss = rgb_loss_fine + rgb_loss_coarse
                logger.write("train_rgb_loss", rgb_loss)
                loss += rgb_loss
            logger.write("train_loss", loss.item())
            tl = lap("loss lines incl. .item() waits", tl)
            loss.backward()
            tl = lap("loss.backward() (host: autograd over the loss lines + queues the render's backward)", tl)
            for flag, opt in ((args.optimize_nerf, optimizer_nerf), (args.optimize_pose, optimizer_pose), (args.optimize_trans, optimizer_trans),
                              (args.optimize_rgb_crf, optimizer_rgb_crf), (args.optimize_event_crf, optimizer_event_crf)):
                if flag:
                    opt.step()
            decay_steps = args.lrate_decay * 1000
            for opt, lr0, rate in ((optimizer_nerf, args.lrate, args.decay_rate), (optimizer_pose, args.pose_lrate, args.decay_rate_pose),
                                   (optimizer_trans, args.transform_lrate, args.decay_rate_transform),
                                   (optimizer_rgb_crf, args.rgb_crf_lrate, args.decay_rate_rgb_crf),
                                   (optimizer_event_crf, args.event_crf_lrate, args.decay_rate_event_crf)):
                new_lrate = lr0 * (rate ** (self.global_step / decay_steps))
                for param_group in opt.param_groups:
                    param_group["lr"] = new_lrate
            logger.update_buffer()
            tl = lap("optimizer steps + lr updates", tl)
            self.global_step += 1
            self.i += 1
            self.last_loss = loss
        return time.perf_counter() - t0


def synthetic_scene(wl, cam, n_events, seed):
    """Events dict (x, y, ts ascending, pol in {-1, +1}) and one blurry image [1, H, W, C] as numpy, like load_data hands them over."""
    rng = np.random.default_rng(seed)
    events = {"x": rng.integers(0, cam["W"], n_events).astype(np.int64), "y": rng.integers(0, cam["H"], n_events).astype(np.int64),
              "ts": np.sort(rng.random(n_events)), "pol": (rng.integers(0, 2, n_events) * 2 - 1).astype(np.float64)}
    img = rng.random((1, cam["H"], cam["W"], wl["channels"])).astype(np.float32)
    return events, img


def run(workload="C2", steps=20, warmup=5, seed=0, n_events=2_000_000, timers=False, over=None, host_times=False):
    """Times `steps` iterations of the loop after `warmup`.  Returns a dict (ms_per_step, rays_per_s, final loss, ...)."""
    from benerf_amd import workloads as WL, kernels as K
    wl = dict(WL.WORKLOADS[workload])
    cam = WL.CAMERAS[wl["cam"]]
    args = WL.make_args(wl, **(over or {}))
    events, img = synthetic_scene(wl, cam, n_events, seed)
    cuda_default_tensor_type(True)
    try:
        drv = Driver(args, cam, events, img, seed)
        drv.iterate(warmup)
        torch.cuda.synchronize()
        # Warm-up includes the interpreter's first FULL garbage collection: CPython runs one when the young generations have overflowed
        # often enough (here: around iteration 45), and a full pass walks every tracked object of the process - with torch imported
        # ~80-130 ms (profiles/r06_dropin_gc_outlier.log).  After it, full collections need the long-lived heap to grow by a quarter,
        # which a training loop in steady state never does: a one-off of any long run, not a per-iteration cost - and not one that a
        # 20-50-iteration sample should carry as if it recurred.
        import gc
        gc.collect()
        if timers:
            K.TIMERS.records.clear()
            K.TIMERS.enabled = True
        if host_times:
            drv.host_times = {}
        marks = []
        t0 = time.perf_counter()
        for _ in range(steps):
            drv.iterate(1)
            marks.append(time.perf_counter())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        K.TIMERS.enabled = False
        series = [b - a for a, b in zip([t0] + marks[:-1], marks)]
        per = sorted(series)
        rays = WL.rays_per_step(wl)
        out = {"ms_per_step": round(dt / steps * 1e3, 3), "median_ms_per_step": round(per[len(per) // 2] * 1e3, 3),
               "rays_per_s": round(rays * steps / dt, 1), "rays_per_step": rays, "steps": steps, "warmup": warmup,
               "max_ms_per_step": round(per[-1] * 1e3, 3),
               "slowest_steps": [(i, round(t * 1e3, 2)) for i, t in sorted(enumerate(series), key=lambda kv: -kv[1])[:4]],
               "final_loss": float(drv.last_loss.item()), "host_syncs_per_step": "6 x .item() (logger.write) + 1 image upload (torch.Tensor(img[0]))"}
        if host_times:
            out["host_ms_per_step"] = {k: round(v / steps * 1e3, 3) for k, v in drv.host_times.items()}
        if timers:
            out["per_kernel"] = {k: {"launches_per_step": round(n / steps, 2), "avg_ms": round(ms / n, 4), "ms_per_step": round(ms / steps, 3)}
                                 for k, (n, ms, _) in K.TIMERS.summary().items()}
            K.TIMERS.records.clear()
        return out
    finally:
        cuda_default_tensor_type(False)


if __name__ == "__main__":
    import argparse
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--timers", action="store_true")
    ap.add_argument("--host-times", action="store_true", help="host wall time per section of the loop (perf_counter; includes waits)")
    ap.add_argument("--mlp-precision", default="split")
    a = ap.parse_args()
    from benerf_amd import kernels as K_
    K_.set_mlp_precision(a.mlp_precision)
    print(json.dumps(run(a.workload, a.steps, a.warmup, timers=a.timers, host_times=a.host_times)))

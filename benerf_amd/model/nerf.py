"""Host-side mirror of the reference's model/nerf.py: same classes, method names, argument
order and return structures; all arithmetic runs in the HIP kernels (benerf_amd.engine).

`from model.nerf import *` in a reference-style driver relies on np / torch / nn / F / os being
re-exported from here (train.py:6,66), so this module defines no __all__.
"""
import os  # noqa: F401  (re-exported, see module docstring)
import abc
from datetime import datetime

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401
from torch import nn as nn

from .. import engine
from .. import kernels as K
from ..engine import Camera, Draws
from . import embedder  # noqa: F401


class Model:
    """Abstract trainer facade (model/nerf.py:28-38)."""

    @abc.abstractmethod
    def build_network(self, args, poses=None, event_poses=None):
        pass

    @abc.abstractmethod
    def setup_optimizer(self, args):
        pass

    def after_train(self):
        print(f"Successfully finished model on {datetime.now()}")


def _draw(fn, shape, device, scale=None, out=None):
    """One draw from the global torch generator, reference order/shape.  out: a contiguous float32 tensor of that
    shape to draw into (a row range of a batch buffer) - the same values as a fresh tensor gets."""
    if out is not None:
        t = fn(shape, out=out)
        if t is not out:            # a stand-in for torch.rand / randn that ignores `out` (tests replaying recorded draws)
            out.copy_(t)
        if scale is not None and scale != 1.0:
            out.mul_(scale)
        return out
    t = fn(shape, device=device)
    if scale is not None and scale != 1.0:
        t = t * scale
    return t.to(device=device, dtype=torch.float32).contiguous()


class PoseMajorRows(torch.Tensor):
    """A rendered colour map [P * R, C] (pose-major: R pixels for each of P poses) that answers the slices the reference's loop
    takes out of it - whole per-pose row blocks `[j * R:(j + 1) * R]` - with the render node's own per-block outputs
    (engine.RenderPair) instead of a generic autograd slice.  Same values, same gradients (tests/test_dropin_gpu.py); what changes
    is the backward cost: no zero-fill + copy + accumulate launches per slice.  Every other operation - any other index, any torch
    function - sees a plain tensor and returns plain tensors."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def __getitem__(self, idx):
        blocks = getattr(self, "_pose_blocks", None)
        if blocks is not None and type(idx) is slice and idx.step in (None, 1):
            R = blocks[0].shape[0]
            a = 0 if idx.start is None else idx.start
            b = self.shape[0] if idx.stop is None else idx.stop
            if type(a) is int and type(b) is int and a >= 0 and b - a == R and a % R == 0 and a // R < len(blocks) and torch.is_grad_enabled():
                return blocks[a // R]
        return super().__getitem__(idx)


def _pose_major(t, blocks):
    rows = t.as_subclass(PoseMajorRows)
    rows._pose_blocks = tuple(blocks)
    return rows


_FRONT_STREAMS = {}


class _FrontStream:
    """A second HIP stream for the parameter-independent head of Graph.forward (see there).  Tensors made inside are registered
    with keep(): close() tells the caching allocator that the main stream uses them too (record_stream) and makes the main
    stream wait for the front's work."""

    def __init__(self, graph, dev):
        self.on = dev.type == "cuda" and os.environ.get("BENERF_FRONT_STREAM", "1") != "0"
        self.open = False
        self.made = []
        if self.on:
            side = _FRONT_STREAMS.get(dev)      # per device, not on the module: a graph stays picklable / deep-copyable
            if side is None:
                side = _FRONT_STREAMS[dev] = torch.cuda.Stream(device=dev)
            self.side = side
            self.main = torch.cuda.current_stream(dev)

    def __enter__(self):
        if self.on:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
            self.open = True
        return self

    def keep(self, t):
        if self.open and t is not None:
            self.made.append(t)
        return t

    def close(self):
        if self.open:
            self.open = False
            self.ctx.__exit__(None, None, None)
            for t in self.made:
                t.record_stream(self.main)
            self.made = []
            self.main.wait_stream(self.side)


class _MlpPoints(torch.autograd.Function):
    """NeRF.forward on explicit points: pts [N,S,3], viewdirs [N,3] -> raw [N,S,C+1]  (K3)."""

    @staticmethod
    def forward(ctx, pts, viewdirs, net, grad_enabled, *params):
        N, S = pts.shape[0], pts.shape[1]
        M = N * S
        dev = pts.device
        pts_c = pts.detach().reshape(M, 3).float().contiguous()
        vd = viewdirs.detach().float()[:, None].expand(N, S, 3).reshape(M, 3).contiguous()
        need = any(ctx.needs_input_grad) and grad_enabled      # needs_input_grad ignores torch.no_grad(): the caller's grad mode is an argument
        net.pack_if_stale()
        # pts = 0 + pts * 1: feed each point as a one-sample ray
        raw, acts = K.mlp_fwd(net, torch.zeros((M, 3), device=dev), pts_c, vd, torch.ones((M, 1), device=dev), need)
        ctx.net, ctx.acts, ctx.shape = net, acts, (N, S)
        return raw.view(N, S, -1)

    @staticmethod
    def backward(ctx, d_raw):
        net, (N, S) = ctx.net, ctx.shape
        gw = [torch.empty_like(w) for w in net.weights]
        gb = [torch.empty_like(b) for b in net.biases]
        d_pts, d_vd = K.mlp_bwd(net, d_raw.contiguous().view(N * S, -1), ctx.acts, N * S, 1, gw, gb, False)
        ctx.acts = None
        return (d_pts.view(N, S, 3), d_vd.view(N, S, 3).sum(1), None, None) + tuple(gw) + tuple(gb)


class _Composite(torch.autograd.Function):
    """NeRF.raw2output (K4).  Differentiable w.r.t. raw and rays_d (through ||rays_d||)."""

    @staticmethod
    def forward(ctx, raw, z, rays_d, noise):
        raw_c, z_c, rd_c = raw.detach().float().contiguous(), z.detach().float().contiguous(), rays_d.detach().float().contiguous()
        out = K.composite_fwd(raw_c, z_c, rd_c, noise)
        ctx.save_for_backward(raw_c, z_c, rd_c, noise)
        ctx.mark_non_differentiable(out["weights"], out["sigma"])
        return out["rgb_map"], out["disp"], out["acc"], out["weights"], out["depth"], out["sigma"]

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth, g_sigma):
        raw, z, rd, noise = ctx.saved_tensors

        def c(t):
            return None if t is None else t.contiguous()

        if g_rgb is None:
            g_rgb = torch.zeros((raw.shape[0], raw.shape[2] - 1), device=raw.device)
        d_raw, d_rd = K.composite_bwd(raw, z, rd, noise, 0.0, 0, 0, c(g_rgb), c(g_acc), c(g_depth), c(g_disp))
        return d_raw, None, d_rd, None


def barf_weights(iter_step, args, device):
    """None, or the BARF coarse-to-fine column weights of this iteration (barf_c2f_weight, model/nerf.py:16-26)."""
    if not getattr(args, "use_barf_c2f", False):
        return None
    return K.barf_pe_weights(iter_step, args.max_iter, args.barf_c2f_start, args.barf_c2f_end, device)


class NeRF(nn.Module):
    """8x256 ReLU MLP with skip at layer 4 and a 128-wide view branch (model/nerf.py:40-64).
    Same constructor and parameter names as the reference; the HIP kernels implement exactly
    the architecture the reference hard-codes at model/optimize.py:9."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=False,
                 channels=3):
        super().__init__()
        if not (D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27 and list(skips) == [4] and use_viewdirs):
            raise NotImplementedError("benerf_amd implements the BeNeRF network shape only: D=8, W=256, input_ch=63, "
                                      "input_ch_views=27, skips=[4], use_viewdirs=True (model/optimize.py:9)")
        if channels not in (1, 3):
            raise NotImplementedError("channels must be 1 or 3")
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.channels = skips, use_viewdirs, channels
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W + input_ch, W) if i in skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, channels)
        self._packed = None

    # ---- kernel-side view of the parameters ----------------------------------------------------------
    def param_lists(self):
        return engine.nerf_param_lists(self)

    def packed(self):
        ws, bs = self.param_lists()
        if self._packed is None or any(a is not b for a, b in zip(self._packed.weights, ws)):
            self._packed = K.PackedMlp(ws, bs, self.channels)
        return self._packed

    def forward(self, iter_step, pts, viewdirs, args):
        """pts [N,S,3], viewdirs [N,3] -> [N,S,channels+1] = [rgb..., sigma]  (model/nerf.py:67-116)."""
        if viewdirs is None:
            raise NotImplementedError("use_viewdirs=False is not supported")
        net = self.packed()
        net.pe_weights = barf_weights(iter_step, args, pts.device)
        return _MlpPoints.apply(pts, viewdirs, net, torch.is_grad_enabled(), *net.weights, *net.biases)

    def raw2output(self, crf_func, enable_crf: bool, sensor_type, raw, z_vals, rays_d, raw_noise_std=1.0):
        """(model/nerf.py:118-148); crf_func / enable_crf / sensor_type are accepted and unused,
        exactly as in the reference where the CRF call is commented out."""
        noise = None
        if raw_noise_std > 0.:
            noise = _draw(torch.randn, raw[..., self.channels].shape, raw.device, raw_noise_std)
        return _Composite.apply(raw, z_vals, rays_d, noise)


class Graph(nn.Module):
    """Scene graph: coarse + fine NeRF, render() and the training forward (model/nerf.py:150-398)."""

    def __init__(self, args, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.nerf = NeRF(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, args.channels)
        self.channels = args.channels
        if args.N_importance > 0:
            self.nerf_fine = NeRF(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, args.channels)
        self.pose_eye = torch.eye(3, 4)
        self._event_cache = None

    # ---- events -----------------------------------------------------------------------------------------
    def _device(self):
        return self.nerf.alpha_linear.weight.device

    def _events_on_device(self, events, tum_vie=False):
        """Upload the event stream once (sorted by ts as recorded) instead of masking it on the
        host every iteration (model/nerf.py:170-178)."""
        key = (id(events["ts"]), len(events["ts"]))
        if self._event_cache is None or self._event_cache[0] != key:
            dev = self._device()
            ts = np.asarray(events["ts"], dtype=np.float64)
            if ts.size > 1 and not bool(np.all(ts[1:] >= ts[:-1])):
                raise ValueError("events['ts'] must be ascending")
            pol = np.asarray(events["pol"]).astype(np.float32)
            if tum_vie:
                pol = np.where(pol == 0, np.float32(-1), pol)      # 0: negative polarity in TUM-VIE (model/nerf.py:194-196)
            cache = {"x": torch.as_tensor(np.asarray(events["x"]).astype(np.int32), device=dev),
                     "y": torch.as_tensor(np.asarray(events["y"]).astype(np.int32), device=dev),
                     "p": torch.as_tensor(pol, device=dev),
                     "ts": torch.as_tensor(ts, device=dev)}
            self._event_cache = (key, cache)
        return self._event_cache[1]

    def _device_lut(self, remap, H, W):
        """TUM_VIE undistortion table [H, W, 2] as a float32 device tensor, uploaded once per source array: the training loop
        hands the same host table to every render() (twice per iteration), inference to every chunk."""
        dev = self._device()
        if isinstance(remap, torch.Tensor) and remap.is_cuda and remap.dtype == torch.float32:
            return remap.reshape(H, W, 2).contiguous()
        cache = getattr(self, "_lut_cache", None)
        if cache is None:
            cache = self._lut_cache = {}
        key = (id(remap), H, W)
        hit = cache.get(key)
        if hit is None or hit[0] is not remap:
            if len(cache) >= 4:
                cache.clear()
            hit = cache[key] = (remap, torch.as_tensor(remap, dtype=torch.float32, device=dev).reshape(H, W, 2).contiguous())
        return hit[1]

    def forward(self, iter_step, events, rgb_exp_ts, H, W, K, K_event, args, img_xy_remap, evt_xy_remap):
        """One training iteration's rendering (model/nerf.py:160-234): event-window accumulation,
        two trajectory queries, two renders.  Same return tuple as the reference.
        Extension (BASELINE.json configs[4], "dense event bins"; the reference has one bin): with args.event_bins = B > 1 the
        selected window is cut into B contiguous equal bins - events_accu becomes [B, H_e, W_e] (bin b: K7 over [t_b, t_b+1]),
        the event batch is rendered at the B + 1 bin boundaries (get_pose_evt(args, ts, seg_num=B + 1)) and ret_event holds
        (B + 1) * sampling_event_rays rows, pose-major: bin b's start / end colours are rows [b R, (b + 1) R) / [(b + 1) R, (b + 2) R)."""
        dev = self._device()
        bins = int(getattr(args, "event_bins", 1))
        if bins > 1 and not args.event_time_window:
            raise NotImplementedError("event_bins > 1 cuts a TIME window into bins: needs args.event_time_window")
        ev = self._events_on_device(events, args.dataset == "TUM_VIE")
        He, We = args.event_height, args.event_width
        # Everything of an iteration that depends on no parameter - the event-window accumulation (K7), the pixel draws (torch's
        # randperm: ~18 sort launches each, 0.2 ms of device time), the eight sampling draws - runs on a second stream: when the host
        # gets here the device is still busy with the PREVIOUS iteration's backward on the main stream (loss.backward() and the
        # optimiser steps only queue work), so these ~80 small launches execute beside it instead of in front of the first MLP launch.
        # Random values are fixed by the host-side call order, not by the stream.  BENERF_FRONT_STREAM=0 keeps them on the main stream.
        front = _FrontStream(self, dev)
        front.__enter__()
        try:
            out = self._forward_body(front, iter_step, ev, events, rgb_exp_ts, H, W, K, K_event, args, img_xy_remap, evt_xy_remap, bins, He, We)
        finally:
            front.close()
        return out

    def _forward_body(self, front, iter_step, ev, events, rgb_exp_ts, H, W, K, K_event, args, img_xy_remap, evt_xy_remap, bins, He, We):
        dev = self._device()
        if args.event_time_window:
            window_t = args.accumulate_time_length
            if args.random_sampling_window:
                low_t = np.random.rand(1) * (1 - window_t)
                upper_t = low_t + window_t
            else:
                low_t = np.random.randint((1 - window_t) // window_t) * window_t
                upper_t = np.min((low_t + window_t, 1.0))
            lo, up = float(np.asarray(low_t).reshape(-1)[0]), float(np.asarray(upper_t).reshape(-1)[0])
            if bins == 1:
                accu = K_.event_window_accumulate(ev["x"], ev["y"], ev["p"], ev["ts"], lo, up, He, We)
            else:   # bin boundaries = the float32 linspace the trajectory kernel evaluates the B + 1 event poses at; interior bins half-open
                accu = torch.stack([K_.event_window_accumulate(ev["x"], ev["y"], ev["p"], ev["ts"], lo_b, up_b, He, We)
                                    for lo_b, up_b in K_.event_bin_windows(lo, up, bins)])
            events_ts = np.stack((low_t, upper_t)).reshape(2)
        else:
            num = len(events["pol"])
            N_window = round(num * args.accumulate_time_length)
            if args.random_sampling_window:
                lo_i = np.random.randint(num - N_window)
            else:
                lo_i = np.random.randint((num - N_window) // N_window) * N_window
            hi_i = int(lo_i + N_window)
            accu = K_.event_accumulate(ev["x"][lo_i:hi_i], ev["y"][lo_i:hi_i], ev["p"][lo_i:hi_i], He, We)
            ts_np = np.asarray(events["ts"])
            events_ts = ts_np[lo_i:hi_i][np.array([0, int(N_window) - 1])]
        events_accu = front.keep(accu.double())    # the reference returns float64 (utils/event_utils.py:256-257)

        # The reference queries the two trajectories and renders the two batches one after the other (model/nerf.py:209-232); here
        # both trajectory evaluations are one launch (K1) and both renders one batched launch sequence behind ONE autograd node
        # (engine.RenderPair).  Nothing in front of it synchronises the host: the window times go to the device by value, the
        # camera matrices are read on the host (the reference's torch.Tensor(K_event) would be a device tensor under its cuda
        # default tensor type, each K[i][j] read a round trip).  Random draws: the reference's order and shapes from the global
        # torch generator - randperm (event pixels), four draws of the event render, randperm (blur pixels), four draws of the
        # blur render.
        ts_e = front.keep(self._ts_on_device(events_ts, dev))
        ts_r = front.keep(self._ts_on_device(rgb_exp_ts, dev))
        Pe, Pr = (2 if bins == 1 else bins + 1), args.num_interpolated_pose
        if not self._stock_queries() or not hasattr(self, "transform") or getattr(self, "nerf_fine", None) is None or not args.use_viewdirs:
            # a graph whose get_pose_* / render were overridden (subclass or instance), one without the optimize.py members or
            # without a fine network: the reference's own sequence of four calls
            front.close()
            spline_evt_poses = self.get_pose_evt(args, ts_e, seg_num=None if bins == 1 else bins + 1)
            spline_rgb_poses = self.get_pose_rgb(args, ts_r)
            ray_idx_event = torch.randperm(He * We, device=dev)[:args.sampling_event_rays]
            ret_event = self.render(iter_step, spline_evt_poses, ray_idx_event, He, We, K_event, args, enable_crf=True, sensor_type="event",
                                    remap=evt_xy_remap, training=True)
            ray_idx_rgb = torch.randperm(H * W, device=dev)[:args.sampling_rgb_rays // args.num_interpolated_pose]
            ret_rgb = self.render(iter_step, spline_rgb_poses, ray_idx_rgb, H, W, K, args, enable_crf=True, sensor_type="rgb",
                                  remap=img_xy_remap, training=True)
            return ret_event, ret_rgb, ray_idx_event, ray_idx_rgb, events_accu
        tum = args.dataset == "TUM_VIE"
        cam_e = Camera.from_K(He, We, K_event, self._device_lut(evt_xy_remap, He, We) if tum else None)
        cam_r = Camera.from_K(H, W, K, self._device_lut(img_xy_remap, H, W) if tum else None)
        S, Ni = args.N_samples, args.N_importance
        std = float(getattr(args, "benerf_raw_noise_std", engine.NOISE_STD_DEFAULT))
        n_rgb_pix = args.sampling_rgb_rays // args.num_interpolated_pose
        Ne, Nr = Pe * args.sampling_event_rays, Pr * n_rgb_pix
        N = Ne + Nr
        philox = getattr(args, "benerf_rng", "torch") == "philox"
        ray_idx_event = front.keep(torch.randperm(He * We, device=dev))[:args.sampling_event_rays]
        if philox:
            ray_idx_rgb = front.keep(torch.randperm(H * W, device=dev))[:n_rgb_pix]
            self._philox_calls = getattr(self, "_philox_calls", 0) + 2
            draws = Draws(seed=int(getattr(args, "benerf_seed", 0)), offset=self._philox_calls, noise_std=std)
        else:
            f32 = dict(dtype=torch.float32, device=dev)
            t_rand, u = front.keep(torch.empty((N, S), **f32)), front.keep(torch.empty((N, Ni), **f32))
            noise0 = front.keep(torch.empty((N, S), **f32)) if std > 0 else None
            noise1 = front.keep(torch.empty((N, S + Ni), **f32)) if std > 0 else None
            for a, b in ((0, Ne), (Ne, N)):
                if a:
                    ray_idx_rgb = front.keep(torch.randperm(H * W, device=dev))[:n_rgb_pix]
                _draw(torch.rand, (b - a, S), dev, out=t_rand[a:b])
                if std > 0:
                    _draw(torch.randn, (b - a, S), dev, std, out=noise0[a:b])
                _draw(torch.rand, [b - a, Ni], dev, out=u[a:b])
                if std > 0:
                    _draw(torch.randn, (b - a, S + Ni), dev, std, out=noise1[a:b])
            draws = Draws(t_rand, noise0, u, noise1, noise_std=0.0 if std <= 0 else std)
        ray_idx_event, ray_idx_rgb = front.keep(ray_idx_event.contiguous()), front.keep(ray_idx_rgb.contiguous())
        front.close()          # back on the main stream, which now waits for the front
        traj = {"spline": 0, "linear": 1}[args.traj]
        spline_evt_poses, spline_rgb_poses = engine.SplinePosesPair.apply(self.evt_knot_pose_se3.params.weight, self.transform.params.weight,
                                                                        ts_e, ts_r, Pe, Pr, traj)
        net_c, net_f = self.nerf.packed(), self.nerf_fine.packed()
        net_c.pe_weights = net_f.pe_weights = barf_weights(iter_step, args, dev)
        draws.grad_enabled = torch.is_grad_enabled()
        chunks = os.environ.get("BENERF_POSE_BLOCKS", "1") != "0" and draws.grad_enabled
        outs = engine.RenderPair.apply(spline_evt_poses, spline_rgb_poses, ray_idx_event, ray_idx_rgb, cam_e, cam_r,
                                       bool(args.ndc), S, Ni, draws, net_c, net_f, chunks, *net_c.weights, *net_c.biases, *net_f.weights, *net_f.biases)
        keys = ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "sigma")
        ret_event, ret_rgb = dict(zip(keys, outs[:7])), dict(zip(keys, outs[7:14]))
        if chunks and len(outs) > 14:
            o = 14
            for ret, P in ((ret_event, Pe), (ret_rgb, Pr)):
                for key in ("rgb_map", "rgb0"):
                    ret[key] = _pose_major(ret[key], outs[o:o + P])
                    o += P
        return ret_event, ret_rgb, ray_idx_event, ray_idx_rgb, events_accu

    def _stock_queries(self):
        """True if get_pose_evt / get_pose_rgb / render are the stock implementations (model/optimize.py's Graph, this class):
        only then may Graph.forward replace the four calls by the fused pair of nodes."""
        stock = getattr(type(self), "_stock_pose_queries", None)
        if stock is None or any(n in self.__dict__ for n in ("get_pose_evt", "get_pose_rgb", "render")):
            return False
        return all(getattr(type(self), n, None) is f for n, f in stock.items()) and type(self).render is Graph.render

    def _ts_on_device(self, ts, dev):
        """[t0, t1] as a float32 device tensor WITHOUT a host-to-device copy (which, from pageable memory, waits for everything
        queued on the stream - i.e. for the previous iteration's backward): two fills by value."""
        if torch.is_tensor(ts) and ts.is_cuda:
            return ts.reshape(-1)[:2].to(torch.float32)
        v = np.asarray(ts.detach().cpu() if torch.is_tensor(ts) else ts, dtype=np.float64).reshape(-1)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        out[0:1].fill_(float(np.float32(v[0])))
        out[1:2].fill_(float(np.float32(v[1])))
        return out

    # ---- render -------------------------------------------------------------------------------------------
    def render(self, iter_step, poses, ray_idx, H, W, K, args, enable_crf: bool, sensor_type: str,
               remap: torch.Tensor, near=0., far=1., training=False):
        """rays (pose-major) -> stratified coarse pass -> importance sampling -> fine pass
        (model/nerf.py:236-343).  `training` only selects how the reference builds its rays; both
        branches give the same rays, generated on the fly here."""
        if not args.use_viewdirs:
            # the reference cannot run this combination either: model/optimize.py:9 builds both networks with use_viewdirs=True and
            # NeRF.forward (model/nerf.py:93) then splits the 63-wide encoding into 63 + 27 columns -> RuntimeError
            raise NotImplementedError("use_viewdirs=False: the reference's own NeRF.forward raises for it (INTEGRATION.md)")
        dev = self._device()
        poses = poses[:, :3, :4]
        lut = None
        if args.dataset == "TUM_VIE":      # rect = remap[j, i] (model/nerf.py:247-250; run_nerf_helpers.py:17-23 for inference)
            lut = self._device_lut(remap, H, W)
        cam = Camera.from_K(H, W, K, lut)
        ray_idx = ray_idx.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        N = poses.shape[0] * ray_idx.shape[0]
        S, Ni = args.N_samples, args.N_importance
        std = float(getattr(args, "benerf_raw_noise_std", engine.NOISE_STD_DEFAULT))
        if getattr(args, "benerf_rng", "torch") == "philox":
            self._philox_calls = getattr(self, "_philox_calls", 0) + 1
            draws = Draws(seed=int(getattr(args, "benerf_seed", 0)), offset=self._philox_calls, noise_std=std, near=near, far=far)
        else:   # reference behaviour: four draws from the global torch generator, in its order
            t_rand = _draw(torch.rand, (N, S), dev)
            noise0 = _draw(torch.randn, (N, S), dev, std) if std > 0 else None
            u = _draw(torch.rand, [N, Ni], dev) if Ni > 0 else None
            noise1 = _draw(torch.randn, (N, S + Ni), dev, std) if (std > 0 and Ni > 0) else None
            draws = Draws(t_rand, noise0, u, noise1, noise_std=0.0 if std <= 0 else std, near=near, far=far)
        net_c = self.nerf.packed()
        net_f = self.nerf_fine.packed() if Ni > 0 else None
        net_c.pe_weights = barf_weights(iter_step, args, dev)
        if net_f is not None:
            net_f.pe_weights = net_c.pe_weights
        params = list(net_c.weights) + list(net_c.biases)
        if net_f is not None:
            params += list(net_f.weights) + list(net_f.biases)
        draws.grad_enabled = torch.is_grad_enabled()      # inside an autograd Function's forward grad mode is always off
        outs = engine.RenderRays.apply(poses.float(), ray_idx, cam, bool(args.ndc), S, Ni, draws, net_c, net_f, *params)
        if Ni > 0:
            keys = ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "sigma")
        else:
            keys = ("rgb_map", "disp_map", "acc_map", "sigma")
        ret = dict(zip(keys, outs))
        if Ni == 0:
            ret.pop("sigma")
        return ret

    def camera_response_func(self, radience, sensor_type):
        if sensor_type == "rgb":
            return self.rgb_crf.forward(radience)
        elif sensor_type == "event":
            return self.event_crf.forward(radience)

    @torch.no_grad()
    def render_video(self, iter_step, poses, H, W, K, args, remap, type):
        """Chunked full-image inference (model/nerf.py:353-390): dict of [H,W,...] tensors."""
        all_ret = {}
        dev = self._device()
        ray_idx = torch.arange(0, H * W, device=dev)
        render_type = str(type)
        for i in range(0, ray_idx.shape[0], args.chunk):
            if render_type == "radience":
                ret = self.render(iter_step, poses, ray_idx[i:i + args.chunk], H, W, K, args, enable_crf=False,
                                  sensor_type=None, remap=remap, training=False)
            elif render_type == "rgb":
                ret = self.render(iter_step, poses, ray_idx[i:i + args.chunk], H, W, K, args, enable_crf=True,
                                  sensor_type="rgb", remap=remap, training=False)
            else:
                raise ValueError("render_video: type must be 'rgb' or 'radience'")
            for k in ret:
                all_ret.setdefault(k, []).append(ret[k])
        for k in all_ret:
            all_ret[k] = torch.cat(all_ret[k], 0).reshape([H, W] + list(all_ret[k][0].shape[1:]))
        return all_ret

    @abc.abstractmethod
    def get_pose(self, args, events_ts):
        pass

    @abc.abstractmethod
    def get_pose_rgb(self, args, seg_num=None):
        pass


K_ = K   # render()/forward() take the camera matrix as an argument named K, like the reference

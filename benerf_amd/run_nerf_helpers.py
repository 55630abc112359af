"""Host-side mirror of the reference's run_nerf_helpers.py public functions (same names,
argument order, return structures).  Inside `Graph.render` these operators are fused into the
ray / sampling kernels; the stand-alone forms here call single-operator HIP kernels
(forward only - gradients flow through the fused path)."""
import os

import numpy as np
import torch
from tqdm import tqdm

from . import kernels as K
from .utils import img_utils


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _f(x):
    return float(x)


def get_rays(H, W, K_, c2w, args, remap):
    """Full H x W ray grid for one pose (run_nerf_helpers.py:13-32): rays_o, rays_d [H,W,3]."""
    dev = c2w.device if c2w.is_cuda else _dev()
    idx = torch.arange(H * W, device=dev)
    if getattr(args, "dataset", None) == "TUM_VIE":      # rect = remap[j, i] (run_nerf_helpers.py:17-23)
        lut = torch.as_tensor(remap, dtype=torch.float32, device=dev).reshape(H, W, 2).contiguous()
        pose1 = c2w[:3, :4].detach().float().to(dev).reshape(1, 3, 4).contiguous()
        ro, rd, _ = K.rays_fwd(pose1, idx, H, W, _f(K_[0][0]), _f(K_[1][1]), _f(K_[0][2]), _f(K_[1][2]), ndc=False, remap=lut)
        return ro.view(H, W, 3), rd.view(H, W, 3)
    i, j = (idx % W).contiguous(), (idx // W).contiguous()
    pose = c2w[:3, :4].detach().float().to(dev).contiguous()
    ro, rd = K.pixel_rays(pose, i, j, _f(K_[0][0]), _f(K_[1][1]), _f(K_[0][2]), _f(K_[1][2]))
    return ro.view(H, W, 3), rd.view(H, W, 3)


def get_specific_rays(i, j, K_, c2w):
    """Rays for pixel columns i / rows j with one pose per ray c2w [N,3,4]
    (run_nerf_helpers.py:35-44)."""
    dev = c2w.device if c2w.is_cuda else _dev()
    i = i.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
    j = j.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
    poses = c2w[..., :3, :4].detach().float().to(dev).contiguous()
    if poses.dim() == 2:
        poses = poses[None]
    if poses.shape[0] == 1 and i.numel() > 1:
        poses = poses.expand(i.numel(), 3, 4).contiguous()
    return K.pixel_rays(poses, i, j, _f(K_[0][0]), _f(K_[1][1]), _f(K_[0][2]), _f(K_[1][2]))


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """LLFF NDC projection (run_nerf_helpers.py:46-71)."""
    shp = rays_o.shape
    oo, od = K.ndc_rays(int(H), int(W), _f(focal), _f(near), rays_o.detach().float().reshape(-1, 3).contiguous(),
                        rays_d.detach().float().reshape(-1, 3).contiguous())
    return oo.view(shp), od.view(shp)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF sampling (run_nerf_helpers.py:74-115).  Draws u from the global torch generator
    like the reference (or linspace when det)."""
    dev = bins.device
    lead = list(bins.shape[:-1])
    if det:
        u = torch.linspace(0.0, 1.0, steps=N_samples, device=dev).expand(lead + [N_samples])
    else:
        u = torch.rand(lead + [N_samples], device=dev)
    if pytest:
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0.0, 1.0, N_samples), lead + [N_samples])
        else:
            u = np.random.rand(*(lead + [N_samples]))
        u = torch.Tensor(u)
    u = u.to(device=dev, dtype=torch.float32).contiguous()
    b2 = bins.detach().float().reshape(-1, bins.shape[-1]).contiguous()
    w2 = weights.detach().float().reshape(-1, weights.shape[-1]).contiguous()
    s = K.sample_pdf(b2, w2, N_samples, u=u.reshape(-1, N_samples))
    return s.view(lead + [N_samples])


def _device_frames(iter_step, graph, render_poses, H, W, K_, args, remap):
    """One full-image render per pose (Graph.render_video, chunked K2-K5 launches), tone-mapped when the colour CRF is
    trained; yields (index, rgb [H,W,C], disparity [H,W]) as DEVICE tensors - the callers convert once, on the device."""
    for index, pose in enumerate(tqdm(render_poses)):
        out = graph.render_video(iter_step, pose[None, :3, :4], H, W, K_, args, remap, type="rgb")
        rgb = graph.rgb_crf.forward(out["rgb_map"]) if args.optimize_rgb_crf else out["rgb_map"]
        yield index, rgb, out["disp_map"]


def _quantise(t):
    """utils/img_utils.to8bit on the device: 255 * clip(x, 0, 1), truncated to uint8."""
    return (t.clamp(0.0, 1.0) * 255.0).to(torch.uint8)


@torch.no_grad()
def render_video_test(iter_step, graph, render_poses, H, W, K_, args, remap):
    """(run_nerf_helpers.py:117-140) -> (rgbs [n,H,W,C], disps [n,H,W]) float32 numpy; the frames are collected in two
    device buffers and cross PCIe once."""
    rgb_all = disp_all = None
    for k, rgb, disp in _device_frames(iter_step, graph, render_poses, H, W, K_, args, remap):
        if rgb_all is None:
            n = len(render_poses)
            rgb_all = torch.empty((n,) + tuple(rgb.shape), dtype=rgb.dtype, device=rgb.device)
            disp_all = torch.empty((n,) + tuple(disp.shape), dtype=disp.dtype, device=disp.device)
            print(rgb.shape, disp.shape)        # the reference reports the frame shapes once
        rgb_all[k], disp_all[k] = rgb, disp
    return rgb_all.cpu().numpy(), disp_all.cpu().numpy()


def _imwrite(path, img, mode):
    try:
        from imageio.v3 import imwrite
        imwrite(path, img, mode=mode)
    except ImportError:      # imageio is not in this image: keep the arrays, skip the PNG
        np.save(os.path.splitext(path)[0] + ".npy", img)


@torch.no_grad()
def render_image_test(iter_step, graph, render_poses, H, W, K_, args, logdir, remap, dir=None, need_depth=True):
    """(run_nerf_helpers.py:142-171) -> (imgs, depth) lists of uint8 arrays; PNGs under logdir/dir/img_test_{iter:06d}/
    with the reference's file names (frames: dir[11:] + index, depth maps: depth_ + index).  8-bit conversion and the
    per-frame disparity normalisation run on the device; only bytes are copied back."""
    target = os.path.join(logdir, dir, "img_test_{:06d}".format(iter_step))
    os.makedirs(target, exist_ok=True)
    colour_mode = "L" if args.channels == 1 else "RGB"
    imgs, depth = [], []
    for k, rgb, disp in _device_frames(iter_step, graph, render_poses, H, W, K_, args, remap):
        frame = _quantise(rgb).cpu().numpy()
        _imwrite(os.path.join(target, "{}{:03d}.png".format(dir[11:], k)), frame.squeeze(), colour_mode)
        imgs.append(frame)
        if need_depth:
            dmap = _quantise(disp / disp.max()).cpu().numpy()
            _imwrite(os.path.join(target, "depth_{:03d}.png".format(k)), dmap, "L")
            depth.append(dmap)
    return imgs, depth


def init_weights(linear):
    torch.nn.init.xavier_uniform_(linear.weight)
    torch.nn.init.zeros_(linear.bias)


def init_nerf(nerf):
    """Xavier-uniform weights / zero biases on every linear (run_nerf_helpers.py:194-208)."""
    for linear in list(nerf.pts_linears) + list(nerf.views_linears):
        init_weights(linear)
    init_weights(nerf.feature_linear)
    init_weights(nerf.alpha_linear)
    init_weights(nerf.rgb_linear)

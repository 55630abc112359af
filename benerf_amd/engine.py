"""Kernel sequencing for the BeNeRF hot path on one MI355X.

Two entry styles over the SAME HIP kernels (benerf_amd.kernels -> C ABI):

  * autograd Functions (`SplinePoses`, `RenderRays`): make the reference's
    `graph.render(...)` / `graph.get_pose_*` + `loss.backward()` + torch.optim flow work
    unchanged (model/nerf.py:236-343, model/optimize.py:58-111, train.py:340-352);
  * `TrainStep`: the lean fixed-shape training iteration (no autograd graph, flat parameter
    / gradient / Adam buffers, event + blur rays batched into one launch sequence, in-kernel
    Philox draws).  This is what bench.py measures and what data-parallel training uses
    (one process per GPU, one RCCL all-reduce of the flat gradient buffer per step).

Random draws follow the reference order per render: stratified jitter,
sigma noise (coarse), importance u, sigma noise (fine).
"""
import math
import os

import torch

from . import _lib
from . import dist
from . import kernels as K
from . import optim

NOISE_STD_DEFAULT = 1.0   # NeRF.raw2output default, never overridden by the reference (model/nerf.py:118)


class Camera:
    """Pinhole intrinsics (+ the optional TUM_VIE undistortion table [H, W, 2] float32 on the device)."""
    __slots__ = ("H", "W", "fx", "fy", "cx", "cy", "remap")

    def __init__(self, H, W, fx, fy, cx, cy, remap=None):
        self.H, self.W = int(H), int(W)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.remap = remap

    @staticmethod
    def from_K(H, W, Kmat, remap=None):
        return Camera(H, W, float(Kmat[0][0]), float(Kmat[1][1]), float(Kmat[0][2]), float(Kmat[1][2]), remap)


class Draws:
    """Random inputs of one render: explicit tensors (parity mode) or Philox (seed, offsets)."""

    def __init__(self, t_rand=None, noise0=None, u=None, noise1=None, seed=0, offset=0, noise_std=NOISE_STD_DEFAULT, near=0.0,
                 far=1.0):
        self.t_rand, self.noise0, self.u, self.noise1 = t_rand, noise0, u, noise1
        self.seed, self.offset, self.noise_std = int(seed), int(offset), float(noise_std)
        self.near, self.far = float(near), float(far)      # depth range of the stratified samples (model/nerf.py:297-299)

    def noise_args(self, which, part=0):
        """part: a render node that composites its batches in separate launches (RenderPair) draws each one's Philox noise from
        a stream of its own (the kernel keys the counter by the row inside the launch)."""
        t = self.noise0 if which == 0 else self.noise1
        if t is not None:
            return t, 0.0, 0, 0
        return None, self.noise_std, self.seed, (self.offset + part * (1 << 20)) * 4 + (1 if which == 0 else 3)

    def jitter_args(self):
        return self.t_rand, self.seed, self.offset * 4 + 0

    def u_args(self):
        return self.u, self.seed, self.offset * 4 + 2


# ------------------------------------------------------------------------------------------
# autograd glue
# ------------------------------------------------------------------------------------------
class SplinePoses(torch.autograd.Function):
    """(knots [4,6], transform [1,6] | None, ts) -> poses [P,3,4]   (K1)"""

    @staticmethod
    def forward(ctx, knots, transform, ts, n_poses, traj, explicit_ts):
        knots_c = knots.detach().contiguous()
        tr_c = None if transform is None else transform.detach().reshape(6).contiguous()
        ts_c = ts.detach().to(device=knots.device, dtype=torch.float32).contiguous()
        ctx.save_for_backward(knots_c, tr_c, ts_c)
        ctx.cfg = (n_poses, traj, explicit_ts, None if transform is None else transform.shape)
        return K.spline_poses_fwd(knots_c, tr_c, ts_c, n_poses, traj, explicit_ts)

    @staticmethod
    def backward(ctx, d_poses):
        knots, tr, ts = ctx.saved_tensors
        n_poses, traj, explicit_ts, tr_shape = ctx.cfg
        d_knots, d_tr = K.spline_poses_bwd(knots, tr, ts, n_poses, traj, d_poses.contiguous(), explicit_ts)
        if d_tr is not None:
            d_tr = d_tr.reshape(tr_shape)
        return d_knots, d_tr, None, None, None, None


class SplinePosesPair(torch.autograd.Function):
    """Both trajectory queries of one training iteration in one launch each way (K1): (knots [4,6], transform [1,6], ts_a [2],
    ts_b [2]) -> (poses_a [n_a,3,4] on the knots - get_pose_evt, model/optimize.py:58-82 -, poses_b [n_b,3,4] on knots +
    transform - get_pose_rgb, model/optimize.py:84-111)."""

    @staticmethod
    def forward(ctx, knots, transform, ts_a, ts_b, n_a, n_b, traj):
        knots_c, tr_c = knots.detach().contiguous(), transform.detach().reshape(6).contiguous()
        ta, tb = ts_a.detach().contiguous(), ts_b.detach().contiguous()
        ctx.save_for_backward(knots_c, tr_c, ta, tb)
        ctx.cfg = (n_a, n_b, traj, transform.shape)
        return K.spline_poses_fwd_pair(knots_c, tr_c, ta, n_a, tb, n_b, traj)

    @staticmethod
    def backward(ctx, d_a, d_b):
        knots, tr, ta, tb = ctx.saved_tensors
        n_a, n_b, traj, tr_shape = ctx.cfg
        z = lambda d, n: torch.zeros((n, 3, 4), dtype=torch.float32, device=knots.device) if d is None else d.contiguous()   # noqa: E731
        dk_a, dk_b, dt_b = K.spline_poses_bwd_pair(knots, tr, ta, n_a, tb, n_b, traj, z(d_a, n_a), z(d_b, n_b))
        return dk_a.add_(dk_b), dt_b.reshape(tr_shape), None, None, None, None, None


def _render_forward(cam, ndc, n_samples, n_importance, draws, poses, ray_idx, net_c, net_f, save, z_fine_forced=None):
    """Shared forward kernel sequence of Graph.render.  Returns (outputs dict, saved dict).
    z_fine_forced [N, S + Ni]: parity runs may hand in the merged fine depths of another evaluation instead of K5's
    (sample_pdf is ill-conditioned in the coarse weights: this isolates everything behind it)."""
    ro, rd, vd = K.rays_fwd(poses, ray_idx, cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, ndc, remap=cam.remap)
    n_rays = ro.shape[0]
    t_rand, seed, off = draws.jitter_args()
    z = K.stratified_z(n_rays, n_samples, ro.device, t_rand, seed, off, draws.near, draws.far)
    raw0, acts0 = K.mlp_fwd(net_c, ro, rd, vd, z, save)
    nz0 = draws.noise_args(0)
    want0 = ("rgb_map", "disp", "acc", "weights") if n_importance > 0 else ("rgb_map", "disp", "acc", "sigma")
    c0 = K.composite_fwd(raw0, z, rd, nz0[0], nz0[1], nz0[2], nz0[3], want=want0)
    out = {"rgb_map": c0["rgb_map"], "disp_map": c0["disp"], "acc_map": c0["acc"]}
    saved = {"ro": ro, "rd": rd, "vd": vd, "z": z, "raw0": raw0, "acts0": acts0}
    if n_importance > 0:
        u, useed, uoff = draws.u_args()
        z_fine = K.sample_pdf_merge(z, c0["weights"], n_importance, u, useed, uoff) if z_fine_forced is None else z_fine_forced.contiguous()
        raw1, acts1 = K.mlp_fwd(net_f, ro, rd, vd, z_fine, save)
        nz1 = draws.noise_args(1)
        c1 = K.composite_fwd(raw1, z_fine, rd, nz1[0], nz1[1], nz1[2], nz1[3], want=("rgb_map", "disp", "acc", "sigma"))
        out = {"rgb_map": c1["rgb_map"], "disp_map": c1["disp"], "acc_map": c1["acc"], "rgb0": c0["rgb_map"],
               "disp0": c0["disp"], "acc0": c0["acc"], "sigma": c1["sigma"]}
        saved.update({"z_fine": z_fine, "raw1": raw1, "acts1": acts1})
    else:
        out["sigma"] = c0["sigma"]
    return out, saved


def _render_backward(cam, ndc, draws, poses, ray_idx, net_c, net_f, saved, g, grads_c, grads_f, accumulate):
    """Backward kernel sequence.  g: dict of upstream grads (rgb_map, acc_map, disp_map, rgb0, acc0,
    disp0; missing = zero).  grads_c/f: (list_w, list_b) written (or accumulated into).
    Returns d_poses [P,3,4]."""
    ro, rd, vd, z = saved["ro"], saved["rd"], saved["vd"], saved["z"]
    n_rays = ro.shape[0]
    dev = ro.device
    d_o = torch.zeros((n_rays, 3), dtype=torch.float32, device=dev)
    d_d = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)   # first written by composite_bwd
    d_v = torch.zeros((n_rays, 3), dtype=torch.float32, device=dev)
    fine = "raw1" in saved
    acc_c, acc_f = accumulate if isinstance(accumulate, tuple) else (accumulate, accumulate)

    def zeros_like_rgb(t):
        return torch.zeros_like(t)

    first = True
    if fine:
        nz1 = draws.noise_args(1)
        g_rgb = g.get("rgb_map")
        if g_rgb is None:
            g_rgb = zeros_like_rgb(saved["raw1"][:, 0, :-1]).contiguous()
        d_raw1, _ = K.composite_bwd(saved["raw1"], saved["z_fine"], rd, nz1[0], nz1[1], nz1[2], nz1[3], g_rgb,
                                    g.get("acc_map"), None, g.get("disp_map"), d_rays_d=d_d, accumulate=False)
        d_pts, d_vp = K.mlp_bwd(net_f, d_raw1.reshape(-1, d_raw1.shape[-1]), saved["acts1"], n_rays,
                                saved["z_fine"].shape[1], grads_f[0], grads_f[1], acc_f)
        # d_d already holds the ||rays_d|| term of the fine compositing: accumulate on top of it
        K.ray_grad_reduce(saved["z_fine"], d_pts, d_vp, d_o, d_d, d_v, True)
        first = False
        g_rgb0, g_acc0, g_disp0 = g.get("rgb0"), g.get("acc0"), g.get("disp0")
    else:
        g_rgb0, g_acc0, g_disp0 = g.get("rgb_map"), g.get("acc_map"), g.get("disp_map")
    nz0 = draws.noise_args(0)
    if g_rgb0 is None:
        g_rgb0 = zeros_like_rgb(saved["raw0"][:, 0, :-1]).contiguous()
    d_raw0, _ = K.composite_bwd(saved["raw0"], z, rd, nz0[0], nz0[1], nz0[2], nz0[3], g_rgb0, g_acc0, None, g_disp0,
                                d_rays_d=d_d, accumulate=not first)
    d_pts, d_vp = K.mlp_bwd(net_c, d_raw0.reshape(-1, d_raw0.shape[-1]), saved["acts0"], n_rays, z.shape[1],
                            grads_c[0], grads_c[1], acc_c)
    K.ray_grad_reduce(z, d_pts, d_vp, d_o, d_d, d_v, True)
    return K.rays_bwd(poses, ray_idx, cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, ndc, d_o, d_d, d_v, remap=cam.remap)


def _pass_id():
    """Id of the running backward pass (autograd graph task), -1 outside one / on a torch without the query."""
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:      # pragma: no cover
        return -1


def _grad_targets(net):
    """Where the 24 weight / bias gradients of one network go in a render node's backward.
    Returns (grad_w, grad_b, accumulate, returned): the tensors K3's dW launch writes (or adds into, `accumulate`), and what the
    node hands autograd for the 24 parameter inputs.

    Parameters that live in an optimiser's arena (optim.FlatAdam): the launch writes the arena's gradient buffer and autograd
    gets fresh VIEWS of it, which AccumulateGrad adopts as `.grad` without a copy when the gradient is None (after the
    reference's zero_grad(), train.py:196-200).  When the slots already hold live data - `.grad` is a window on them (a caller
    accumulating over several backward passes), or another render node of THIS pass wrote them (two `graph.render` calls in one
    loss) - the launch adds on top and autograd is handed nothing (None = no contribution).
    Anything else (no arena, mixed states, a torch without graph-task ids): fresh buffers, autograd accumulates them."""
    params = list(net.weights) + list(net.biases)
    n_w = len(net.weights)
    hit = optim.arena_of(params)
    pid = _pass_id()
    if hit is not None and pid != -1:
        arena, idx = hit
        n = len(params)
        windows = [arena.grad_is_window(i) for i in idx]
        same_pass = arena.pass_ids.get(idx[0]) == pid
        views = [arena.grad_view(i) for i in idx]
        if same_pass or all(windows):
            arena.pass_ids[idx[0]] = pid
            return views[:n_w], views[n_w:], True, [None] * n
        if not any(windows):
            arena.pass_ids[idx[0]] = pid
            return views[:n_w], views[n_w:], False, views
    gw, gb = [torch.empty_like(w) for w in net.weights], [torch.empty_like(b) for b in net.biases]
    return gw, gb, False, gw + gb


class RenderRays(torch.autograd.Function):
    """Graph.render as ONE autograd node (model/nerf.py:236-343).

    inputs: poses [P,3,4], then the 24 coarse and 24 fine parameter tensors (weights then
    biases, reference state-dict order).  outputs: rgb_map, disp_map, acc_map, rgb0, disp0,
    acc0, sigma (sigma is not differentiable here - it is never used in a loss)."""

    @staticmethod
    def forward(ctx, poses, ray_idx, cam, ndc, n_samples, n_importance, draws, net_c, net_f, *params):
        poses_c = poses.detach().contiguous()
        ctx.set_materialize_grads(False)
        # ctx.needs_input_grad says which inputs REQUIRE grad, also under torch.no_grad() (where forward still runs with grad mode
        # off and no backward can follow): the caller's grad mode travels on the Draws object.  Without it render_video / render_*_test
        # (torch.no_grad) ran TRAINING launches - activations saved for nobody, 25 % slower, and not the BENERF_MLP_AUTO inference
        # launch with its exact-f32 fallback (round 6: found in the kernel trace of a frame)
        need_grad = any(ctx.needs_input_grad) and getattr(draws, "grad_enabled", True)
        if need_grad and K.is_split():
            # range guard of the autograd path: the previous backward posted the device's status words to pinned host memory;
            # gradients that left the f16 range (inf / NaN into torch.optim) are reported here, without a synchronisation
            K.range_guard(poses_c.device).poll()
        net_c.pack_if_stale()
        if net_f is not None:
            net_f.pack_if_stale()
        out, saved = _render_forward(cam, ndc, n_samples, n_importance, draws, poses_c, ray_idx, net_c, net_f, need_grad)
        ctx.cam, ctx.ndc, ctx.draws, ctx.net_c, ctx.net_f = cam, ndc, draws, net_c, net_f
        ctx.poses, ctx.ray_idx, ctx.saved_k = poses_c, ray_idx, saved if need_grad else None
        ctx.fine = n_importance > 0
        ctx.mark_non_differentiable(out["sigma"])
        if ctx.fine:
            return (out["rgb_map"], out["disp_map"], out["acc_map"], out["rgb0"], out["disp0"], out["acc0"], out["sigma"])
        return (out["rgb_map"], out["disp_map"], out["acc_map"], out["sigma"])

    @staticmethod
    def backward(ctx, *gs):
        def c(t):
            return None if t is None else t.contiguous()

        if ctx.fine:
            g = {"rgb_map": c(gs[0]), "disp_map": c(gs[1]), "acc_map": c(gs[2]), "rgb0": c(gs[3]), "disp0": c(gs[4]),
                 "acc0": c(gs[5])}
        else:
            g = {"rgb_map": c(gs[0]), "disp_map": c(gs[1]), "acc_map": c(gs[2])}
        net_c, net_f = ctx.net_c, ctx.net_f
        tc = _grad_targets(net_c)
        tf = _grad_targets(net_f) if net_f is not None else (None, None, False, [])
        d_poses = _render_backward(ctx.cam, ctx.ndc, ctx.draws, ctx.poses, ctx.ray_idx, net_c, net_f, ctx.saved_k, g,
                                   (tc[0], tc[1]), (tf[0], tf[1]), (tc[2], tf[2]))
        if K.is_split():
            K.range_guard(d_poses.device).post()
        ctx.saved_k = None
        return (d_poses, None, None, None, None, None, None, None, None) + tuple(tc[3]) + tuple(tf[3])


class RenderPair(torch.autograd.Function):
    """The two renders of one training iteration - event batch (cam_e, poses_e) and blur batch (cam_r, poses_r) of
    Graph.forward (model/nerf.py:160-234) - as ONE autograd node over ONE batched launch sequence: rays of both batches in one
    buffer, one coarse and one fine K3 launch, compositing per batch (separate, non-view outputs: the reference's loop may
    scale what it gets in place), one dX / dW launch per network in the backward.  Same kernels and order as TrainStep.

    inputs: poses_e [Pe,3,4], poses_r [Pr,3,4], then the 24 coarse and 24 fine parameter tensors.
    outputs: (rgb_map, disp_map, acc_map, rgb0, disp0, acc0, sigma) of the event batch, then the same seven of the blur batch
    (sigma is not differentiable here - it is never used in a loss); then, with pose_chunks, the per-pose row blocks of the four
    colour maps as outputs of their own (event rgb_map, event rgb0, blur rgb_map, blur rgb0: Pe + Pe + Pr + Pr views of the maps
    above).  The reference's loop takes exactly these blocks out of the maps - `ret_event["rgb_map"][:pixels_num]`, the
    num_interpolated_pose slices of the blur loop (train.py:166-175, 307-315) - and autograd answers every such slice of a
    differentiable tensor with three launches in the backward pass (zero-fill of a full-size gradient, copy of the slice, add
    into the running sum): ~126 launches per iteration, 1 ms of host time with the device idle.  model/nerf.py's PoseMajorRows
    hands out these outputs for those very slices instead: the block gradients arrive here directly and are joined by one
    concatenation per map."""

    @staticmethod
    def forward(ctx, poses_e, poses_r, idx_e, idx_r, cam_e, cam_r, ndc, n_samples, n_importance, draws, net_c, net_f, pose_chunks, *params):
        pe, pr = poses_e.detach().contiguous(), poses_r.detach().contiguous()
        dev = pe.device
        ctx.set_materialize_grads(False)       # outputs no loss uses (disp, acc, unused blocks) arrive as None, not as zero-filled tensors
        need_grad = any(ctx.needs_input_grad) and getattr(draws, "grad_enabled", True)      # the caller's grad mode: see RenderRays.forward
        if need_grad and K.is_split():
            K.range_guard(dev).poll()          # what the previous backward posted (see RenderRays.forward)
        if net_c._key() != net_c.version or net_f._key() != net_f.version:
            K.PackedMlp.pack_pair(net_c, net_f)        # one re-pack launch per optimiser step for both networks
        S, Ni = n_samples, n_importance
        Ne, Nr = pe.shape[0] * idx_e.shape[0], pr.shape[0] * idx_r.shape[0]
        N = Ne + Nr
        ro = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rd, vd = torch.empty_like(ro), torch.empty_like(ro)
        K.rays_fwd(pe, idx_e, cam_e.H, cam_e.W, cam_e.fx, cam_e.fy, cam_e.cx, cam_e.cy, ndc, out=(ro[:Ne], rd[:Ne], vd[:Ne]), remap=cam_e.remap)
        K.rays_fwd(pr, idx_r, cam_r.H, cam_r.W, cam_r.fx, cam_r.fy, cam_r.cx, cam_r.cy, ndc, out=(ro[Ne:], rd[Ne:], vd[Ne:]), remap=cam_r.remap)
        t_rand, seed, off = draws.jitter_args()
        z = K.stratified_z(N, S, dev, t_rand, seed, off, draws.near, draws.far)
        parts = ((0, Ne), (Ne, N))

        def composite(raw, zz, which, weights=None):
            """compositing per batch: separate outputs; the sampling weights of both land in one [N, S] buffer"""
            outs = []
            for k, (a, b) in enumerate(parts):
                nz = draws.noise_args(which, k)
                o_ = {"weights": weights[a:b]} if weights is not None else None
                want = ("rgb_map", "disp", "acc") + (("weights",) if weights is not None else ("sigma",))
                outs.append(K.composite_fwd(raw[a:b], zz[a:b], rd[a:b], None if nz[0] is None else nz[0][a:b], nz[1], nz[2], nz[3],
                                            want=want, out=o_))
            return outs

        raw0, acts0 = K.mlp_fwd(net_c, ro, rd, vd, z, need_grad)
        w0 = torch.empty((N, S), dtype=torch.float32, device=dev)
        c0 = composite(raw0, z, 0, w0)
        u, useed, uoff = draws.u_args()
        z_fine = K.sample_pdf_merge(z, w0, Ni, u, useed, uoff)
        raw1, acts1 = K.mlp_fwd(net_f, ro, rd, vd, z_fine, need_grad)
        c1 = composite(raw1, z_fine, 1)
        ctx.cfg = (cam_e, cam_r, ndc, draws, net_c, net_f, pe, pr, idx_e, idx_r, Ne, N)
        ctx.saved_k = (ro, rd, vd, z, z_fine, raw0, raw1, acts0, acts1) if need_grad else None
        outs = []
        for k in range(2):
            outs += [c1[k]["rgb_map"], c1[k]["disp"], c1[k]["acc"], c0[k]["rgb_map"], c0[k]["disp"], c0[k]["acc"], c1[k]["sigma"]]
        ctx.mark_non_differentiable(outs[6], outs[13])
        ctx.chunks = None
        if pose_chunks:
            Pe, Pr, Re, Rr = pe.shape[0], pr.shape[0], idx_e.shape[0], idx_r.shape[0]
            ctx.chunks = (Pe, Pr)
            for k, (P, R) in enumerate(((Pe, Re), (Pr, Rr))):
                for m in (outs[7 * k], outs[7 * k + 3]):            # rgb_map, rgb0 of batch k
                    outs += [m[j * R:(j + 1) * R] for j in range(P)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        cam_e, cam_r, ndc, draws, net_c, net_f, pe, pr, idx_e, idx_r, Ne, N = ctx.cfg
        ro, rd, vd, z, z_fine, raw0, raw1, acts0, acts1 = ctx.saved_k
        ctx.saved_k = None
        dev = ro.device
        S, Sf = z.shape[1], z_fine.shape[1]
        C1 = raw0.shape[-1]
        parts = [[0, Ne, list(gs[0:7])], [Ne, N, list(gs[7:14])]]
        if ctx.chunks is not None:
            # gradients of the per-pose row blocks: joined (one concatenation per map) and added to the whole map's gradient, if any
            o = 14
            for k, P in enumerate(ctx.chunks):
                rows = (parts[k][1] - parts[k][0]) // P
                for slot in (0, 3):
                    blk = gs[o:o + P]
                    o += P
                    if all(b is None for b in blk):
                        continue
                    ref = next(b for b in blk if b is not None)
                    if P > 2 and all(b is not None and b.data_ptr() == ref.data_ptr() and b.shape == ref.shape for b in blk):
                        # the blur average hands every pose block the SAME gradient tensor (a sum passes its gradient through): one
                        # tiled copy instead of a concatenation of P tensors (a third of its host time)
                        joined = ref.repeat(P, 1) if ref.dim() == 2 else ref.repeat(P)
                    else:
                        joined = torch.cat([b if b is not None else ref.new_zeros((rows,) + tuple(ref.shape[1:])) for b in blk])
                    parts[k][2][slot] = joined if parts[k][2][slot] is None else parts[k][2][slot] + joined

        def c(t):
            return None if t is None else t.contiguous()

        d_d = torch.empty_like(ro)
        amax = torch.zeros(2, dtype=torch.float32, device=dev)

        def composite_bwd(raw, zz, which, slot, first):
            """d_raw of both batches in one [N, S, C+1] buffer; the ||rays_d|| term starts (first) or extends d_d"""
            d_raw = torch.empty_like(raw)
            for k, (a, b, g) in enumerate(parts):
                nz = draws.noise_args(which, k)
                g_rgb, g_disp, g_acc = (g[0], g[1], g[2]) if slot == 1 else (g[3], g[4], g[5])
                if g_rgb is None:
                    g_rgb = torch.zeros((b - a, C1 - 1), dtype=torch.float32, device=dev)
                K.composite_bwd(raw[a:b], zz[a:b], rd[a:b], None if nz[0] is None else nz[0][a:b], nz[1], nz[2], nz[3], c(g_rgb), c(g_acc),
                                None, c(g_disp), d_rays_d=d_d[a:b], accumulate=not first, absmax_out=amax[slot:slot + 1], d_raw_out=d_raw[a:b])
            return d_raw

        # the device idles while autograd walks the caller's loss lines: the first large launch (the fine network's dX chain) goes out
        # with as little host work in front of it as possible, everything else is prepared while it runs
        d_raw1 = composite_bwd(raw1, z_fine, 1, 1, True)
        dx1 = K.mlp_bwd_dx(net_f, d_raw1.view(-1, C1), acts1, N, Sf, slot="_fine", d_raw_absmax=amax[1:2])
        d_raw0 = composite_bwd(raw0, z, 0, 0, False)
        d_o, d_v = torch.empty_like(ro), torch.empty_like(ro)
        grads = {}
        for net, d_raw, acts, zz, ns, slot, dx in ((net_f, d_raw1, acts1, z_fine, Sf, 1, dx1), (net_c, d_raw0, acts0, z, S, 0, None)):
            gw, gb, acc, ret = _grad_targets(net)
            if dx is None:
                dx = K.mlp_bwd_dx(net, d_raw.view(-1, C1), acts, N, ns, slot="_coarse", d_raw_absmax=amax[slot:slot + 1])
            d_pts, d_vp, dacts = dx
            K.mlp_bwd_dw(net, d_raw.view(-1, C1), acts, dacts, N, ns, gw, gb, acc)
            K.ray_grad_reduce(zz, d_pts, d_vp, d_o, d_d, d_v, 2 if slot == 1 else 1)   # d_d holds the compositing part; d_o, d_v start with the fine pass
            grads[slot] = ret
        dp_e = K.rays_bwd(pe, idx_e, cam_e.H, cam_e.W, cam_e.fx, cam_e.fy, cam_e.cx, cam_e.cy, ndc, d_o[:Ne], d_d[:Ne], d_v[:Ne], remap=cam_e.remap)
        dp_r = K.rays_bwd(pr, idx_r, cam_r.H, cam_r.W, cam_r.fx, cam_r.fy, cam_r.cx, cam_r.cy, ndc, d_o[Ne:], d_d[Ne:], d_v[Ne:], remap=cam_r.remap)
        if K.is_split():
            K.range_guard(dev).post()
        return (dp_e, dp_r) + (None,) * 11 + tuple(grads[0]) + tuple(grads[1])


# ------------------------------------------------------------------------------------------
# lean fixed-shape training step
# ------------------------------------------------------------------------------------------
class FlatNet:
    """One NeRF's 24 parameter tensors re-homed into flat fp32 buffers (param, grad, Adam m/v).
    The nn.Parameters keep their names/shapes (checkpoint compatible) but alias the flat
    storage, so the fused Adam (K8) and the gradient all-reduce see one contiguous range."""

    def __init__(self, weights, biases, channels, flat_param, flat_grad, offset):
        self.views_w, self.views_b, self.gviews_w, self.gviews_b = [], [], [], []
        o = offset
        for lst, vl, gl in ((weights, self.views_w, self.gviews_w), (biases, self.views_b, self.gviews_b)):
            for t in lst:
                n = t.numel()
                v = flat_param[o:o + n].view(t.shape)
                v.copy_(t.detach())
                if isinstance(t, torch.nn.Parameter) or t.requires_grad:
                    t.data = v
                vl.append(v)
                gl.append(flat_grad[o:o + n].view(t.shape))
                o += n
        self.end = o
        self.packed = K.PackedMlp(self.views_w, self.views_b, channels)


def nerf_param_lists(module):
    """(weights, biases) of a NeRF module in C-ABI order."""
    ws = [getattr_path(module, n).weight for n in K.LAYER_NAMES]
    bs = [getattr_path(module, n).bias for n in K.LAYER_NAMES]
    return ws, bs


def getattr_path(obj, dotted):
    for part in dotted.split("."):
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj


class TrainStep:
    """One full training iteration (train.py:153-394 semantics) as a fixed kernel sequence.

    Per rank it renders its shard of the event pixels (2 poses) and blur pixels (n poses) in one
    batched launch sequence, computes loss + gradients (K6), backpropagates through both MLPs,
    rays and the trajectory, all-reduces the flat gradient buffer (when world_size > 1) and
    applies Adam with the exponential LR schedule.

    Range guard (split arithmetic modes): a step whose activations / scaled gradients left the f16 range, or whose loss gradient
    is not finite, leaves parameters and Adam moments untouched on every rank and is counted on the device; Adam's bias
    correction counts APPLIED steps only (t = iteration - skipped steps, like torch's GradScaler), the LR schedule follows the
    iteration count like the reference's (train.py:355-394).
    """
    MAX_SKIPPED_IN_A_ROW = 3     # consecutive range-guard skips after which step() raises (checked without synchronising)
    GUARD_POST_EVERY = 8         # steps between two copies of the guard's counters to the host

    def __init__(self, graph, cfg, cam_rgb, cam_evt, device, world_size=1, rank=0, process_group=None, seed=0, event_bins=1,
                 uneven_shards=False):
        # Dense event bins (BASELINE.json configs[4]; SURVEY 8, note under the config table: an extension, the reference has one
        # bin per step): the event span [evt_ts2[0], evt_ts2[1]] is cut into `event_bins` contiguous equal bins, the event batch
        # is rendered ONCE at the event_bins + 1 bin boundaries (get_pose_evt(args, ts, seg_num=B + 1), model/optimize.py:58-82)
        # and every bin contributes the reference's event-loss term (train.py:204-292) on its pose pair and its own accumulated
        # polarity image; event_bins = 1 is the reference's step.
        # uneven_shards: a global batch the ranks cannot split evenly is rendered whole instead of being refused - the left-over
        # blur pixels go one each to the low ranks and the event pixels are dealt so that every rank renders the same number of
        # RAYS +- one pixel's worth (dist.balanced_shard_bounds); the loss means use the true global counts either way
        self.uneven_shards = bool(uneven_shards)
        # wait_events: when a list, every wait of the main / side stream for a gradient bucket is bracketed by HIP events
        # (bucket name, before, after) - bench.py reports how long a step actually stalls on each exchange
        self.wait_events = None
        self.event_bins = int(event_bins)
        if self.event_bins < 1:
            raise ValueError("TrainStep: event_bins must be >= 1")
        if self.event_bins > 1 and (getattr(cfg, "optimize_event_crf", False) or getattr(cfg, "optimize_rgb_crf", False)):
            raise NotImplementedError("TrainStep: event_bins > 1 with the CRF tone-mappers is not built (they are off in every shipped config)")
        # switches of train.py:180-352 this fused sequence does not implement are refused, not ignored
        self.use_rgb_crf = bool(getattr(cfg, "optimize_rgb_crf", False))
        self.use_evt_crf = bool(getattr(cfg, "optimize_event_crf", False))
        if (self.use_rgb_crf or self.use_evt_crf) and cfg.channels != 1:
            raise NotImplementedError("the reference's tone-mappers are built with input_type='Gray' (model/optimize.py:24-26): one "
                                      "channel; with channels = 3 its own Linear(1, width) cannot take the colours either")
        if cfg.N_importance <= 0 or not hasattr(graph, "nerf_fine"):
            raise NotImplementedError("TrainStep needs the fine network (N_importance > 0), as in every shipped config")
        if cfg.dataset == "TUM_VIE" and (cam_rgb.remap is None or cam_evt.remap is None):
            raise ValueError("TrainStep: TUM_VIE needs the undistortion tables (Camera(..., remap=...), model/nerf.py:247-250)")
        if not (getattr(cfg, "event_loss", True) or getattr(cfg, "rgb_loss", True)):
            raise ValueError("TrainStep: event_loss and rgb_loss are both off - nothing to optimise")
        self.g, self.cfg, self.cam_rgb, self.cam_evt = graph, cfg, cam_rgb, cam_evt
        self.dev, self.world, self.rank, self.pg = device, world_size, rank, process_group
        if process_group is not None:      # RCCL's streams join the step's: GPU_MAX_HW_QUEUES matters (benerf_amd.configure_runtime)
            import benerf_amd
            benerf_amd.warn_if_few_hw_queues()
        self.seed = seed
        self.dw_stream = torch.cuda.Stream(device=device)    # weight-gradient launches (step(): backward)
        # BENERF_DW_STREAM (measurement knob): "last" - only the LAST weight-gradient launch of the step (the coarse
        # network's) goes to the side stream; "both" - the fine network's too, beside the coarse dX chain; "main" - none; "auto"
        # (default) - "last" when the fine launch fills the device many times over, "both" for small per-rank batches (step())
        mode = os.environ.get("BENERF_DW_STREAM", "auto")
        if mode not in ("auto", "last", "both", "main"):
            raise ValueError("BENERF_DW_STREAM must be auto, last, both or main, not %r" % mode)
        self.dw_stream_mode = mode
        self.dw_on_side_stream = mode != "main"
        self.C = cfg.channels
        wc, bc = nerf_param_lists(graph.nerf)
        wf, bf = nerf_param_lists(graph.nerf_fine)
        n_net = sum(t.numel() for t in wc + bc)
        self.n_net = n_net
        # flat layout: [coarse net | fine net | knots 24 | transform 6 | range-guard verdict 1 | tone-mapper parameters]
        crf_mods = [(getattr(graph, "rgb_crf", None), getattr(cfg, "rgb_crf_lrate", 5e-4), getattr(cfg, "decay_rate_rgb_crf", 0.1), 3)
                    if self.use_rgb_crf else None,
                    (getattr(graph, "event_crf", None), getattr(cfg, "event_crf_lrate", 5e-4), getattr(cfg, "decay_rate_event_crf", 0.1), 4)
                    if self.use_evt_crf else None]
        crf_mods = [m for m in crf_mods if m is not None]
        n_crf = sum(p_.numel() for m in crf_mods for p_ in m[0].parameters())
        n_total = 2 * n_net + 24 + 6 + 1 + n_crf
        self.flat_p = torch.zeros(n_total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(n_total, dtype=torch.float32, device=device)
        self.flat_m = torch.zeros(n_total, dtype=torch.float32, device=device)
        self.flat_v = torch.zeros(n_total, dtype=torch.float32, device=device)
        self.net_c = FlatNet(wc, bc, self.C, self.flat_p, self.flat_g, 0)
        self.net_f = FlatNet(wf, bf, self.C, self.flat_p, self.flat_g, n_net)
        o = 2 * n_net
        kn = graph.evt_knot_pose_se3.params.weight
        tr = graph.transform.params.weight
        self.knots = self.flat_p[o:o + 24].view(4, 6)
        self.knots.copy_(kn.detach())
        kn.data = self.knots
        self.transform = self.flat_p[o + 24:o + 30].view(1, 6)
        self.transform.copy_(tr.detach())
        tr.data = self.transform
        # the nn.Parameters alias the flat buffer through `.data = view`, which does NOT share version counters: a torch-side
        # write through graph.evt_knot_pose_se3 / graph.transform bumps THEIR counters, not flat_p's (_param_versions)
        self._traj_params = (kn, tr)
        self.g_knots = self.flat_g[o:o + 24].view(4, 6)
        self.g_transform = self.flat_g[o + 24:o + 30].view(1, 6)
        self.off_pose = o
        # Range guard of the split-f16 mode (include/benerf_hip.h `status`): this step's own words.  flag: this rank's verdict
        # as a float behind the trajectory gradients - it rides their all-reduce, so every replica skips the same steps.
        self.guard = K.RangeGuard(device)
        self.flag = self.flat_g[o + 30:o + 31]
        # max |d_raw| of the two networks (compositing backward -> dX chain): the guard's two per-step scratch words, which the
        # step gate zeroes on the device at the end of every step - no memset launch of their own
        self.amax = self.guard.words[_lib.ST_STEP_SCRATCH:_lib.ST_STEP_SCRATCH + 2].view(torch.float32)
        # CRF tone-mappers (train.py:180-192; off in every shipped config): 385 parameters each, applied to the rendered colours
        # between compositing and the loss kernels as torch modules (per-ray work, thousands of rows); their parameters live in
        # the flat buffers like everything else: same fused Adam (same range gate), same all-reduce bucket
        self.crf_params, self.crf_groups = [], []       # groups: (offset, count, lr0, decay, index in setup_optimizer's tuple)
        oc = o + 31
        for mod, lr0, dr, slot in crf_mods:
            first = oc
            for p_ in mod.parameters():
                n_ = p_.numel()
                v = self.flat_p[oc:oc + n_].view(p_.shape)
                v.copy_(p_.detach())
                p_.data = v
                self.crf_params.append(p_)
                oc += n_
            self.crf_groups.append((first, oc - first, lr0, dr, slot))
        self.off_crf = o + 31
        self.global_step = 0
        # keep_maps: parity runs set it - the step then also composites acc / disp and leaves the per-ray outputs of its two
        # compositing passes in last_maps (rows: event batch pose-major, then blur batch pose-major)
        self.keep_maps = False
        self.last_maps = None
        self._prefetched = None     # next step's ray set-up, computed in this step's slack (step(overlap=...))
        # replicas start from rank 0's parameters (and its - zero - Adam state), whatever their local initialisation was
        for buf in (self.flat_p, self.flat_m, self.flat_v):
            dist.broadcast_(buf, self.world, 0, self.pg)
        self.net_c.packed.pack()
        self.net_f.packed.pack()
        self.last_losses = None

    # -- schedule (train.py:355-394) ------------------------------------------------------------
    def _lr(self, lr0, decay):
        # the reference recomputes lr AFTER optimizer.step() of iteration i from global_step = i
        # (train.py:355-394), so iteration k >= 1 runs with lr0 * decay^((k-1)/steps), iteration 0 with lr0
        k = max(self.global_step, 1) - 1
        return lr0 * (decay ** (k / (self.cfg.lrate_decay * 1000)))

    # -- optimiser state <-> torch.optim.Adam (checkpoint format of train.py:443-455) ------------------------------
    def _flat_slice(self, p):
        off = (p.data_ptr() - self.flat_p.data_ptr()) // 4
        assert 0 <= off and off + p.numel() <= self.flat_p.numel(), "parameter is not backed by this TrainStep's flat buffer"
        return off, p.numel()

    def _trained_optimizers(self, optimizers):
        cfg = self.cfg
        return ((optimizers[0], self._lr(cfg.lrate, cfg.decay_rate)), (optimizers[1], self._lr(cfg.pose_lrate, cfg.decay_rate_pose)),
                (optimizers[2], self._lr(cfg.transform_lrate, cfg.decay_rate_transform))) + \
            tuple((optimizers[slot], self._lr(lr0, dr)) for _, _, lr0, dr, slot in self.crf_groups)

    def export_optimizer_state(self, optimizers):
        """Fills the Adam objects of Model.setup_optimizer (nerf, pose, transform, ...) with this step's moments, step
        count and current learning rates, ready for checkpoint.save (the tone-mapper optimisers too, when they are trained)."""
        # Adam's step count = APPLIED steps (a step the range guard skipped did not touch the moments); synchronises
        applied = self.global_step - int(self.guard.words[_lib.ST_SKIPPED_TOTAL].item())
        for opt, lr in self._trained_optimizers(optimizers):
            for group in opt.param_groups:
                group["lr"] = lr
                for p in group["params"]:
                    off, n = self._flat_slice(p)
                    opt.state[p] = {"step": torch.tensor(float(applied), device="cpu"),
                                    "exp_avg": self.flat_m[off:off + n].view_as(p).clone(),
                                    "exp_avg_sq": self.flat_v[off:off + n].view_as(p).clone()}

    def import_optimizer_state(self, optimizers, global_step):
        """Inverse of export_optimizer_state after checkpoint.load: parameters were restored in place (they alias the
        flat buffer); moments and the step counter come from the Adam objects; weights are re-packed.  Adam's `step` in the
        checkpoint counts APPLIED steps (export_optimizer_state); the difference to the iteration count - steps the range guard
        skipped before the checkpoint - goes back into the device's [SKIPPED_TOTAL] word, so that the resumed run's bias
        correction continues exactly where the saved one stopped."""
        applied = None
        for opt, _ in self._trained_optimizers(optimizers):
            for group in opt.param_groups:
                for p in group["params"]:
                    st = opt.state.get(p)
                    if not st:
                        continue
                    off, n = self._flat_slice(p)
                    self.flat_m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    self.flat_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    if "step" in st:
                        a_ = int(float(st["step"]))
                        applied = a_ if applied is None else max(applied, a_)
        self.global_step = int(global_step)
        skipped = 0 if applied is None else max(self.global_step - applied, 0)
        self.guard.words[_lib.ST_SKIPPED_TOTAL:_lib.ST_SKIPPED_TOTAL + 1].fill_(skipped)
        self._prefetched = None
        self.net_c.packed.pack()
        self.net_f.packed.pack()

    def _param_versions(self):
        """Version counters that a torch-side write to the trajectory parameters bumps, whichever alias it goes through: the flat
        buffer's (step.knots / step.transform are true views of it) and the two nn.Parameters' own (their `.data` was re-pointed
        at those views; set_data does not share counters).  The fused Adam writes through raw pointers and bumps none of them."""
        return (self.flat_p._version, self._traj_params[0]._version, self._traj_params[1]._version)

    def invalidate_prefetch(self):
        """Drop the next step's prefetched poses / rays / depths.  Needed only after a write to the trajectory parameters that
        torch's version counters cannot see: through `param.data` (every access makes a fresh counter) or through a raw pointer."""
        self._prefetched = None

    def check_range(self, reset=True):
        """Synchronises.  Raises kernels.BenerfRangeError if a step since the last reset left the f16 range of the split mode
        (those steps were skipped on every rank: parameters and Adam moments untouched)."""
        self.guard.check(reset)

    def shard(self, idx_evt, idx_rgb):
        """This rank's contiguous slices of the global event / blur pixel-index vectors."""
        if not self.uneven_shards or self.world == 1:
            return dist.shard_indices(idx_evt, self.rank, self.world), dist.shard_indices(idx_rgb, self.rank, self.world)
        table = dist.balanced_shard_bounds(idx_evt.shape[0], idx_rgb.shape[0], self.event_bins + 1, self.cfg.num_interpolated_pose, self.world)
        # every rank computes the whole table and refuses the SAME batches: a rank with no event or no blur pixels would fail in its
        # own kernels (positive sizes required) and leave the others waiting in the all-reduce
        empty = [k for k, ((a0, a1), (b0, b1)) in enumerate(table) if a1 <= a0 or b1 <= b0]
        if empty:
            raise ValueError("TrainStep(uneven_shards): %d event / %d blur pixels over %d ranks leave rank(s) %s without event or blur "
                             "pixels - use a larger global batch or fewer ranks" % (idx_evt.shape[0], idx_rgb.shape[0], self.world, empty))
        (e0, e1), (r0, r1) = table[self.rank]
        return idx_evt[e0:e1].contiguous(), idx_rgb[r0:r1].contiguous()

    def _ray_setup(self, evt_ts2, rgb_ts2, idx_e, idx_r, d):
        """Poses of both trajectories (K1), rays of both batches (K2), stratified coarse depths: everything in front of the
        first MLP launch.  idx_*: this rank's shard; d: the step's Draws."""
        cfg, dev = self.cfg, self.dev
        P, S = cfg.num_interpolated_pose, cfg.N_samples
        Pe = self.event_bins + 1
        Ne, Nr = Pe * idx_e.shape[0], P * idx_r.shape[0]
        N = Ne + Nr
        traj = 1 if cfg.traj == "linear" else 0
        poses_e, poses_r = K.spline_poses_fwd_pair(self.knots, self.transform.view(6), evt_ts2, Pe, rgb_ts2, P, traj)
        ro = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rd = torch.empty_like(ro)
        vd = torch.empty_like(ro)
        ce, cr = self.cam_evt, self.cam_rgb
        K.rays_fwd(poses_e, idx_e, ce.H, ce.W, ce.fx, ce.fy, ce.cx, ce.cy, cfg.ndc, out=(ro[:Ne], rd[:Ne], vd[:Ne]), remap=ce.remap)
        K.rays_fwd(poses_r, idx_r, cr.H, cr.W, cr.fx, cr.fy, cr.cx, cr.cy, cfg.ndc, out=(ro[Ne:], rd[Ne:], vd[Ne:]), remap=cr.remap)
        t_rand, sd, off = d.jitter_args()
        z = K.stratified_z(N, S, dev, t_rand, sd, off)
        return poses_e, poses_r, ro, rd, vd, z

    def _timed_wait(self, name, handle):
        """handle.wait() on the current stream; bracketed by events when self.wait_events is a list (bench.py)."""
        if self.wait_events is None:
            handle.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        handle.wait()
        e1.record()
        self.wait_events.append((name, e0, e1))

    def step(self, *args, **kwargs):
        """One training iteration: see _step for the arguments.  If anything raises between the compositing backward and the
        step gate, the two per-step maxima of |d_raw| (atomic maxima in the guard's scratch words, zeroed by the gate) would
        stay behind and scale the next step's dX chain - they are cleared here, with the prefetched ray set-up."""
        try:
            return self._step(*args, **kwargs)
        except BaseException:
            self._prefetched = None
            try:
                self.amax.zero_()
            except Exception:      # noqa: BLE001 - the device may be the thing that failed; the original error matters
                pass
            raise

    def _step(self, evt_ts2, rgb_ts2, idx_evt_global, idx_rgb_global, events_accu, image, draws_evt=None,
              draws_rgb=None, z_fine_forced=None, overlap=None):
        """evt_ts2/rgb_ts2: [2] device floats; idx_*_global: int64 device pixel indices (global batch,
        identical on every rank); events_accu [H_e*W_e] float32 device ([event_bins, H_e*W_e] with dense event bins: row b = the
        polarity sum of bin b, K7 over [t_b, t_b+1]); image [H*W, C] float32 device.
        z_fine_forced [N, S+Ni]: parity runs may supply the merged fine depths instead of K5's (sample_pdf is
        ill-conditioned in the coarse weights: isolates everything behind it).
        overlap: optional callable, run on the main stream behind the step's last backward launch and before the step waits
        for its weight-gradient stream - where the main stream has ~0.6 ms of slack at C2.  For work that touches neither this
        step's parameters nor its gradients, e.g. preparing the NEXT batch's inputs (this step's reads of events_accu / image /
        the index vectors are already queued ahead of it on the same stream).  If it RETURNS the next step's
        (evt_ts2, rgb_ts2, idx_evt_global, idx_rgb_global), the step also sets up that step's poses, rays and coarse depths in
        the same slack (behind its own trajectory update); the next step() call uses them when it is handed those very
        tensor objects, and computes them itself otherwise."""
        cfg, C, dev = self.cfg, self.C, self.dev
        # range guard: a run whose steps keep being skipped on the device must not go on silently - looks at the last copy
        # of the counters that has landed in pinned host memory (no synchronisation)
        self.guard.poll(max_consecutive=self.MAX_SKIPPED_IN_A_ROW)
        st = self.guard.words
        P = cfg.num_interpolated_pose
        S, Ni = cfg.N_samples, cfg.N_importance
        idx_e, idx_r = self.shard(idx_evt_global, idx_rgb_global)
        Re, Rr = idx_e.shape[0], idx_r.shape[0]
        B = self.event_bins
        Pe = B + 1
        Ne, Nr = Pe * Re, P * Rr
        N = Ne + Nr
        traj = 1 if cfg.traj == "linear" else 0
        step_id = self.global_step
        if B > 1 and events_accu.numel() != B * ce_hw(self.cam_evt):
            raise ValueError("TrainStep: event_bins = %d needs events_accu of shape [%d, H_e * W_e]" % (B, B))

        # ---- forward ---------------------------------------------------------------------------
        ce, cr = self.cam_evt, self.cam_rgb
        if draws_evt is not None:   # parity mode: explicit draws for both renders, concatenated
            d = Draws(torch.cat([draws_evt.t_rand, draws_rgb.t_rand]), torch.cat([draws_evt.noise0, draws_rgb.noise0]),
                      torch.cat([draws_evt.u, draws_rgb.u]), torch.cat([draws_evt.noise1, draws_rgb.noise1]))
        else:
            d = Draws(seed=self.seed + self.rank * 7919, offset=step_id)
        pre, self._prefetched = self._prefetched, None
        if (pre is not None and draws_evt is None and pre["step_id"] == step_id and
                pre["versions"] == self._param_versions() + (evt_ts2._version, rgb_ts2._version, idx_evt_global._version, idx_rgb_global._version) and
                all(x is y for x, y in zip(pre["inputs"], (evt_ts2, rgb_ts2, idx_evt_global, idx_rgb_global)))):
            poses_e, poses_r, ro, rd, vd, z = pre["rays"]      # set up in the previous step's slack (see the end of this method)
        else:
            poses_e, poses_r, ro, rd, vd, z = self._ray_setup(evt_ts2, rgb_ts2, idx_e, idx_r, d)
        pw = None
        if getattr(cfg, "use_barf_c2f", False):     # iter_step of graph.forward(i, ...) = the iteration counter (train.py:160)
            pw = K.barf_pe_weights(step_id, cfg.max_iter, cfg.barf_c2f_start, cfg.barf_c2f_end, dev)
        self.net_c.packed.pe_weights = self.net_f.packed.pe_weights = pw
        raw0, acts0 = K.mlp_fwd(self.net_c.packed, ro, rd, vd, z, True, status=st)
        nz0 = d.noise_args(0)
        extra = ("disp", "acc") if self.keep_maps else ()
        c0 = K.composite_fwd(raw0, z, rd, nz0[0], nz0[1], nz0[2], nz0[3], want=("rgb_map", "weights") + extra)
        u, usd, uoff = d.u_args()
        z_fine = K.sample_pdf_merge(z, c0["weights"], Ni, u, usd, uoff) if z_fine_forced is None else z_fine_forced.contiguous()
        raw1, acts1 = K.mlp_fwd(self.net_f.packed, ro, rd, vd, z_fine, True, status=st)
        nz1 = d.noise_args(1)
        c1 = K.composite_fwd(raw1, z_fine, rd, nz1[0], nz1[1], nz1[2], nz1[3], want=("rgb_map",) + extra)
        rgb_map, rgb0 = c1["rgb_map"], c0["rgb_map"]
        if self.keep_maps:
            self.last_maps = {"rgb_map": rgb_map, "rgb0": rgb0, "acc_map": c1["acc"], "acc0": c0["acc"], "disp_map": c1["disp"],
                              "disp0": c0["disp"], "n_event_rays": Ne}

        # ---- loss + gradient w.r.t. rendered colours (K6) ------------------------------------------
        target_rgb = K.gather_rows(image, idx_r)
        syn = cfg.event_threshold > 0
        lcfg = K.make_loss_cfg(C, cfg.dataset.startswith("E2NeRF"), Re, Rr, P, cfg.event_threshold,
                               cfg.event_coeff_syn if syn else cfg.event_coeff_real, cfg.rgb_coeff,
                               idx_evt_global.shape[0], idx_rgb_global.shape[0])
        # args.event_loss / args.rgb_loss (train.py:201,299): a disabled term contributes neither loss nor gradient
        use_e, use_r = getattr(cfg, "event_loss", True), getattr(cfg, "rgb_loss", True)
        crf = self.use_rgb_crf or self.use_evt_crf
        if crf:   # train.py:180-192: the losses see the tone-mapped colours; the mappers are differentiated by torch autograd
            leaves = [rgb_map.requires_grad_(True), rgb0.requires_grad_(True)]
            mapped = []
            with torch.enable_grad():
                for t in leaves:
                    e_part = self.g.event_crf(t[:Ne]) if self.use_evt_crf else t[:Ne]
                    r_part = self.g.rgb_crf(t[Ne:]) if self.use_rgb_crf else t[Ne:]
                    mapped.append(torch.cat([e_part, r_part], 0))
            raw_maps = (rgb_map, rgb0)
            rgb_map, rgb0 = mapped[0].detach(), mapped[1].detach()
        # The L2-normalised event loss (train.py:238-292) needs the sums of squares - GLOBAL ones when data-parallel: a blocking
        # 16-double exchange - before its gradient.  The mean-squared losses do not: their gradients use the global COUNTS only,
        # so the gradient launch goes first and the sums + loss values follow behind the backward launches, off the critical
        # path (values are linear in the sums: every rank computes its part, the 8 numbers are summed asynchronously).
        stats_first = use_e and not syn
        g_rgb = torch.empty_like(rgb_map) if (use_e and use_r) else torch.zeros_like(rgb_map)
        g_rgb0 = torch.empty_like(rgb0) if (use_e and use_r) else torch.zeros_like(rgb0)
        loss_sum = None
        if B == 1:
            target_acc = K.gather_rows(events_accu.view(-1, 1), idx_e).view(-1)
            largs = ((rgb_map[:Ne], rgb0[:Ne], target_acc) if use_e else (None, None, None)) + \
                    ((rgb_map[Ne:], rgb0[Ne:], target_rgb) if use_r else (None, None, None))
            if stats_first:
                stats = K.loss_stats(lcfg, *largs)
                dist.allreduce_sum_(stats, self.world, self.pg)
            losses, _ = K.loss_grads(lcfg, stats if stats_first else None, *largs,
                                     out=((g_rgb[:Ne], g_rgb0[:Ne]) if use_e else (None, None)) + ((g_rgb[Ne:], g_rgb0[Ne:]) if use_r else (None, None)),
                                     want_losses=stats_first)

            def late_losses():      # loss values of the mean-squared losses: in the main stream's slack (below)
                st_ = K.loss_stats(lcfg, *largs)
                return K.loss_grads(lcfg, st_, *largs, want_grads=False)[0]
        else:
            # Dense event bins: bin b is the reference's event term (train.py:204-292) on the rendered colours of poses b (start)
            # and b + 1 (end) - rows [b Re, (b + 2) Re) of the pose-major event batch, a contiguous view - against bin b's
            # accumulated polarities; an interior pose collects the gradient of the two bins it bounds.  The blur term once.
            hw = ce_hw(ce)
            accu_b = events_accu.view(B, hw)
            tacc = [K.gather_rows(accu_b[b].reshape(-1, 1), idx_e).view(-1) for b in range(B)] if use_e else []
            pair = lambda t, b: t[b * Re:(b + 2) * Re]            # noqa: E731
            none3 = (None, None, None)
            rargs = (rgb_map[Ne:], rgb0[Ne:], target_rgb)
            if use_e:
                g_rgb[:Ne].zero_()
                g_rgb0[:Ne].zero_()
            stats_b = None
            if stats_first:      # ONE exchange for the sums of all bins and of the blur term
                stats_b = torch.stack([K.loss_stats(lcfg, pair(rgb_map, b), pair(rgb0, b), tacc[b], *none3) for b in range(B)] +
                                      ([K.loss_stats(lcfg, *none3, *rargs)] if use_r else []))
                dist.allreduce_sum_(stats_b, self.world, self.pg)
            tmp, tmp0 = torch.empty_like(rgb_map[:2 * Re]), torch.empty_like(rgb0[:2 * Re])
            ev_losses = []
            for b in range(B if use_e else 0):
                lb, _ = K.loss_grads(lcfg, stats_b[b] if stats_first else None, pair(rgb_map, b), pair(rgb0, b), tacc[b], *none3,
                                     out=(tmp, tmp0, None, None), want_losses=stats_first)
                pair(g_rgb, b).add_(tmp)
                pair(g_rgb0, b).add_(tmp0)
                ev_losses.append(lb)

            def sum_losses(ev, rg):
                """[total, event, event fine, event coarse, rgb, rgb fine, rgb coarse, 0] of the step from the per-bin vectors"""
                out_ = torch.zeros(8, dtype=torch.float32, device=dev)
                for lb in ev:
                    out_[1:4] += lb[1:4]
                if rg is not None:
                    out_[4:7] = rg[4:7]
                out_[0] = out_[1] + out_[4]
                return out_
            lr_ = None
            if use_r:
                lr_, _ = K.loss_grads(lcfg, stats_b[B] if stats_first else None, *none3, *rargs, out=(None, None, g_rgb[Ne:], g_rgb0[Ne:]),
                                      want_losses=stats_first)
            losses = sum_losses(ev_losses, lr_) if stats_first else None

            def late_losses():
                ev = [K.loss_grads(lcfg, K.loss_stats(lcfg, pair(rgb_map, b), pair(rgb0, b), tacc[b], *none3), pair(rgb_map, b),
                                   pair(rgb0, b), tacc[b], *none3, want_grads=False)[0] for b in range(B if use_e else 0)]
                rg = K.loss_grads(lcfg, K.loss_stats(lcfg, *none3, *rargs), *none3, *rargs, want_grads=False)[0] if use_r else None
                return sum_losses(ev, rg)

        if crf:   # gradients w.r.t. the tone-mapped colours -> the rendered colours and the tone-mapper parameters
            for p_ in self.crf_params:
                p_.grad = None
            torch.autograd.backward(mapped, [g_rgb, g_rgb0])
            g_rgb, g_rgb0 = leaves[0].grad.contiguous(), leaves[1].grad.contiguous()
            n_crf = sum(p_.numel() for p_ in self.crf_params)
            torch.cat([p_.grad.reshape(-1) for p_ in self.crf_params], out=self.flat_g[self.off_crf:self.off_crf + n_crf])
            rgb_map, rgb0 = raw_maps[0].detach(), raw_maps[1].detach()

        # ---- backward -------------------------------------------------------------------------------
        d_o = torch.empty_like(ro)
        d_d = torch.empty_like(ro)
        d_v = torch.empty_like(ro)
        # The weight-gradient launches (HBM-bound: they stream the saved activations and activation gradients once) go to
        # a second stream and run beside the other network's activation-gradient chain and the trajectory tail
        # (MFMA-bound, little HBM traffic); the two networks keep separate activation-gradient buffers for that.
        n = self.n_net
        main = torch.cuda.current_stream(dev)
        side = self.dw_stream if self.dw_on_side_stream else main
        # max |d_raw| of both networks comes out of the compositing backward (the split-f16 dX chain scales by it)
        amax = self.amax
        d_raw1, _ = K.composite_bwd(raw1, z_fine, rd, nz1[0], nz1[1], nz1[2], nz1[3], g_rgb, d_rays_d=d_d, absmax_out=amax[0:1])
        d_raw0, _ = K.composite_bwd(raw0, z, rd, nz0[0], nz0[1], nz0[2], nz0[3], g_rgb0, d_rays_d=d_d, accumulate=True,
                                    absmax_out=amax[1:2])
        # At full batch two K3 launches side by side are time slicing (0.91-1.00 of the sum of their times alone,
        # profiles/r04_overlap_probe.log: each fills the CUs' LDS or registers by itself), so the first network's weight gradients
        # stay on the main stream and only the LAST weight-gradient launch goes to the side stream, where it hides the trajectory
        # tail and the loss values.  With the first launch on the side stream too the C2 step measured 0.3 % shorter (9.07 vs 9.10 ms;
        # the same in the exact-f32 mode) at the price of HIP-event durations that count the co-running launch (dX read 2.40 instead
        # of 1.51 ms, the f32 dW 0.56 instead of 0.77 of its roof): not worth a roofline nobody can add up.  A small per-rank batch
        # is different - its launches are one or two waves of workgroups, the device is not full for long and the pair does
        # overlap: 1/8 of C2 1.45 vs 1.54 ms, C4 2.51 vs 2.70, C5 3.26 vs 3.38 - so below 2048 fine tiles (8 per CU) both go there.
        fine_on_side = self.dw_stream_mode == "both" or (self.dw_stream_mode == "auto" and N * (S + Ni) < 2048 * 128)
        # Which chain goes first is a knob, not a gain: with dX on the main stream and dW on the side stream the two chains are a
        # two-machine flow shop and Johnson's rule says "coarse first" for a small per-rank batch - measured at 1/8 of C2 / C4 / C5:
        # 1.688 / 2.698 / 3.400 ms against 1.671 / 2.689 / 3.377 ms fine-first (profiles/r05_one_eighth_batch_same_box.log: the
        # co-running launches slow each other by what the rule would save).  Default: fine first.
        coarse_first = os.environ.get("BENERF_BWD_ORDER", "fine_first") == "coarse_first"
        chains = [("fine_net", self.net_f, d_raw1, acts1, S + Ni, "_fine", amax[0:1], slice(n, 2 * n)),
                  ("coarse_net", self.net_c, d_raw0, acts0, S, "_coarse", amax[1:2], slice(0, n))]
        if coarse_first:
            chains.reverse()
        pending, dxs, last_bucket = {}, {}, None
        for k_, (name_, net_, d_raw_, acts_, ns_, slot_, amax_, sl_) in enumerate(chains):
            dxs[name_] = K.mlp_bwd_dx(net_.packed, d_raw_.view(-1, C + 1), acts_, N, ns_, slot=slot_, status=st, d_raw_absmax=amax_)
            stream_ = side if (k_ == 1 or fine_on_side) else main       # the LAST weight-gradient launch always goes to the side stream
            stream_.wait_stream(main)
            with torch.cuda.stream(stream_):
                K.mlp_bwd_dw(net_.packed, d_raw_.view(-1, C + 1), acts_, dxs[name_][2], N, ns_, net_.gviews_w, net_.gviews_b, False)
                # gradient exchange, buckets 1 and 3 of 3: this network's gradients are final - their all-reduce (RCCL over xGMI)
                # runs on the communicator's stream while the other network's backward computes.  A communicator runs its
                # collectives in ISSUE order, so the second network's bucket is issued BEHIND the trajectory bucket (below): the
                # trajectory gradients are final long before the last weight-gradient launch is, and round 6's loopback run showed the
                # main stream waiting 0.17-0.19 ms for "its" 31 floats - queued behind a bucket that waits for that launch
                if k_ == 0:
                    pending[name_] = dist.allreduce_sum_async_(self.flat_g[sl_], self.world, self.pg)
                else:
                    last_bucket = (name_, stream_, sl_)
        d_pts1, d_vp1, _ = dxs["fine_net"]
        d_pts0, d_vp0, _ = dxs["coarse_net"]
        K.ray_grad_reduce(z_fine, d_pts1, d_vp1, d_o, d_d, d_v, 2)       # d_d holds the compositing part; d_o, d_v start here
        K.ray_grad_reduce(z, d_pts0, d_vp0, d_o, d_d, d_v, 1)
        dp_e = K.rays_bwd(poses_e, idx_e, ce.H, ce.W, ce.fx, ce.fy, ce.cx, ce.cy, cfg.ndc, d_o[:Ne], d_d[:Ne], d_v[:Ne], remap=ce.remap)
        dp_r = K.rays_bwd(poses_r, idx_r, cr.H, cr.W, cr.fx, cr.fy, cr.cx, cr.cy, cfg.ndc, d_o[Ne:], d_d[Ne:], d_v[Ne:], remap=cr.remap)
        dk_e, dk_r, dt_r = K.spline_poses_bwd_pair(self.knots, self.transform.view(6), evt_ts2, Pe, rgb_ts2, P, traj, dp_e, dp_r)
        torch.add(dk_e, dk_r, out=self.g_knots)
        self.g_transform.copy_(dt_r)

        # ---- gradient exchange, bucket 3: the 30 trajectory gradients + this rank's range-guard verdict (+ the tone-mapper
        # gradients); then every bucket must have landed ---------------------------------------------------------------
        if self.world > 1:
            self.guard.gate(self.flag, phase=0)
        pending["trajectory"] = dist.allreduce_sum_async_(self.flat_g[2 * n:], self.world, self.pg)
        if not stats_first:      # loss values of the mean-squared losses: in the main stream's slack, like the caller's overlap work
            losses = late_losses()
            loss_sum = dist.allreduce_sum_async_(losses, self.world, self.pg)
        with torch.cuda.stream(last_bucket[1]):      # bucket 3: the network whose weight gradients finish last
            pending[last_bucket[0]] = dist.allreduce_sum_async_(self.flat_g[last_bucket[2]], self.world, self.pg)
        nxt = overlap() if overlap is not None else None

        # ---- Adam (K8) with the reference's per-group switches and LR schedule ---------------------------------
        # The trajectory's part comes first, still in the main stream's slack: its gradients (bucket 3) and the range guard's
        # verdict for this step (summed over the ranks) are final long before the weight-gradient stream is.  [SKIP] makes
        # every Adam launch of the step a no-op.
        self._timed_wait("trajectory", pending["trajectory"])
        self.guard.gate(self.flag if self.world > 1 else None, phase=1)
        t = self.global_step + 1

        def adam(lo, cnt, lr):
            K.adam_step(self.flat_p[lo:lo + cnt], self.flat_g[lo:lo + cnt], self.flat_m[lo:lo + cnt], self.flat_v[lo:lo + cnt], lr, t,
                        status=st)
        o = self.off_pose
        if cfg.optimize_pose:
            adam(o, 24, self._lr(cfg.pose_lrate, cfg.decay_rate_pose))
        if cfg.optimize_trans:
            adam(o + 24, 6, self._lr(cfg.transform_lrate, cfg.decay_rate_transform))
        if nxt is not None:
            # overlap() returned the NEXT step's (evt_ts2, rgb_ts2, idx_evt_global, idx_rgb_global): its poses (on the trajectory
            # just updated), rays and coarse depths need nothing else - set up here, behind the running weight-gradient
            # launches, instead of in front of the next forward launch (~45 us of dependent small kernels)
            nxt = tuple(nxt)
            dn = Draws(seed=self.seed + self.rank * 7919, offset=step_id + 1)
            # torch-side writes to the parameters (a checkpoint load, a test poking a weight) or to the inputs before the next
            # call bump these version counters - the set-up is then recomputed (the fused Adam writes through raw pointers)
            self._prefetched = {"inputs": nxt, "step_id": step_id + 1,
                                "versions": self._param_versions() + tuple(t_._version for t_ in nxt),
                                "rays": self._ray_setup(nxt[0], nxt[1], *self.shard(nxt[2], nxt[3]), dn)}
        with torch.cuda.stream(side):
            for nm in ("fine_net", "coarse_net"):
                self._timed_wait(nm, pending[nm])
        main.wait_stream(side)
        if loss_sum is not None:
            loss_sum.wait()
        if cfg.optimize_nerf:
            adam(0, 2 * self.n_net, self._lr(cfg.lrate, cfg.decay_rate))
        for lo, cnt, lr0, dr, _ in self.crf_groups:
            adam(lo, cnt, self._lr(lr0, dr))
        if self.global_step % self.GUARD_POST_EVERY == 0:
            self.guard.post()
        K.PackedMlp.pack_pair(self.net_c.packed, self.net_f.packed)
        self.global_step += 1
        self.last_losses = losses
        return losses


def ce_hw(cam):
    return cam.H * cam.W


def psnr(img, gt):
    """-10 log10(MSE) on [0,1] images (metrics.py:34,51-52,79-81)."""
    return float(-10.0 * math.log10(float(torch.mean((img - gt) ** 2))))

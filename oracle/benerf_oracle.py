"""CPU oracle for the BeNeRF training/rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The shipped path (benerf_amd/) never routes through it and fails loudly when
the HIP library is missing.

What it is: a plain PyTorch-fp32 (CPU) + numpy restatement, written for this
repository, of the algorithms on the hot path of WU-CVGL/BeNeRF.  Every function
cites the reference file:line it follows.  Random draws are explicit arguments
(the reference pulls them from the global torch generator in the order listed in
SURVEY.md section 3.3) so that the HIP kernels, the oracle and the reference can
be fed identical numbers.

Parity status: PINNED.  oracle/gen_golden.py imports the unmodified reference in
the build container, runs both on identical seeded inputs, asserts agreement and
writes the golden vectors under tests/golden/ (G1..G10 of SURVEY.md section 8c).
tests/test_oracle_golden.py re-checks the oracle against those files on every
run.  The reference ships no tests of its own for this path.

Parameter containers are plain dicts keyed with the reference's state-dict names
(`pts_linears.0.weight`, ... - model/nerf.py:53-64).
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# SE(3) / quaternion helpers                                    (reference: spline.py)
# --------------------------------------------------------------------------------------

TAYLOR_TERMS = 11  # nth=10 -> i = 0..10                     (spline.py:46,55)


def _taylor(theta, first_pair):
    """sum_i (-1)^i theta^(2i) / d_i with d_i = prod_{j<=i} (2j+a)(2j+a+1).

    first_pair a=1 gives (1-cos x)/x^2 (spline.py:46-53), a=2 gives (x-sin x)/x^3
    (spline.py:55-62).  Accumulation order and the pow/div sequence follow the
    reference loop so the float32 result is identical.
    """
    acc = torch.zeros_like(theta)
    denom = 1.0
    for i in range(TAYLOR_TERMS):
        denom *= (2 * i + first_pair) * (2 * i + first_pair + 1)
        acc = acc + (-1) ** i * theta ** (2 * i) / denom
    return acc


def hat(w):
    """so(3) vector -> 3x3 skew matrix (spline.py:28-34)."""
    w0, w1, w2 = w.unbind(-1)
    z = torch.zeros_like(w0)
    return torch.stack([torch.stack([z, -w2, w1], -1),
                        torch.stack([w2, z, -w0], -1),
                        torch.stack([-w1, w0, z], -1)], -2)


def rotvec_to_quat(r, eps=1e-9):
    """Rotation vector -> quaternion xyzw (spline.py:79-100).

    theta is HALF the rotation angle; the series branch is selected below eps with
    torch.where (both branches are evaluated, as in the reference).
    """
    x, y, z = r[..., 0], r[..., 1], r[..., 2]
    th = 0.5 * torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    lam = torch.sin(th) / (2.0 * th)
    big = torch.stack([lam * x, lam * y, lam * z, torch.cos(th)], -1)
    s = 1.0 / 2.0 - 1.0 / 12.0 * th ** 2 - 1.0 / 240.0 * th ** 4
    small = torch.stack([s * x, s * y, s * z, 1.0 - 1.0 / 2.0 * th ** 2 + 1.0 / 24.0 * th ** 4], -1)
    pick = (th < eps).unsqueeze(-1).expand(big.shape)
    return torch.where(pick, small, big)


def quat_to_rotvec(q, eps_theta=1e-20, eps_w=1e-10):
    """Quaternion xyzw -> rotation vector; plain arctan, not atan2 (spline.py:167-192)."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    th = torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    w_small = torch.abs(w) < eps_w
    lam = torch.where(
        w_small,
        torch.where(torch.logical_and(w_small, w < 0), -torch.pi / th, torch.pi / th),
        torch.where(th < eps_theta,
                    2.0 / w - 2.0 / 3.0 * (th ** 2) / (w * w * w),
                    2.0 * torch.arctan(th / w) / th))
    return torch.stack([lam * x, lam * y, lam * z], -1)


def quat_conj(q):
    """(spline.py:145-148)"""
    return torch.stack([-q[..., 0], -q[..., 1], -q[..., 2], q[..., 3]], -1)


def quat_left_matrix(q):
    """4x4 matrix Q(q) with Q(q) p = q (x) p, xyzw layout (spline.py:130-138)."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return torch.stack([torch.stack([w, -z, y, x], -1),
                        torch.stack([z, w, -x, y], -1),
                        torch.stack([-y, x, w, z], -1),
                        torch.stack([-x, -y, -z, w], -1)], -2)


def quat_mul(a, b):
    return (quat_left_matrix(a) @ b[..., None]).squeeze(-1)


def quat_to_rot(q):
    """(spline.py:111-118)"""
    b, c, d, a = q.unbind(-1)
    return torch.stack([
        torch.stack([1 - 2 * (c ** 2 + d ** 2), 2 * (b * c - a * d), 2 * (a * c + b * d)], -1),
        torch.stack([2 * (b * c + a * d), 1 - 2 * (b ** 2 + d ** 2), 2 * (c * d - a * b)], -1),
        torch.stack([2 * (b * d - a * c), 2 * (a * b + c * d), 1 - 2 * (b ** 2 + c ** 2)], -1)], -2)


def se3_to_quat_trans(wu):
    """se(3) [w,u] -> (q xyzw, t = V(w) u), V = I + B wx + C wx^2 (spline.py:16-26)."""
    w, u = wu[..., :3], wu[..., 3:]
    wx = hat(w)
    theta = w.norm(dim=-1)[..., None, None]
    eye = torch.eye(3, dtype=wu.dtype)
    V = eye + _taylor(theta, 1) * wx + _taylor(theta, 2) * wx @ wx
    t = (V @ u[..., None]).squeeze(-1)
    return rotvec_to_quat(w), t


def nudge_unit_times(ts):
    """t==0 -> +1e-6, t==1 -> -1e-6 (spline.py:249-252; done in place there)."""
    ts = ts.clone()
    ts = torch.where(ts == 0, ts + 0.000001, ts)
    ts = torch.where(ts == 1, ts - 0.000001, ts)
    return ts


def bezier_poses(knots, ts):
    """Cubic Bezier in SE(3): Bernstein translation (bezier.py:7-20), cumulative-Bernstein rotation - the evident intent
    of the reference's non-executable bezier.py:22-74 (it raises IndexError on every call; no reference output exists)."""
    return cubic_spline_poses(knots, ts, bezier=True)


def cubic_spline_poses(knots, ts, bezier=False):
    """Uniform cubic B-spline in SE(3), cumulative form for rotation.

    knots [4,6] se(3), ts [P] in [0,1] -> poses [P,3,4]      (spline.py:247-303)
    """
    ts = nudge_unit_times(ts)
    u = ts.reshape(1, -1, 1)
    q = []
    t = []
    for k in range(4):
        qk, tk = se3_to_quat_trans(knots[k].reshape(1, 1, 6))
        q.append(qk)
        t.append(tk)
    uu = u ** 2
    uuu = u ** 3
    sixth = 1.0 / 6.0
    half = 0.5
    c0 = sixth - half * u + half * uu - sixth * uuu
    c1 = 4 * sixth - uu + half * uuu
    c2 = sixth + half * u + half * uu - half * uuu
    c3 = sixth * uuu
    r1 = 5 * sixth + half * u - half * uu + sixth * uuu
    r2 = sixth + half * u + half * uu - 2 * sixth * uuu
    r3 = sixth * uuu
    if bezier:
        v = 1.0 - u
        c0, c1, c2, c3 = v * v * v, 3.0 * v * v * u, 3.0 * v * uu, uuu
        r1, r2, r3 = c1 + c2 + c3, c2 + c3, c3
    trans = c0 * t[0] + c1 * t[1] + c2 * t[2] + c3 * t[3]
    d01 = quat_mul(quat_conj(q[0]), q[1])
    d12 = quat_mul(quat_conj(q[1]), q[2])
    d23 = quat_mul(quat_conj(q[2]), q[3])
    e0 = rotvec_to_quat(quat_to_rotvec(d01) * r1)
    e1 = rotvec_to_quat(quat_to_rotvec(d12) * r2)
    e2 = rotvec_to_quat(quat_to_rotvec(d23) * r3)
    p1 = quat_left_matrix(e1) @ e2[..., None]
    p2 = quat_left_matrix(e0) @ p1
    qt = (quat_left_matrix(q[0]) @ p2).squeeze(-1)
    R = quat_to_rot(qt)
    return torch.cat([R, trans.unsqueeze(-1)], -1).reshape(-1, 3, 4)


def linear_poses(knot_start, knot_end, ts):
    """traj=linear alternative: lerp t, geodesic rotation (spline.py:305-331)."""
    ts = nudge_unit_times(ts)
    qs, t_s = se3_to_quat_trans(knot_start.reshape(1, 1, 6))
    qe, t_e = se3_to_quat_trans(knot_end.reshape(1, 1, 6))
    s = ts.reshape(1, -1, 1)
    trans = (1 - s) * t_s + s * t_e
    rel = quat_mul(quat_conj(qs), qe)
    step = rotvec_to_quat(s * quat_to_rotvec(rel))
    qt = (quat_left_matrix(qs) @ step[..., None]).squeeze(-1)
    R = quat_to_rot(qt)
    return torch.cat([R, trans.unsqueeze(-1)], -1).reshape(-1, 3, 4)


def trajectory_poses(knots, transform, ts2, n_poses, traj="spline"):
    """get_pose_evt / get_pose_rgb (model/optimize.py:58-111).

    knots [4,6]; transform [1,6] or None (event camera: None; rgb camera: knots +
    transform in se(3), model/optimize.py:86-89); ts2 = (t_start, t_end).
    """
    k = knots if transform is None else knots + transform.reshape(1, 6)
    # sample times are float32 data in every evaluation precision (the float64 runs of oracle/f64_truth.py differentiate the
    # same function of the same inputs)
    ts = torch.linspace(float(ts2[0]), float(ts2[1]), n_poses, dtype=torch.float32).to(knots.dtype)
    if traj == "linear":
        return linear_poses(k[0], k[3], ts)
    return cubic_spline_poses(k, ts)


# --------------------------------------------------------------------------------------
# Rays                                                    (reference: run_nerf_helpers.py)
# --------------------------------------------------------------------------------------

def pixel_rays(ray_idx, W, K, poses):
    """Pose-major pinhole rays for pixel indices (model/nerf.py:241-254 +
    run_nerf_helpers.py:35-44).  ray_idx [R] int64, poses [P,3,4], K 3x3 float32.
    Returns rays_o, rays_d [P*R,3]."""
    P = poses.shape[0]
    R = ray_idx.shape[0]
    K = K.to(poses.dtype)      # float32 intrinsics (a no-op in the float32 evaluation)
    idx = ray_idx.repeat(P)
    c2w = poses.unsqueeze(1).repeat(1, R, 1, 1).reshape(-1, 3, 4)
    j = idx // W
    i = idx % W
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[..., :3, :3], -1)
    rays_o = c2w[..., :3, -1]
    return rays_o, rays_d


def ndc_project(H, W, focal, near, rays_o, rays_d):
    """LLFF NDC (run_nerf_helpers.py:46-71)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1.0 / (W / (2.0 * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1.0 / (H / (2.0 * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = -1.0 / (W / (2.0 * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1.0 / (H / (2.0 * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def make_rays(poses, ray_idx, H, W, K, ndc=True):
    """rays_o, rays_d (NDC'd when ndc), viewdirs (pre-NDC unit dirs)  (model/nerf.py:241-286)."""
    rays_o, rays_d = pixel_rays(ray_idx, W, K, poses)
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    if ndc:
        rays_o, rays_d = ndc_project(H, W, K[0][0], 1.0, rays_o, rays_d)
    return rays_o.to(poses.dtype), rays_d.to(poses.dtype), viewdirs.to(poses.dtype)


def stratified_z(n_rays, n_samples, t_rand, near=0.0, far=1.0):
    """Coarse depths with per-bin jitter t_rand [N,S] (model/nerf.py:297-307)."""
    t_vals = torch.linspace(0.0, 1.0, steps=n_samples)
    z = near * (1.0 - t_vals) + far * t_vals
    z = z.expand([n_rays, n_samples])
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


# --------------------------------------------------------------------------------------
# Positional encoding + MLP                 (reference: model/embedder.py, model/nerf.py)
# --------------------------------------------------------------------------------------

def posenc(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (model/embedder.py:9-34)."""
    out = [x]
    bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    for f in bands:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def xavier_params(rng, channels):
    """Xavier-uniform weights / zero biases for one NeRF (run_nerf_helpers.py:194-208,
    shapes model/nerf.py:53-64), drawn from a numpy Generator so the build can
    regenerate them without torch's RNG."""
    shapes = {"pts_linears.0": (256, 63)}
    for i in range(1, 8):
        shapes["pts_linears.%d" % i] = (256, 319 if i == 5 else 256)
    shapes["views_linears.0"] = (128, 283)
    shapes["feature_linear"] = (256, 256)
    shapes["alpha_linear"] = (1, 256)
    shapes["rgb_linear"] = (channels, 128)
    p = {}
    for name, (fo, fi) in shapes.items():
        a = math.sqrt(6.0 / (fi + fo))
        p[name + ".weight"] = torch.from_numpy(rng.uniform(-a, a, size=(fo, fi)).astype(np.float32))
        p[name + ".bias"] = torch.zeros(fo, dtype=torch.float32)
    return p


def barf_c2f(embedded, n_freqs, progress, start, end):
    """barf_c2f_weight (model/nerf.py:16-26) on an encoding WITHOUT the raw input ([.., 6L]): the reference's
    `embedded.view(-1, L) * weight` scales element e by weight[e % L]."""
    L = n_freqs
    alpha = (progress - start) / (end - start) * L
    k = torch.arange(L)
    weight = (1 - (alpha - k).clamp(min=0, max=1).mul(math.pi).cos()) / 2
    return (embedded.reshape(-1, L) * weight).reshape(embedded.shape)


def mlp_forward(p, pts, viewdirs, multires=10, multires_views=4, want_acts=False, barf=None, relu_masks=None):
    """NeRF.forward (model/nerf.py:67-116).  pts [N,S,3], viewdirs [N,3] -> raw [N,S,C+1].

    Layer 5 consumes cat[input_pts, h] (input first).  alpha_linear has no
    activation; rgb and alpha are concatenated as [rgb..., sigma].
    barf = (iter_step, max_iter, c2f_start, c2f_end): use_barf_c2f (model/nerf.py:78-89) - the encodings are built
    without the raw input, coarse-to-fine weighted, and the raw input is concatenated in front.
    relu_masks = {"h0".."h7", "hv": 0/1 tensors [N*S, width]}: the ReLUs are replaced by multiplication with these fixed
    masks (tests/test_f64_truth_gpu.py: a float64 evaluation of exactly the piecewise-linear branch another evaluation took -
    a pre-activation within round-off of zero otherwise flips between any two evaluations and moves gradients by far more
    than their arithmetic differs)."""
    def act(name, x):
        return torch.relu(x) if relu_masks is None else x * relu_masks[name].to(x.dtype)

    N, S = pts.shape[0], pts.shape[1]
    x = posenc(pts.reshape(-1, 3), multires)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    xd = posenc(dirs, multires_views)
    if barf is not None:
        it, max_iter, c0, c1 = barf
        x = torch.cat([x[:, :3], barf_c2f(x[:, 3:], multires, it / max_iter, c0, c1)], -1)
        xd = torch.cat([xd[:, :3], barf_c2f(xd[:, 3:], multires_views, it / max_iter, c0, c1)], -1)
    acts = {"pe": x, "ped": xd}
    h = x
    for i in range(8):
        h = act("h%d" % i, torch.nn.functional.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i]))
        acts["h%d" % i] = h
        if i == 4:
            h = torch.cat([x, h], -1)
    alpha = torch.nn.functional.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
    feat = torch.nn.functional.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
    acts["feat"] = feat
    hv = act("hv", torch.nn.functional.linear(torch.cat([feat, xd], -1), p["views_linears.0.weight"], p["views_linears.0.bias"]))
    acts["hv"] = hv
    rgb = torch.nn.functional.linear(hv, p["rgb_linear.weight"], p["rgb_linear.bias"])
    raw = torch.cat([rgb, alpha], -1).reshape(N, S, -1)
    if want_acts:
        return raw, acts
    return raw


# --------------------------------------------------------------------------------------
# Compositing                                         (reference: model/nerf.py:118-148)
# --------------------------------------------------------------------------------------

def composite(raw, z, rays_d, noise, channels):
    """raw2output.  noise [N,S] is the N(0,1)*raw_noise_std draw (std 1.0 always on in
    the reference, model/nerf.py:118,133-135) or None for no noise.
    Returns rgb_map, disp_map, acc_map, weights, depth_map, sigma."""
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10]).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :channels])
    pre = raw[..., channels] if noise is None else raw[..., channels] + noise
    sigma = torch.relu(pre)
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth = torch.sum(weights * z, -1)
    acc = torch.sum(weights, -1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    return rgb_map, disp, acc, weights, depth, sigma


# --------------------------------------------------------------------------------------
# Hierarchical sampling                          (reference: run_nerf_helpers.py:74-115)
# --------------------------------------------------------------------------------------

def sample_pdf_torch(bins, weights, u):
    """Op-for-op torch restatement (float `sum`, torch cumsum).  Returns samples, inds."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0 = torch.gather(cdf, 1, below)
    c1 = torch.gather(cdf, 1, above)
    b0 = torch.gather(bins, 1, below)
    b1 = torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0), inds


def sample_pdf_exact(bins, weights, u):
    """Fully specified numpy restatement - the arithmetic the HIP kernel K5 implements
    instruction for instruction, so kernel-vs-oracle is BIT-EXACT (indices and values).

    Reduction order is fixed where torch leaves it unspecified (SURVEY.md hard part 4):
      total = sequential float64 sum of float32 (w+1e-5), rounded to float32
      pdf   = float32 division
      cdf_k = float32( float64 running sum of pdf )   (== torch CPU cumsum, verified
              bit-identical in gen_golden.py)
      inds  = first k with cdf[k] > u (searchsorted right=True)
    bins [N,B], weights [N,B-1], u [N,Ni]  (float32) -> samples float32, inds int64,
    cdf float32 [N,B]."""
    bins = np.asarray(bins, np.float32)
    w = (np.asarray(weights, np.float32) + np.float32(1e-5)).astype(np.float32)
    u = np.asarray(u, np.float32)
    total = np.cumsum(w.astype(np.float64), axis=-1)[:, -1].astype(np.float32)
    pdf = (w / total[:, None]).astype(np.float32)
    cdf = np.concatenate([np.zeros((w.shape[0], 1), np.float32),
                          np.cumsum(pdf.astype(np.float64), axis=-1).astype(np.float32)], -1)
    nb = cdf.shape[-1]
    inds = np.empty(u.shape, np.int64)
    for r in range(u.shape[0]):
        inds[r] = np.searchsorted(cdf[r], u[r], side="right")
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, nb - 1)
    c0 = np.take_along_axis(cdf, below, 1)
    c1 = np.take_along_axis(cdf, above, 1)
    b0 = np.take_along_axis(bins, below, 1)
    b1 = np.take_along_axis(bins, above, 1)
    denom = (c1 - c0).astype(np.float32)
    denom = np.where(denom < np.float32(1e-5), np.float32(1.0), denom).astype(np.float32)
    t = ((u - c0).astype(np.float32) / denom).astype(np.float32)
    samples = (b0 + (t * (b1 - b0).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return samples, inds, cdf


def fine_depths(z, weights, u, exact=False):
    """z_mid, sample_pdf on weights[1:-1], detach, sort-merge (model/nerf.py:322-326)."""
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    if exact:     # numpy on the host whatever device the tensors live on (tests may evaluate the torch parts of the oracle on a GPU)
        s, _, _ = sample_pdf_exact(z_mid.detach().cpu().numpy(), weights[..., 1:-1].detach().cpu().numpy(), u.cpu().numpy())
        z_samples = torch.from_numpy(s).to(z.device)
    else:
        z_samples, _ = sample_pdf_torch(z_mid, weights[..., 1:-1], u)
    z_samples = z_samples.detach()
    z_all, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
    return z_all, z_samples


# --------------------------------------------------------------------------------------
# render()                                            (reference: model/nerf.py:236-343)
# --------------------------------------------------------------------------------------

def render(p_coarse, p_fine, poses, ray_idx, H, W, K, channels, n_samples, n_importance,
           draws, ndc=True, exact_pdf=False, want_extras=False, barf=None, z_forced=None, mlp_inputs_forced=None,
           near=0.0, far=1.0):
    """Graph.render.  draws = dict(t_rand [N,S], noise0 [N,S] | None, u [N,Ni],
    noise1 [N,S+Ni] | None) - the four RNG draws in reference order.
    z_forced = (z_coarse [N,S], z_fine [N,S+Ni]): depths of another evaluation of the same render, used INSTEAD of the
    stratified / importance samples (sample_pdf is ill-conditioned in the coarse weights; oracle/f64_truth.py hands the
    float32 evaluation's depths to the float64 one so that both differentiate the same function).
    mlp_inputs_forced = (pts_coarse [N,S,3], pts_fine [N,S+Ni,3], viewdirs [N,3]): VALUES of the network inputs of another
    evaluation, substituted with a straight-through gradient (x + (forced - x).detach()).  The float32 rounding of
    pts = o + d z is common to every float32 implementation and is amplified 2^9 times by the positional encoding; with the
    float32 values forced, a float64 evaluation measures the arithmetic of everything behind the inputs.
    Returns the reference's dict (rgb_map, disp_map, acc_map, rgb0, disp0, acc0, sigma)."""
    rays_o, rays_d, viewdirs = make_rays(poses, ray_idx, H, W, K, ndc)
    N = rays_o.shape[0]
    z = stratified_z(N, n_samples, draws["t_rand"], near, far) if z_forced is None else z_forced[0]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    pts_coarse = pts
    if mlp_inputs_forced is not None:
        pts = pts + (mlp_inputs_forced[0] - pts).detach()
        viewdirs = viewdirs + (mlp_inputs_forced[2] - viewdirs).detach()
    raw0 = mlp_forward(p_coarse, pts, viewdirs, barf=barf)
    rgb0, disp0, acc0, w0, depth0, sigma0 = composite(raw0, z, rays_d, draws.get("noise0"), channels)
    ret = {"rgb_map": rgb0, "disp_map": disp0, "acc_map": acc0}
    extras = {"rays_o": rays_o, "rays_d": rays_d, "viewdirs": viewdirs, "z_coarse": z,
              "raw0": raw0, "weights0": w0, "pts_coarse": pts_coarse}
    if n_importance > 0:
        if z_forced is None:
            z_all, z_samples = fine_depths(z, w0, draws["u"], exact=exact_pdf)
        else:
            z_all, z_samples = z_forced[1], None
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
        extras["pts_fine"] = pts
        if mlp_inputs_forced is not None:
            pts = pts + (mlp_inputs_forced[1] - pts).detach()
        raw1 = mlp_forward(p_fine, pts, viewdirs, barf=barf)
        rgb1, disp1, acc1, w1, depth1, sigma1 = composite(raw1, z_all, rays_d, draws.get("noise1"), channels)
        ret = {"rgb_map": rgb1, "disp_map": disp1, "acc_map": acc1,
               "rgb0": rgb0, "disp0": disp0, "acc0": acc0, "sigma": sigma1}
        extras.update({"z_fine": z_all, "z_samples": z_samples, "raw1": raw1, "weights1": w1})
    if want_extras:
        return ret, extras
    return ret


# --------------------------------------------------------------------------------------
# Events + losses     (reference: utils/event_utils.py:246-259, train.py:163-337,
#                      utils/math_utils.py, utils/img_utils.py:7-16, loss/imgloss.py:3-5)
# --------------------------------------------------------------------------------------

def accumulate_events(height, width, xs, ys, ps):
    """Polarity histogram: out[y,x] += p, duplicates summed.  float64 like the
    reference (np.zeros float64 + float32 dense, utils/event_utils.py:256-257)."""
    out = np.zeros((height, width), np.float64)
    np.add.at(out, (np.asarray(ys, np.int64), np.asarray(xs, np.int64)), np.asarray(ps, np.float64))
    return torch.from_numpy(out)


def event_window(ts, low_t, window_t):
    """Inclusive time window mask (model/nerf.py:165-178)."""
    upper_t = low_t + window_t
    return np.where((low_t <= ts) * (ts <= upper_t))[0], upper_t


def bright_log(x, dataset):
    """rgb2brightlog (utils/math_utils.py:4-23)."""
    if dataset in ("BeNeRF_Blender", "BeNeRF_Unreal"):
        return torch.log(x + 1e-9)
    c = x * 255
    slope = torch.log(torch.tensor(20) + 1e-9) / 20
    return torch.where(c < 20, slope * c, torch.log(c + 1e-9))


def to_gray(rgb):
    """RGB2Gray (utils/img_utils.py:7-16) -> [n,1]."""
    wts = torch.tensor([0.299, 0.587, 0.114])
    g = torch.sum(rgb * wts[None, :], dim=-1)
    return g.reshape(g.shape[0], 1)


def mse(a, b):
    return torch.mean((a - b) ** 2)


def event_loss(rgb_evt, rgb0_evt, n_pix, target_acc, channels, dataset, threshold,
               coeff_syn, coeff_real):
    """train.py:163-292.  rgb_evt/rgb0_evt [2R,C] (start rows then end rows),
    target_acc [R,1] = events_accu at the sampled pixels (float64 in the reference)."""
    def diff(img):
        a, b = img[:n_pix], img[n_pix:]
        if channels == 3:
            a, b = to_gray(a), to_gray(b)
        return bright_log(b, dataset) - bright_log(a, dataset)

    if threshold > 0:
        tgt = target_acc * torch.tensor(threshold)
        fine = mse(diff(rgb_evt), tgt) * coeff_syn
        coarse = mse(diff(rgb0_evt), tgt) * coeff_syn
    else:
        def nrm(v):
            return v / (torch.linalg.norm(v, dim=0, keepdim=True) + 1e-9)
        tgt = nrm(target_acc)
        fine = mse(nrm(diff(rgb_evt)), tgt) * coeff_real
        coarse = mse(nrm(diff(rgb0_evt)), tgt) * coeff_real
    return coarse + fine, fine, coarse


def blur_loss(rgb_map, rgb0, target, n_poses, coeff):
    """Blur synthesis = mean over the n virtual poses, then MSE (train.py:299-331).
    rgb_map/rgb0 [P*R,C] pose-major, target [R,C]."""
    R = target.shape[0]
    acc = 0
    acc0 = 0
    for j in range(n_poses):
        acc = acc + rgb_map[j * R:(j + 1) * R]
        acc0 = acc0 + rgb0[j * R:(j + 1) * R]
    acc = acc / n_poses
    acc0 = acc0 / n_poses
    fine = mse(acc, target) * coeff
    coarse = mse(acc0, target) * coeff
    return fine + coarse, fine, coarse


# --------------------------------------------------------------------------------------
# Optimiser                        (reference: model/optimize.py:36-55, train.py:343-394)
# --------------------------------------------------------------------------------------

def decayed_lr(lr0, decay_rate, global_step, lrate_decay=200):
    """train.py:355-394: lr0 * decay^(step / (lrate_decay*1000))."""
    return lr0 * (decay_rate ** (global_step / (lrate_decay * 1000)))


def adam_update(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (defaults), one tensor, in place.  step counts from 1."""
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-(lr / bc1))


def psnr(img, gt):
    """-10 log10(MSE) on [0,1] images (metrics.py:34,51-52,79-81 - see SURVEY 8d)."""
    return float(-10.0 * torch.log10(torch.mean((img - gt) ** 2)))


# --------------------------------------------------------------------------------------
# One full training step on explicit inputs (used for gradient goldens and the CPU baseline)
# --------------------------------------------------------------------------------------

class StepConfig:
    def __init__(self, H=480, W=768, fx=548.409, fy=548.409, cx=384.0, cy=240.0, channels=1,
                 n_samples=64, n_importance=64, n_poses=19, dataset="BeNeRF_Unreal",
                 threshold=0.1, coeff_syn=0.1, coeff_real=2.0, rgb_coeff=1.0, traj="spline",
                 window=0.1):
        self.H, self.W = H, W
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.channels = channels
        self.n_samples, self.n_importance, self.n_poses = n_samples, n_importance, n_poses
        self.dataset, self.threshold = dataset, threshold
        self.coeff_syn, self.coeff_real, self.rgb_coeff = coeff_syn, coeff_real, rgb_coeff
        self.traj = traj
        self.window = window

    def K(self):
        return torch.tensor([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=torch.float32)


def tone_map(p, x):
    """ColorToneMapper / LuminanceToneMapper with hidden = 0 (model/component.py:38-149): sigmoid(Linear(ReLU(Linear(x)))).
    p: {"0.weight" [width, 1], "0.bias" [width], "2.weight" [1, width], "2.bias" [1]}; x [N, 1]."""
    h = torch.relu(torch.nn.functional.linear(x, p["0.weight"], p["0.bias"]))
    return torch.sigmoid(torch.nn.functional.linear(h, p["2.weight"], p["2.bias"]))


def step_loss(cfg, p_coarse, p_fine, knots, transform, evt_ts, rgb_ts, idx_evt, idx_rgb,
              target_acc, target_rgb, draws_evt, draws_rgb, exact_pdf=False, event_crf=None, rgb_crf=None, barf=None,
              z_forced_evt=None, z_forced_rgb=None, want_extras=False, inputs_forced_evt=None, inputs_forced_rgb=None):
    """Forward of one training iteration (model/nerf.py:208-232 + train.py:163-337) on
    explicit inputs.  event_crf / rgb_crf: tone-mapper parameters applied to the rendered colours as train.py:180-192
    does when optimize_event_crf / optimize_rgb_crf are set.  Returns (loss, dict of parts)."""
    K = cfg.K()
    poses_e = trajectory_poses(knots, None, evt_ts, 2, cfg.traj)
    poses_r = trajectory_poses(knots, transform, rgb_ts, cfg.n_poses, cfg.traj)
    ret_e, ex_e = render(p_coarse, p_fine, poses_e, idx_evt, cfg.H, cfg.W, K, cfg.channels, cfg.n_samples, cfg.n_importance,
                         draws_evt, exact_pdf=exact_pdf, barf=barf, z_forced=z_forced_evt, want_extras=True,
                         mlp_inputs_forced=inputs_forced_evt)
    ret_r, ex_r = render(p_coarse, p_fine, poses_r, idx_rgb, cfg.H, cfg.W, K, cfg.channels, cfg.n_samples, cfg.n_importance,
                         draws_rgb, exact_pdf=exact_pdf, barf=barf, z_forced=z_forced_rgb, want_extras=True,
                         mlp_inputs_forced=inputs_forced_rgb)
    if event_crf is not None:
        ret_e = dict(ret_e, rgb_map=tone_map(event_crf, ret_e["rgb_map"]), rgb0=tone_map(event_crf, ret_e["rgb0"]))
    if rgb_crf is not None:
        ret_r = dict(ret_r, rgb_map=tone_map(rgb_crf, ret_r["rgb_map"]), rgb0=tone_map(rgb_crf, ret_r["rgb0"]))
    le, le_f, le_c = event_loss(ret_e["rgb_map"], ret_e["rgb0"], idx_evt.shape[0], target_acc,
                                cfg.channels, cfg.dataset, cfg.threshold, cfg.coeff_syn, cfg.coeff_real)
    lr_, lr_f, lr_c = blur_loss(ret_r["rgb_map"], ret_r["rgb0"], target_rgb, cfg.n_poses, cfg.rgb_coeff)
    loss = le + lr_
    parts = {"event": le, "event_fine": le_f, "event_coarse": le_c, "rgb": lr_,
             "rgb_fine": lr_f, "rgb_coarse": lr_c, "ret_event": ret_e, "ret_rgb": ret_r,
             "poses_evt": poses_e, "poses_rgb": poses_r}
    if want_extras:
        parts["extras_evt"], parts["extras_rgb"] = ex_e, ex_r
    return loss, parts


def event_loss_binned(rgb_evt, rgb0_evt, n_pix, target_acc_bins, channels, dataset, threshold, coeff_syn, coeff_real):
    """Dense event bins (BASELINE.json configs[4]; an extension - the reference has ONE bin per step): the event batch is
    rendered at B + 1 poses (pose-major rows [(B + 1) n_pix, C]); bin b contributes the reference's event term
    (train.py:204-292, `event_loss` above) on the colours of poses b (start) and b + 1 (end) against its own accumulated
    polarities target_acc_bins[b] [n_pix, 1].  B = 1 is `event_loss`.  Returns (sum, fine sum, coarse sum)."""
    tot = fine = coarse = 0.0
    for b in range(len(target_acc_bins)):
        rows = slice(b * n_pix, (b + 2) * n_pix)
        t, f, c = event_loss(rgb_evt[rows], rgb0_evt[rows], n_pix, target_acc_bins[b], channels, dataset, threshold, coeff_syn,
                             coeff_real)
        tot, fine, coarse = tot + t, fine + f, coarse + c
    return tot, fine, coarse


def step_loss_binned(cfg, p_coarse, p_fine, knots, transform, evt_ts, n_bins, rgb_ts, idx_evt, idx_rgb, target_acc_bins, target_rgb,
                     draws_evt, draws_rgb, exact_pdf=False, z_forced_evt=None, z_forced_rgb=None):
    """`step_loss` with dense event bins: the span evt_ts = (t_0, t_B) cut into n_bins contiguous equal bins, ONE event render
    at the n_bins + 1 boundaries (get_pose_evt(args, ts, seg_num = B + 1): linspace, model/optimize.py:58-82), the event
    term of every bin, the blur term once.  draws_evt: the draws of the (n_bins + 1) * R event rays, pose-major.  Equals the
    sum over bins of step_loss's event part on (t_b, t_b+1) with the draws of poses b, b + 1, plus one blur part
    (tests/test_oracle_golden.py checks exactly that)."""
    K = cfg.K()
    poses_e = trajectory_poses(knots, None, evt_ts, n_bins + 1, cfg.traj)
    poses_r = trajectory_poses(knots, transform, rgb_ts, cfg.n_poses, cfg.traj)
    ret_e, ex_e = render(p_coarse, p_fine, poses_e, idx_evt, cfg.H, cfg.W, K, cfg.channels, cfg.n_samples, cfg.n_importance,
                         draws_evt, exact_pdf=exact_pdf, z_forced=z_forced_evt, want_extras=True)
    ret_r, ex_r = render(p_coarse, p_fine, poses_r, idx_rgb, cfg.H, cfg.W, K, cfg.channels, cfg.n_samples, cfg.n_importance,
                         draws_rgb, exact_pdf=exact_pdf, z_forced=z_forced_rgb, want_extras=True)
    le, le_f, le_c = event_loss_binned(ret_e["rgb_map"], ret_e["rgb0"], idx_evt.shape[0], target_acc_bins, cfg.channels, cfg.dataset,
                                       cfg.threshold, cfg.coeff_syn, cfg.coeff_real)
    lr_, lr_f, lr_c = blur_loss(ret_r["rgb_map"], ret_r["rgb0"], target_rgb, cfg.n_poses, cfg.rgb_coeff)
    return le + lr_, {"event": le, "event_fine": le_f, "event_coarse": le_c, "rgb": lr_, "rgb_fine": lr_f, "rgb_coarse": lr_c,
                      "ret_event": ret_e, "ret_rgb": ret_r, "extras_evt": ex_e, "extras_rgb": ex_r}


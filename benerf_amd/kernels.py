"""Tensor-level wrappers over the C ABI (one function per entry point of
include/benerf_hip.h).  PyTorch is plumbing only: device memory, streams, autograd glue.
Every wrapper validates dtype / device / contiguity and raises on any non-zero return code.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import LossCfg, MlpGrads, MlpParams

LAYER_NAMES = tuple(["pts_linears.%d" % i for i in range(8)] + ["views_linears.0", "feature_linear", "alpha_linear",
                                                               "rgb_linear"])


try:        # the raw handle of the current stream without building a torch.cuda.Stream object (~50 launches per training step)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:      # pragma: no cover - older / newer torch without these private entry points
    _raw_stream = None


def _stream():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.BenerfHipError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


_scratch = {}


def scratch(key, n_floats, device, dtype=torch.float32):
    """Grow-only cached scratch buffers (activation-gradient tiles, dW partial sums)."""
    k = (key, str(device), dtype)
    buf = _scratch.get(k)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(int(n_floats), dtype=dtype, device=device)
        _scratch[k] = buf
    return buf


# ----------------------------------------------------------------------------- K1 trajectory
def spline_poses_fwd(knots, transform, ts, n_poses, traj=0, explicit_ts=False, out=None):
    """ts: [2] range (linspace applied in-kernel) or, with explicit_ts, [n_poses] sample times."""
    lib = _lib.load()
    poses = _new((n_poses, 3, 4), knots) if out is None else out
    assert ts.numel() == (n_poses if explicit_ts else 2)
    _lib.check(lib.benerf_spline_poses_fwd(_chk(knots, name="knots"), _chk(transform, name="transform"),
                                           _chk(ts, name="ts"), n_poses, traj, int(bool(explicit_ts)),
                                           _chk(poses), _stream()), "spline_poses_fwd")
    return poses


def spline_poses_bwd(knots, transform, ts, n_poses, traj, d_poses, explicit_ts=False):
    lib = _lib.load()
    d_knots = _new((4, 6), knots)
    d_tr = _new((1, 6), knots) if transform is not None else None
    _lib.check(lib.benerf_spline_poses_bwd(_chk(knots), _chk(transform), _chk(ts), n_poses, traj,
                                           int(bool(explicit_ts)), _chk(d_poses, name="d_poses"), d_knots.data_ptr(),
                                           None if d_tr is None else d_tr.data_ptr(), _stream()), "spline_poses_bwd")
    return d_knots, d_tr


def spline_poses_fwd_pair(knots, transform_b, ts_a, n_a, ts_b, n_b, traj=0):
    """Both trajectory evaluations of a step in one launch -> (poses_a [n_a,3,4] on the knots, poses_b [n_b,3,4] on knots +
    transform_b); ts_* are [2] ranges."""
    lib = _lib.load()
    pa, pb = _new((n_a, 3, 4), knots), _new((n_b, 3, 4), knots)
    assert ts_a.numel() == 2 and ts_b.numel() == 2
    _lib.check(lib.benerf_spline_poses_fwd_pair(_chk(knots, name="knots"), _chk(transform_b, name="transform"), _chk(ts_a, name="ts"), n_a,
                                                _chk(ts_b, name="ts"), n_b, traj, pa.data_ptr(), pb.data_ptr(), _stream()),
               "spline_poses_fwd_pair")
    return pa, pb


def spline_poses_bwd_pair(knots, transform_b, ts_a, n_a, ts_b, n_b, traj, d_poses_a, d_poses_b):
    """Both trajectory backward passes of a step in one launch -> (d_knots_a, d_knots_b, d_transform_b)."""
    lib = _lib.load()
    dk_a, dk_b, dt_b = _new((4, 6), knots), _new((4, 6), knots), _new((1, 6), knots)
    _lib.check(lib.benerf_spline_poses_bwd_pair(_chk(knots), _chk(transform_b), _chk(ts_a), n_a, _chk(ts_b), n_b, traj,
                                                _chk(d_poses_a, name="d_poses_a"), _chk(d_poses_b, name="d_poses_b"),
                                                dk_a.data_ptr(), dk_b.data_ptr(), dt_b.data_ptr(), _stream()),
               "spline_poses_bwd_pair")
    return dk_a, dk_b, dt_b


SPLINE_OPS = {"se3_2_qt": (0, 6, 7), "exp_r2q": (1, 3, 4), "log_q2r": (2, 4, 3), "q_to_R": (3, 4, 9), "taylor_B": (4, 1, 1),
              "taylor_C": (5, 1, 1), "skew_symmetric": (6, 3, 9), "q_to_Q": (7, 4, 16), "q_to_q_conj": (8, 4, 4)}


def spline_op_fwd(name, x):
    """x [n, in] -> [n, out] (include/benerf_hip.h: benerf_spline_op_fwd)."""
    lib = _lib.load()
    op, di, do = SPLINE_OPS[name]
    n = x.numel() // di
    out = _new((n, do), x)
    _lib.check(lib.benerf_spline_op_fwd(op, _chk(x, name=name + " input"), n, out.data_ptr(), _stream()), "spline_op_fwd")
    return out


def spline_op_bwd(name, x, d_out):
    lib = _lib.load()
    op, di, do = SPLINE_OPS[name]
    n = x.numel() // di
    d_in = _new((n, di), x)
    _lib.check(lib.benerf_spline_op_bwd(op, _chk(x), n, _chk(d_out, name="d_out"), d_in.data_ptr(), _stream()), "spline_op_bwd")
    return d_in


# ----------------------------------------------------------------------------- K2 rays
def rays_fwd(poses, ray_idx, H, W, fx, fy, cx, cy, ndc=True, out=None, remap=None):
    """remap: None or the TUM_VIE undistortion table [H, W, 2] float32 (include/benerf_hip.h)."""
    lib = _lib.load()
    n = poses.shape[0] * ray_idx.shape[0]
    if out is None:
        out = (_new((n, 3), poses), _new((n, 3), poses), _new((n, 3), poses))
    ro, rd, vd = out
    _lib.check(lib.benerf_rays_fwd(_chk(poses, name="poses"), _chk(ray_idx, torch.int64, "ray_idx"), poses.shape[0],
                                   ray_idx.shape[0], H, W, fx, fy, cx, cy, int(bool(ndc)), _chk(remap, name="remap"), _chk(ro),
                                   _chk(rd), _chk(vd), _stream()), "rays_fwd")
    return ro, rd, vd


def rays_bwd(poses, ray_idx, H, W, fx, fy, cx, cy, ndc, d_rays_o, d_rays_d, d_viewdirs, remap=None):
    lib = _lib.load()
    d_poses = _new((poses.shape[0], 3, 4), poses)
    n_ws = lib.benerf_rays_bwd_workspace_floats(poses.shape[0], ray_idx.shape[0])
    ws = scratch("rays_bwd", n_ws, poses.device) if n_ws else None
    _lib.check(lib.benerf_rays_bwd(_chk(poses), _chk(ray_idx, torch.int64), poses.shape[0], ray_idx.shape[0], H, W, fx,
                                   fy, cx, cy, int(bool(ndc)), _chk(remap, name="remap"), _chk(d_rays_o), _chk(d_rays_d),
                                   _chk(d_viewdirs), d_poses.data_ptr(), None if ws is None else ws.data_ptr(), n_ws, _stream()),
               "rays_bwd")
    return d_poses


def stratified_z(n_rays, n_samples, device, t_rand=None, seed=0, offset=0, near=0.0, far=1.0):
    lib = _lib.load()
    z = torch.empty((n_rays, n_samples), dtype=torch.float32, device=device)
    _lib.check(lib.benerf_stratified_z(n_rays, n_samples, near, far, _chk(t_rand, name="t_rand"), seed, offset,
                                       z.data_ptr(), _stream()), "stratified_z")
    return z


def ray_grad_reduce(z, d_pts, d_vdir_pts, d_rays_o, d_rays_d, d_viewdirs, accumulate):
    """accumulate: False / 0 overwrite, True / 1 add into all three outputs, 2 add into d_rays_d only."""
    lib = _lib.load()
    _lib.check(lib.benerf_ray_grad_reduce(z.shape[0], z.shape[1], _chk(z), _chk(d_pts), _chk(d_vdir_pts),
                                          int(accumulate), _chk(d_rays_o), _chk(d_rays_d), _chk(d_viewdirs),
                                          _stream()), "ray_grad_reduce")


# ----------------------------------------------------------------------------- K3 fused MLP
class KernelTimers:
    """Optional HIP-event brackets around the three MLP launches (same stream the kernels run
    on).  bench.py enables them to report per-kernel launch durations for the roofline."""

    def __init__(self):
        self.enabled = False
        self.records = []      # (name, n_points, start_event, end_event)
        self._open = None

    def mark(self, name, n_points):
        if not self.enabled:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if self._open is not None:
            self.records.append(self._open + (ev,))
        self._open = (name, n_points, ev) if name is not None else None

    def summary(self):
        """{name: (launches, total_ms, total_points)} - call after a device synchronize."""
        out = {}
        for name, npts, a, b in self.records:
            n, ms, pts = out.get(name, (0, 0.0, 0))
            out[name] = (n + 1, ms + a.elapsed_time(b), pts + npts)
        return out


TIMERS = KernelTimers()


def _timer(name, n_points):
    TIMERS.mark(name, n_points)


_struct_cache = {}


def _param_struct(cls, tensors_w, tensors_b):
    """Pointer table of one network's 12 weights + 12 biases (or their gradients).  The tables of a training run never
    change (parameters and gradients are views of flat buffers): validated once per distinct set of pointers."""
    # dtype / shape / stride / device belong to the key: w.t() shares its pointer with w, a reallocated tensor may recycle an address
    key = (cls,) + tuple((t.data_ptr(), t.dtype, t.shape, t.stride(), t.device) for t in tensors_w) + \
        tuple((t.data_ptr(), t.dtype, t.shape, t.stride(), t.device) for t in tensors_b)
    hit = _struct_cache.get(key)
    if hit is None:
        if len(_struct_cache) > 64:
            _struct_cache.clear()
        hit = [_chk(tensors_w[i], name="weight %d" % i) for i in range(_lib.NLAYERS)], \
              [_chk(tensors_b[i], name="bias %d" % i) for i in range(_lib.NLAYERS)]
        _struct_cache[key] = hit
    s = cls()
    for i in range(_lib.NLAYERS):
        s.w[i] = hit[0][i]
        s.b[i] = hit[1][i]
    return s


MLP_PRECISIONS = {"f32": 0, "split": 1, "split_f16bwd": 3}
_MLP_AUTO = 2
_default_precision = os.environ.get("BENERF_MLP_PRECISION", "split")
if _default_precision not in MLP_PRECISIONS:
    raise _lib.BenerfHipError("BENERF_MLP_PRECISION must be one of %s, not %r" % (sorted(MLP_PRECISIONS), _default_precision))


def is_split(mode=None):
    """True for the modes whose GEMM operands are f16 pairs (range guard, BENERF_MLP_AUTO inference launches)."""
    return (mode or _default_precision) != "f32"


def set_mlp_precision(mode):
    """Default arithmetic of the MLP launches issued through this module (the C ABI takes it per call):
    'f32': exact f32 MFMA;  'split' (default): 3 x f16 MFMA on hi/lo-split operands in the forward pass and in both backward
    GEMMs, f32 accumulate - fp32-equivalent;  'split_f16bwd': the split forward with a reduced-precision (f16-operand)
    backward, opt-in (include/benerf_hip.h)."""
    global _default_precision
    if mode not in MLP_PRECISIONS:
        raise ValueError("unknown MLP precision %r" % (mode,))
    _default_precision = mode


def get_mlp_precision():
    return _default_precision


class RangeGuard:
    """Range-guard words of the split-f16 MLP mode (include/benerf_hip.h, K3 `status`) for ONE owner: a TrainStep, or
    the per-device guard of the stand-alone / autograd entry points.  Nothing here synchronises except check():
    post() queues a copy of the words into pinned host memory, poll() looks at the last copy that has landed."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.words = torch.zeros(_lib.ST_WORDS, dtype=torch.int32, device=self.device)
        self._mirror = self._event = None
        self._posted = False        # a copy queued by post() that poll() has not looked at yet

    def gate(self, reduce_flag=None, phase=1):
        """benerf_step_gate: phase 0 writes this rank's verdict into reduce_flag[0]; phase 1 turns the (summed) verdict
        into the skip word the Adam launches read, updates the counters and clears the per-step words."""
        _lib.check(_lib.load().benerf_step_gate(self.words.data_ptr(), _chk(reduce_flag, name="reduce_flag"), phase, _stream()),
                   "step_gate")

    def post(self):
        if self._mirror is None:
            self._mirror = torch.zeros(_lib.ST_WORDS, dtype=torch.int32, device="cpu").pin_memory()   # explicit: the reference's drivers make CUDA the default tensor type
            self._event = torch.cuda.Event()
        self._mirror.copy_(self.words, non_blocking=True)
        self._event.record()
        self._posted = True

    def _raise(self, h):
        import struct
        f = lambda w: struct.unpack("f", struct.pack("I", int(w) & 0xffffffff))[0]   # noqa: E731
        if h[_lib.ST_SKIPPED]:
            raise _lib.BenerfRangeError(
                "mlp: %d of %d training steps were skipped on the device (%d in a row at the end): an activation (max %g) or a scaled "
                "gradient (max %g; inf = the loss gradient itself was not finite) left the f16 range (65504) of the split mode on this "
                "or another rank - train with mlp precision 'f32' unless the loss is NaN"
                % (h[_lib.ST_SKIPPED], h[_lib.ST_STEPS], h[_lib.ST_CONSECUTIVE], f(h[_lib.ST_LAST_ACT]), f(h[_lib.ST_LAST_GRAD])))
        raise _lib.BenerfRangeError("mlp(split): activation max %g / scaled gradient max %g left the f16 range (65504): the gradients "
                                    "of that backward pass are inf / NaN - use mlp precision 'f32'" % (f(h[_lib.ST_ACT]), f(h[_lib.ST_GRAD])))

    def poll(self, max_consecutive=1):
        """Non-blocking.  Raises BenerfRangeError when the last landed copy shows `max_consecutive` skipped steps in a row
        (TrainStep) or a violation nobody has gated (autograd path)."""
        if not self._posted or not self._event.query():
            return
        self._posted = False
        h = [int(v) & 0xffffffff for v in self._mirror.tolist()]
        lim = 0x477fe000
        # [MAX_CONSECUTIVE]: the longest run of skipped steps since the host last cleared the counters - a run that ended between
        # two posts is seen too
        if max(h[_lib.ST_CONSECUTIVE], h[_lib.ST_MAX_CONSECUTIVE]) >= max_consecutive or h[_lib.ST_ACT] >= lim or h[_lib.ST_GRAD] >= lim:
            # re-arm: the longest-run word is only cleared by the host; a caller that catches the error and trains on must not be
            # told about the SAME run at every later post (queued on this stream, behind the step that raised)
            self.words[_lib.ST_MAX_CONSECUTIVE:_lib.ST_MAX_CONSECUTIVE + 1].zero_()
            self._raise(h)

    def check(self, reset=True):
        """Synchronises.  Raises BenerfRangeError if a launch since the last reset saw a value outside the f16 range or a
        training step was skipped for it; BenerfHipError for a precision-mode mix-up of forward and backward buffers."""
        rc = _lib.load().benerf_mlp_status_check(self.words.data_ptr(), _stream())
        self._posted = False            # whatever an earlier post() copied is older than this synchronous look
        if rc != 0 and reset:
            self.words[:_lib.ST_SKIPPED_TOTAL].zero_()      # [SKIPPED_TOTAL] stays: Adam's bias correction counts applied steps
            self.words[_lib.ST_MAX_CONSECUTIVE:].zero_()
        _lib.check(rc, "mlp_status_check")


_guards = {}
_auto_words = {}


def range_guard(device):
    """The device's guard of the stand-alone and autograd entry points (TrainStep owns its own)."""
    key = str(torch.device(device))
    g = _guards.get(key)
    if g is None:
        g = _guards[key] = RangeGuard(device)
    return g


def mlp_status(device):
    """The device's default status words (uint32[BENERF_ST_WORDS] as int32, include/benerf_hip.h K3)."""
    return range_guard(device).words


def _auto_status(device):
    # BENERF_MLP_AUTO launches (inference) handle an overflow themselves by re-running in exact f32: their words are kept
    # apart, so that an overflow during e.g. render_image_test can never gate a training step
    key = str(torch.device(device))
    t = _auto_words.get(key)
    if t is None:
        t = _auto_words[key] = torch.zeros(_lib.ST_WORDS, dtype=torch.int32, device=device)
    return t


def auto_fallback_max(device):
    """Synchronises: largest |activation| that made a BENERF_MLP_AUTO launch fall back to exact f32 so far (0.0 if none)."""
    return float(_auto_status(device)[_lib.ST_ACT:_lib.ST_ACT + 1].view(torch.float32).item())


def check_mlp_status(device, reset=True):
    """Synchronises.  Raises BenerfRangeError if a split-mode launch on the device's default status words since the last
    reset saw an activation or a scaled gradient outside the f16 range."""
    range_guard(device).check(reset)


_param_generation = 0


def params_changed():
    """Tell every PackedMlp that parameter storage was rewritten behind autograd's back (fused Adam, checkpoint load)."""
    global _param_generation
    _param_generation += 1


class PackedMlp:
    """MFMA-shaped copy of one NeRF's weights; `pack()` after every optimiser step."""

    def __init__(self, weights, biases, channels):
        self.weights = list(weights)   # 12 tensors, nn.Linear layout [out,in]
        self.biases = list(biases)
        self.channels = channels
        lib = _lib.load()
        self.packed = torch.empty(lib.benerf_mlp_packed_floats(), dtype=torch.float32, device=self.weights[0].device)
        self.version = None
        self.pe_weights = None        # BARF c2f column weights of the current iteration (barf_pe_weights) or None

    def struct(self):
        s = _param_struct(MlpParams, self.weights, self.biases)
        s.pe_weights = _chk(self.pe_weights, name="pe_weights")
        return s

    def _key(self):
        return (_param_generation,) + tuple((t._version, t.data_ptr()) for t in self.weights + self.biases)

    def pack(self):
        lib = _lib.load()
        s = self.struct()
        _lib.check(lib.benerf_mlp_pack_weights(ctypes.byref(s), self.channels, self.packed.data_ptr(), _stream()),
                   "mlp_pack_weights")
        self.version = self._key()

    @staticmethod
    def pack_pair(a, b):
        """Re-packs two networks of the same channel count (a training step's coarse and fine one) in one launch."""
        lib = _lib.load()
        assert a.channels == b.channels
        sa, sb = a.struct(), b.struct()
        _lib.check(lib.benerf_mlp_pack_weights_pair(ctypes.byref(sa), a.packed.data_ptr(), ctypes.byref(sb), b.packed.data_ptr(),
                                                    a.channels, _stream()), "mlp_pack_weights_pair")
        a.version, b.version = a._key(), b._key()

    def pack_if_stale(self):
        # torch updates bump ._version; updates through the raw Adam kernel / in-place loads bump the generation
        if self._key() != self.version:
            self.pack()


def barf_pe_weights(iter_step, max_iter, start, end, device):
    """BARF coarse-to-fine weights of the positional-encoding columns (model/nerf.py:16-26,78-89) as the 96-float table of
    BenerfMlpParams.pe_weights.  The reference computes w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2, k < L, with
    alpha = (iter_step / max_iter - start) / (end - start) * L, and applies it through `embedded.view(-1, L) * weight`,
    i.e. encoding element e (sin / cos interleaved per frequency, input not included) is scaled by w[e % L]; the raw
    input concatenated in front keeps weight 1.  L = 10 for points, 4 for view directions."""
    import math
    tab = [1.0] * 96
    for base, n_enc, L in ((0, 60, 10), (64, 24, 4)):
        alpha = (iter_step / max_iter - start) / (end - start) * L
        w = [(1.0 - math.cos(math.pi * min(max(alpha - k, 0.0), 1.0))) / 2.0 for k in range(L)]
        for e in range(n_enc):
            tab[base + 3 + e] = w[e % L]
    return torch.tensor(tab, dtype=torch.float32, device=device)


def mlp_fwd(net, rays_o, rays_d, viewdirs, z, save_acts, precision=None, status=None):
    """Returns (raw, acts).  Inference launches (save_acts False) in split mode run as BENERF_MLP_AUTO: the output is
    valid even if an activation leaves the f16 range (include/benerf_hip.h).  status: the owner's range-guard words
    (default: the device's)."""
    lib = _lib.load()
    n_rays, n_samples = z.shape
    C = net.channels
    mode = precision or _default_precision
    raw = _new((n_rays, n_samples, C + 1), z)
    acts = None
    if save_acts:
        acts = torch.empty(lib.benerf_mlp_act_floats_for(n_rays * n_samples, MLP_PRECISIONS[mode]), dtype=torch.float32, device=z.device)
        acts.benerf_precision = mode
        acts.benerf_pe_weights = net.pe_weights       # the backward of THIS forward uses the same column weights
        acts.benerf_pack_version = (net.version, _param_generation)      # mlp_bwd_dw: the pack this forward ran on must still be the current one
    s = net.struct()
    code = MLP_PRECISIONS[mode]
    if mode == "split" and not save_acts:
        code = _MLP_AUTO
        status = _auto_status(z.device)
    # 'split_f16bwd' inference launches keep their own code: that mode's forward runs the UNFUSED layer sequence (its backward
    # wants the feature layer's output), training and inference alike, so that the two agree bit for bit - without the exact-f32
    # fallback launch of BENERF_MLP_AUTO (the range guard still records a violation)
    if status is None:
        status = mlp_status(z.device)
    _timer("mlp_fwd", n_rays * n_samples)
    _lib.check(lib.benerf_mlp_fwd(ctypes.byref(s), net.packed.data_ptr(), C, n_rays, n_samples, _chk(rays_o),
                                  _chk(rays_d), _chk(viewdirs), _chk(z), raw.data_ptr(), _chk(acts), code,
                                  status.data_ptr(), _stream()), "mlp_fwd")
    _timer(None, 0)
    return raw, acts


def mlp_h8_roundtrip(x, residual_log2_scale=12):
    """The lo8 codec of the split mode's saved operands on x (float32, numel a multiple of 8): returns (hi as float16 tensor,
    codes uint8, decoded float32) - include/benerf_hip.h: benerf_mlp_h8_roundtrip."""
    lib = _lib.load()
    n = x.numel()
    hi = torch.empty(n, dtype=torch.float16, device=x.device)
    codes = torch.empty(n, dtype=torch.uint8, device=x.device)
    dec = torch.empty(n, dtype=torch.float32, device=x.device)
    _lib.check(lib.benerf_mlp_h8_roundtrip(_chk(x, name="x"), n, residual_log2_scale, hi.data_ptr(), codes.data_ptr(), dec.data_ptr(),
                                           _stream()), "mlp_h8_roundtrip")
    return hi, codes, dec


def mlp_bwd_dx(net, d_raw, acts, n_rays, n_samples, slot="", status=None, d_raw_absmax=None):
    """Activation-gradient chain of one network: returns per-point (d_pts [M,3], d_vdir [M,3]) and the per-layer
    activation gradients (scratch buffer `slot`: give the two networks different slots when the weight-gradient launch
    of one is to overlap the chain of the other)."""
    lib = _lib.load()
    M = n_rays * n_samples
    dev = d_raw.device
    code = MLP_PRECISIONS[getattr(acts, "benerf_precision", _default_precision)]
    dacts = scratch("dacts" + slot, lib.benerf_mlp_dact_floats_for(M, code), dev)
    d_pts = torch.empty((M, 3), dtype=torch.float32, device=dev)
    d_vd = torch.empty((M, 3), dtype=torch.float32, device=dev)
    pe_w = getattr(acts, "benerf_pe_weights", net.pe_weights)
    s = net.struct()
    s.pe_weights = _chk(pe_w, name="pe_weights")
    _timer("mlp_bwd_dx", M)
    _lib.check(lib.benerf_mlp_bwd_dx(ctypes.byref(s), net.packed.data_ptr(), net.channels, n_rays, n_samples,
                                     _chk(d_raw, name="d_raw"), _chk(acts), dacts.data_ptr(), d_pts.data_ptr(),
                                     d_vd.data_ptr(), code, (mlp_status(dev) if status is None else status).data_ptr(),
                                     _chk(d_raw_absmax, name="d_raw_absmax"), _stream()), "mlp_bwd_dx")
    _timer(None, 0)
    return d_pts, d_vd, dacts


def mlp_bwd_dw(net, d_raw, acts, dacts, n_rays, n_samples, grad_w, grad_b, accumulate):
    """Weight / bias gradients of one network from its saved activations and activation gradients (current stream)."""
    lib = _lib.load()
    M = n_rays * n_samples
    code = MLP_PRECISIONS[getattr(acts, "benerf_precision", _default_precision)]
    ws_floats = lib.benerf_mlp_dw_workspace_floats(M)
    ws = scratch("dw_ws", ws_floats, d_raw.device)
    pe_w = getattr(acts, "benerf_pe_weights", net.pe_weights)
    g = _param_struct(MlpGrads, grad_w, grad_b)
    if code == MLP_PRECISIONS["split"] and getattr(acts, "benerf_pack_version", None) not in (None, (net.version, _param_generation)):
        # BENERF_MLP_SPLIT composes the feature / views weight gradients from G = dhv^T h7 and the LIVE W_v, W_f, b_f, while the forward
        # and dX passes used the W_c = W_v[:, :256] W_f snapshot of the pack they ran on: a fused optimiser step or a re-pack between
        # THAT forward and this call (kernels.params_changed / PackedMlp.pack) would give silently inconsistent gradients.  (Version
        # counters are not compared here: the parameters of a TrainStep are views of one flat buffer and share ONE counter - a
        # write to the trajectory knots would make both networks look stale.)
        raise _lib.BenerfHipError("mlp_bwd_dw(split): the network was re-packed or its parameters were updated between the forward pass "
                                  "that saved these activations and this backward call: the composed feature / views gradients would "
                                  "mix two parameter sets")
    s = net.struct()        # BENERF_MLP_SPLIT composes the feature / views weight gradients from dhv^T h7 and these weights
    _timer("mlp_bwd_dw", M)
    _lib.check(lib.benerf_mlp_bwd_dw(ctypes.byref(s), net.channels, n_rays, n_samples, _chk(d_raw), _chk(acts), dacts.data_ptr(),
                                     ws.data_ptr(), ws_floats, ctypes.byref(g), int(bool(accumulate)), code,
                                     _chk(pe_w, name="pe_weights"), _stream()), "mlp_bwd_dw")
    _timer(None, 0)


def mlp_bwd(net, d_raw, acts, n_rays, n_samples, grad_w, grad_b, accumulate):
    """Returns per-point (d_pts [M,3], d_vdir [M,3]); writes/accumulates weight grads."""
    d_pts, d_vd, dacts = mlp_bwd_dx(net, d_raw, acts, n_rays, n_samples)
    mlp_bwd_dw(net, d_raw, acts, dacts, n_rays, n_samples, grad_w, grad_b, accumulate)
    return d_pts, d_vd


# ----------------------------------------------------------------------------- K4 compositing
def composite_fwd(raw, z, rays_d, noise=None, noise_std=0.0, seed=0, offset=0, want=("rgb_map", "disp", "acc", "weights",
                                                                                    "depth", "sigma"), out=None):
    """out: optional dict of caller-owned tensors for some of the outputs (e.g. a row range of a batch buffer); the others in
    `want` are allocated.  Philox noise (noise None, noise_std > 0) is keyed by the ray's row INSIDE this call."""
    lib = _lib.load()
    n_rays, n_samples, c1 = raw.shape
    C = c1 - 1
    out = dict(out) if out else {}
    shapes = {"rgb_map": (n_rays, C), "disp": (n_rays,), "acc": (n_rays,), "weights": (n_rays, n_samples),
              "depth": (n_rays,), "sigma": (n_rays, n_samples)}
    for k in want:
        if k not in out:
            out[k] = _new(shapes[k], raw)
        else:
            _chk(out[k], name=k)
    p = lambda k: out[k].data_ptr() if k in out else None  # noqa: E731
    _lib.check(lib.benerf_composite_fwd(_chk(raw), _chk(z), _chk(rays_d), _chk(noise), noise_std, seed, offset, C, n_rays,
                                        n_samples, p("rgb_map"), p("disp"), p("acc"), p("weights"), p("depth"),
                                        p("sigma"), _stream()), "composite_fwd")
    return out


def composite_bwd(raw, z, rays_d, noise, noise_std, seed, offset, d_rgb_map, d_acc=None, d_depth=None, d_disp=None,
                  d_rays_d=None, accumulate=False, absmax_out=None, d_raw_out=None):
    """absmax_out: None, or a zeroed 1-element device float that receives max |d_raw| (hand it to mlp_bwd_dx; an atomic
    maximum: several calls may fold into one word).  d_raw_out: caller-owned [n_rays, n_samples, C + 1] (a row range of a batch buffer)."""
    lib = _lib.load()
    n_rays, n_samples, c1 = raw.shape
    d_raw = torch.empty_like(raw) if d_raw_out is None else d_raw_out
    _chk(d_raw, name="d_raw_out")
    assert d_raw.shape == raw.shape
    if d_rays_d is None:
        d_rays_d = _new((n_rays, 3), raw)
        accumulate = False
    _lib.check(lib.benerf_composite_bwd(_chk(raw), _chk(z), _chk(rays_d), _chk(noise), noise_std, seed, offset, c1 - 1,
                                        n_rays, n_samples, _chk(d_rgb_map), _chk(d_acc), _chk(d_depth), _chk(d_disp),
                                        d_raw.data_ptr(), d_rays_d.data_ptr(), int(bool(accumulate)), _chk(absmax_out, name="absmax_out"),
                                        _stream()),
               "composite_bwd")
    return d_raw, d_rays_d


# ----------------------------------------------------------------------------- K5 sample_pdf
def sample_pdf_merge(z_coarse, weights, n_importance, u=None, seed=0, offset=0, want_debug=False):
    lib = _lib.load()
    n_rays, n_samples = z_coarse.shape
    z_fine = _new((n_rays, n_samples + n_importance), z_coarse)
    zs = inds = None
    if want_debug:
        zs = _new((n_rays, n_importance), z_coarse)
        inds = _new((n_rays, n_importance), z_coarse, torch.int64)
    _lib.check(lib.benerf_sample_pdf_merge(_chk(z_coarse), _chk(weights), _chk(u, name="u"), seed, offset, n_rays,
                                           n_samples, n_importance, z_fine.data_ptr(), _chk(zs),
                                           _chk(inds, torch.int64), _stream()), "sample_pdf_merge")
    if want_debug:
        return z_fine, zs, inds
    return z_fine


# ----------------------------------------------------------------------------- K6 losses
def make_loss_cfg(channels, linlog, n_evt_pix, n_rgb_pix, n_poses, threshold, event_coeff, rgb_coeff,
                  n_evt_pix_global=None, n_rgb_pix_global=None):
    c = LossCfg()
    c.channels, c.linlog = channels, int(bool(linlog))
    c.n_evt_pix, c.n_rgb_pix, c.n_poses = n_evt_pix, n_rgb_pix, n_poses
    c.n_evt_pix_global = n_evt_pix if n_evt_pix_global is None else n_evt_pix_global
    c.n_rgb_pix_global = n_rgb_pix if n_rgb_pix_global is None else n_rgb_pix_global
    c.event_threshold, c.event_coeff, c.rgb_coeff = threshold, event_coeff, rgb_coeff
    return c


def loss_stats(cfg, rgb_evt, rgb0_evt, target_acc, rgb_rgb, rgb0_rgb, target_rgb):
    lib = _lib.load()
    ref = rgb_evt if rgb_evt is not None else rgb_rgb
    stats = torch.empty(_lib.LOSS_NSTATS, dtype=torch.float64, device=ref.device)
    _lib.check(lib.benerf_loss_stats(ctypes.byref(cfg), _chk(rgb_evt), _chk(rgb0_evt), _chk(target_acc), _chk(rgb_rgb),
                                     _chk(rgb0_rgb), _chk(target_rgb), stats.data_ptr(), _stream()), "loss_stats")
    return stats


def loss_grads(cfg, stats, rgb_evt, rgb0_evt, target_acc, rgb_rgb, rgb0_rgb, target_rgb, out=None, want_losses=True,
               want_grads=True):
    """Returns (losses [8] or None, [four gradient tensors or None]).  want_losses=False with stats=None: gradient-only call
    (mean-squared losses need no sums); want_grads=False: loss values only."""
    lib = _lib.load()
    ref = rgb_evt if rgb_evt is not None else rgb_rgb
    losses = torch.empty(8, dtype=torch.float32, device=ref.device) if want_losses else None
    if not want_grads:
        g = [None] * 4
    elif out is not None:
        g = list(out)
    else:
        g = [None if t is None else torch.empty_like(t) for t in (rgb_evt, rgb0_evt, rgb_rgb, rgb0_rgb)]
    _lib.check(lib.benerf_loss_grads(ctypes.byref(cfg), _chk(stats, torch.float64), _chk(rgb_evt), _chk(rgb0_evt),
                                     _chk(target_acc), _chk(rgb_rgb), _chk(rgb0_rgb), _chk(target_rgb),
                                     _chk(losses), _chk(g[0]), _chk(g[1]), _chk(g[2]), _chk(g[3]), _stream()),
               "loss_grads")
    return losses, g


# ----------------------------------------------------------------------------- K7 events
def event_accumulate(xs, ys, ps, H, W, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.zeros((H, W), dtype=torch.float32, device=xs.device)
    _lib.check(lib.benerf_event_accumulate(_chk(xs, torch.int32, "xs"), _chk(ys, torch.int32, "ys"), _chk(ps, name="ps"),
                                           xs.numel(), H, W, out.data_ptr(), _stream()), "event_accumulate")
    return out


def event_window_accumulate(xs, ys, ps, ts, low_t, upper_t, H, W, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.zeros((H, W), dtype=torch.float32, device=xs.device)
    _lib.check(lib.benerf_event_window_accumulate(_chk(xs, torch.int32), _chk(ys, torch.int32), _chk(ps),
                                                  _chk(ts, torch.float64, "ts"), xs.numel(), float(low_t),
                                                  float(upper_t), H, W, out.data_ptr(), _stream()),
               "event_window_accumulate")
    return out


def event_bin_windows(low_t, upper_t, bins):
    """[(lo_b, up_b)] of `bins` contiguous equal bins of the window [low_t, upper_t] for event_window_accumulate, whose window is
    CLOSED on both ends like the reference's single window (model/nerf.py:170: low_t <= ts <= upper_t).  Edges = the float32
    linspace the trajectory kernel evaluates the bins + 1 event poses at.  Interior bins are half-open [t_b, t_b+1) - their upper
    bound is the float64 just below the next edge - so that an event whose timestamp equals an interior edge is counted once and
    the bins add up to the one-window sum; only the last bin is closed."""
    import math
    edges = torch.linspace(float(low_t), float(upper_t), int(bins) + 1, dtype=torch.float32, device="cpu").tolist()
    return [(edges[b], edges[b + 1] if b == bins - 1 else math.nextafter(edges[b + 1], -math.inf)) for b in range(int(bins))]


def gather_rows(src, idx):
    lib = _lib.load()
    width = src.shape[-1] if src.dim() > 1 else 1
    out = torch.empty((idx.numel(), width), dtype=torch.float32, device=src.device)
    _lib.check(lib.benerf_gather_rows(_chk(src), _chk(idx, torch.int64), idx.numel(), width, out.data_ptr(), _stream()),
               "gather_rows")
    return out


def sample_pixels(n_total, count, seed, offset, device):
    """`count` distinct pixel indices in [0, n_total): deterministic in (seed, offset) (include/benerf_hip.h)."""
    lib = _lib.load()
    out = torch.empty((count,), dtype=torch.int64, device=device)
    _lib.check(lib.benerf_sample_pixels(n_total, count, seed, offset, out.data_ptr(), _stream()), "sample_pixels")
    return out


# ----------------------------------------------------------------------------- K8 optimiser
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0,
              guard_range=True, status=None):
    """guard_range: skip the update on the device when the range-guard words (`status`, default: the device's) show a
    violation or carry benerf_step_gate's skip verdict."""
    lib = _lib.load()
    st = (mlp_status(param.device) if status is None else status).data_ptr() if guard_range else None
    _lib.check(lib.benerf_adam_step(_chk(param), _chk(grad), _chk(exp_avg), _chk(exp_avg_sq), param.numel(), lr, beta1,
                                    beta2, eps, step, grad_scale, st, _stream()), "adam_step")
    params_changed()


# ----------------------------------------------------------------------------- stand-alone helpers
def sample_pdf(bins, weights, n_draws, u=None, seed=0, offset=0, want_inds=False):
    lib = _lib.load()
    n_rays, n_bins = bins.shape
    samples = _new((n_rays, n_draws), bins)
    inds = _new((n_rays, n_draws), bins, torch.int64) if want_inds else None
    _lib.check(lib.benerf_sample_pdf(_chk(bins), _chk(weights), _chk(u), seed, offset, n_rays, n_bins, n_draws,
                                     samples.data_ptr(), _chk(inds, torch.int64), _stream()), "sample_pdf")
    return (samples, inds) if want_inds else samples


def pixel_rays(c2w, i, j, fx, fy, cx, cy):
    lib = _lib.load()
    n = i.numel()
    per_ray = int(c2w.dim() == 3 and c2w.shape[0] == n and n > 1 or (c2w.dim() == 3 and c2w.shape[0] == n))
    ro, rd = _new((n, 3), c2w), _new((n, 3), c2w)
    _lib.check(lib.benerf_pixel_rays(_chk(c2w), per_ray, _chk(i, torch.int64), _chk(j, torch.int64), n, fx, fy, cx, cy,
                                     ro.data_ptr(), rd.data_ptr(), _stream()), "pixel_rays")
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    lib = _lib.load()
    n = rays_o.numel() // 3
    oo, od = torch.empty_like(rays_o), torch.empty_like(rays_d)
    _lib.check(lib.benerf_ndc_rays(H, W, focal, near, _chk(rays_o), _chk(rays_d), n, oo.data_ptr(), od.data_ptr(),
                                   _stream()), "ndc_rays")
    return oo, od


def posenc(x, n_freqs, include_input=True):
    lib = _lib.load()
    dims = x.shape[-1]
    n = x.numel() // dims
    width = (dims if include_input else 0) + 2 * dims * n_freqs
    out = _new(tuple(x.shape[:-1]) + (width,), x)
    _lib.check(lib.benerf_posenc(_chk(x), n, dims, n_freqs, int(bool(include_input)), out.data_ptr(), _stream()), "posenc")
    return out


def bright_log_fwd(x, linlog):
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(lib.benerf_bright_log_fwd(_chk(x, name="x"), x.numel(), int(bool(linlog)), out.data_ptr(), _stream()), "bright_log_fwd")
    return out


def bright_log_bwd(x, grad, linlog):
    lib = _lib.load()
    dx = torch.empty_like(x)
    _lib.check(lib.benerf_bright_log_bwd(_chk(x, name="x"), _chk(grad, name="grad"), x.numel(), int(bool(linlog)), dx.data_ptr(), _stream()),
               "bright_log_bwd")
    return dx


def rgb2gray_fwd(rgb):
    lib = _lib.load()
    n = rgb.shape[0]
    out = torch.empty((n, 1), dtype=torch.float32, device=rgb.device)
    _lib.check(lib.benerf_rgb2gray_fwd(_chk(rgb, name="rgb"), n, out.data_ptr(), _stream()), "rgb2gray_fwd")
    return out


def rgb2gray_bwd(grad, n):
    lib = _lib.load()
    d = torch.empty((n, 3), dtype=torch.float32, device=grad.device)
    _lib.check(lib.benerf_rgb2gray_bwd(_chk(grad, name="grad"), n, d.data_ptr(), _stream()), "rgb2gray_bwd")
    return d


def mse_fwd(a, b):
    lib = _lib.load()
    out = _new((), a)       # 0-dim: the reference scales the loss IN PLACE (train.py:222 `event_loss_fine *= ...`) - a view out of an autograd Function forbids that
    _lib.check(lib.benerf_mse_fwd(_chk(a), _chk(b), a.numel(), out.data_ptr(), _stream()), "mse_fwd")
    return out


def mse_bwd(a, b, grad_out, need_a=True, need_b=True):
    lib = _lib.load()
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(b) if need_b else None
    _lib.check(lib.benerf_mse_bwd(_chk(a), _chk(b), a.numel(), _chk(grad_out), _chk(da), _chk(db), _stream()), "mse_bwd")
    return da, db

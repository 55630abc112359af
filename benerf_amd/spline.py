"""Host-side mirror of the reference's spline.py public entry points.  The whole
se(3)->(q,t) / log / exp / cumulative-B-spline chain runs in ONE kernel launch (K1, one thread
per pose) with autograd support; the reference issues ~1300 ATen dispatches per call.
The single-step helpers (`*_parallel`, spline.py:16-192) are element-wise kernels over the same device
functions, differentiable through forward-mode duals."""
import torch

from . import engine
from . import kernels as K


def _knots(*poses):
    return torch.cat([p.reshape(1, 6) for p in poses], 0)


def cubic_spline_pose_unit_time(pose0, pose1, pose2, pose3, sample_time):
    """Uniform cubic B-spline in SE(3) (spline.py:247-303).  pose* are se(3) knots shaped
    [1,1,6] (any shape with 6 elements), sample_time [P] in [0,1]; returns [P,3,4].
    Exact 0 / 1 sample times are nudged by +-1e-6 like the reference, but NOT written back into
    the caller's tensor (the reference mutates its argument in place)."""
    knots = _knots(pose0, pose1, pose2, pose3)
    ts = sample_time.reshape(-1)
    return engine.SplinePoses.apply(knots, None, ts, int(ts.shape[0]), 0, True)


def linear_pose_unit_time(start_pose, end_pose, sample_time):
    """Linear translation + geodesic rotation between two knots (spline.py:305-331)."""
    knots = _knots(start_pose, start_pose, end_pose, end_pose)   # kernel reads knots 0 and 3
    ts = sample_time.reshape(-1)
    return engine.SplinePoses.apply(knots, None, ts, int(ts.shape[0]), 1, True)


class _Op(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, name):
        di = K.SPLINE_OPS[name][1]
        xc = x.detach().to(torch.float32).reshape(-1, di).contiguous()
        ctx.save_for_backward(xc)
        ctx.name, ctx.shape = name, x.shape
        return K.spline_op_fwd(name, xc)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return K.spline_op_bwd(ctx.name, xc, g.contiguous()).reshape(ctx.shape), None


def _apply(name, x, tail):
    """x [..., in] -> [..., *tail]"""
    return _Op.apply(x, name).reshape(tuple(x.shape[:-1]) + tuple(tail))


def se3_2_qt_parallel(wu):
    """se(3) [w, u] -> (q xyzw [...,4], t = V(w) u [...,3]) with the 11-term Taylor B, C (spline.py:16-26)."""
    qt = _apply("se3_2_qt", wu, (7,))
    return qt[..., :4], qt[..., 4:]


def skew_symmetric(w):
    return _apply("skew_symmetric", w, (3, 3))          # spline.py:28-34


def taylor_B(x, nth=10):
    """(1 - cos x) / x^2 as the reference's 11-term series (spline.py:46-53)."""
    if nth != 10:
        raise NotImplementedError("taylor_B: the kernels implement the reference's default nth=10")
    return _apply("taylor_B", x[..., None], ())


def taylor_C(x, nth=10):
    """(x - sin x) / x^3 as the reference's 11-term series (spline.py:55-62)."""
    if nth != 10:
        raise NotImplementedError("taylor_C: the kernels implement the reference's default nth=10")
    return _apply("taylor_C", x[..., None], ())


def exp_r2q_parallel(r, eps=1e-9):
    """rotation vector -> quaternion xyzw; series branch below half-angle eps (spline.py:79-100)."""
    if eps != 1e-9:
        raise NotImplementedError("exp_r2q_parallel: eps is fixed at the reference's default 1e-9")
    return _apply("exp_r2q", r, (4,))


def log_q2r_parallel(q, eps_theta=1e-20, eps_w=1e-10):
    """quaternion xyzw -> rotation vector, plain arctan (spline.py:167-192)."""
    if eps_theta != 1e-20 or eps_w != 1e-10:
        raise NotImplementedError("log_q2r_parallel: thresholds are fixed at the reference's defaults")
    return _apply("log_q2r", q, (3,))


def q_to_R_parallel(q):
    return _apply("q_to_R", q, (3, 3))                   # spline.py:111-118


def q_to_Q_parallel(q):
    return _apply("q_to_Q", q, (4, 4))                   # spline.py:130-138


def q_to_q_conj_parallel(q):
    return _apply("q_to_q_conj", q, (4,))                # spline.py:145-148

"""Host-side mirror of the reference's spline.py public entry points.  The whole
se(3)->(q,t) / log / exp / cumulative-B-spline chain runs in ONE kernel launch (K1, one thread
per pose) with autograd support; the reference issues ~1300 ATen dispatches per call."""
import torch

from . import engine


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

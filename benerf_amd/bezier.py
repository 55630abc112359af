"""Host-side mirror of the reference's bezier.py.

The reference function (bezier.py:22-74) is dead code and cannot run: after `sample_time.unsqueeze(-1)` its
coefficient matrix is [P,1,4] and `bezier_coeff[:,1]` raises IndexError on every call, so there are no reference outputs
to match (parity N/A).  What it evidently drafts - the cubic-Bezier sibling of
spline.cubic_spline_pose_unit_time - is implemented by the same kernel (K1, traj = 2): translation by the Bernstein
basis of compute_bezier_coefficient_mat (bezier.py:7-20), rotation by the cumulative construction of spline.py:276-295
with the cumulative Bernstein basis.  tests/test_oracle_golden.py / test_kernels_gpu.py check it against its own CPU
restatement (oracle/benerf_oracle.py: bezier_poses) and its end-point / control-point identities."""
import torch

from . import engine
from .spline import _knots


def compute_bezier_coefficient_mat(sample_time, order_curve):
    """C(n,k) (1-t)^(n-k) t^k, k = 0..n, stacked on the last axis (bezier.py:7-20); tiny host-side helper."""
    import math
    t = sample_time
    return torch.stack([math.comb(order_curve, k) * torch.pow(1 - t, order_curve - k) * torch.pow(t, k)
                        for k in range(order_curve + 1)], dim=-1)


def cubic_bezier_poses_unit_time(knot_0, knot_1, knot_2, knot_3, sample_time):
    """se(3) control poses [.., 6] x4, sample_time [P] in [0,1] -> [P,3,4]; exact 0 / 1 are nudged by +-1e-6 like
    bezier.py:24-27 (not written back into the caller's tensor)."""
    knots = _knots(knot_0, knot_1, knot_2, knot_3)
    ts = sample_time.reshape(-1)
    return engine.SplinePoses.apply(knots, None, ts, int(ts.shape[0]), 2, True)

"""`bezier.cubic_bezier_poses_unit_time` of the reference (bezier.py:22-74) is dead code that
raises IndexError on every call, so there is no behaviour to be at parity
with.  The symbol is kept so imports resolve; calling it reports that fact instead of guessing
semantics.  PARITY: N/A (unpinned by construction)."""


def cubic_bezier_poses_unit_time(*args, **kwargs):
    raise NotImplementedError(
        "the reference's bezier.cubic_bezier_poses_unit_time cannot execute (IndexError at bezier.py:56 for any "
        "input) and nothing imports it; use spline.cubic_spline_pose_unit_time (traj = spline)")

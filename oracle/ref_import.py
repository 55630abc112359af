"""Harness-only loader for the UNMODIFIED reference (/root/reference), CPU.

TEST INFRASTRUCTURE -- used only by oracle/gen_golden.py in the build
container to pin the oracle and emit golden vectors.  Nothing here (and no
reference source/bytecode) travels to the GPU box; /root/reference does not
exist there.

The four stubbed modules are non-arithmetic imports the container lacks
(cv2 / numba / h5py / imageio); see SURVEY.md Appendix A.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    """Import the reference's hot-path modules; returns a namespace object."""
    import numpy as np
    import torch

    if "cv2" not in sys.modules:
        _stub("cv2")
    if "numba" not in sys.modules:
        _stub("numba").jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    if "h5py" not in sys.modules:
        _stub("h5py", File=object)
    if "imageio" not in sys.modules:
        io = _stub("imageio")
        io.v3 = _stub("imageio.v3", imread=None, imwrite=None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import spline as ref_spline
    import run_nerf_helpers as ref_helpers
    from model import nerf as ref_nerf, optimize as ref_optimize, embedder as ref_embedder
    from model import component as ref_component
    from loss import imgloss as ref_imgloss
    from utils import math_utils as ref_math, img_utils as ref_img, event_utils as ref_event

    # utils/event_utils.py:256-257 hard-codes .to('cuda'); same sparse->dense
    # arithmetic kept on CPU for the harness.
    def _acc_cpu(out, xs, ys, ps):
        idx = torch.tensor(np.array([ys, xs]), dtype=torch.long)
        vals = torch.tensor(ps, dtype=torch.float32)
        dense = torch.sparse_coo_tensor(idx, vals, torch.Size(out.shape)).to_dense()
        return torch.from_numpy(out) + dense

    ref_event.accumulate_events_on_gpu = _acc_cpu

    ns = types.SimpleNamespace(
        spline=ref_spline, helpers=ref_helpers, nerf=ref_nerf, optimize=ref_optimize,
        embedder=ref_embedder, component=ref_component, imgloss=ref_imgloss,
        math_utils=ref_math, img_utils=ref_img, event_utils=ref_event)
    return ns


def make_args(**over):
    """Minimal args namespace the hot path reads."""
    d = dict(
        channels=1, N_samples=64, N_importance=64, use_viewdirs=True, multires=10,
        multires_views=4, i_embed=0, use_barf_c2f=False, ndc=True, dataset="BeNeRF_Unreal",
        traj="spline", num_interpolated_pose=19, rgb_crf_net_hidden=0, rgb_crf_net_width=128,
        event_crf_net_hidden=0, event_crf_net_width=128, lrate=5e-4, pose_lrate=5e-4,
        transform_lrate=5e-4, rgb_crf_lrate=5e-4, event_crf_lrate=5e-4, chunk=4096,
        max_iter=80000, event_time_window=True, random_sampling_window=True,
        accumulate_time_length=0.1, event_height=480, event_width=768,
        sampling_event_rays=1024, sampling_rgb_rays=1024, event_threshold=0.1,
        event_coeff_syn=0.1, event_coeff_real=2.0, rgb_coeff=1.0, rgb_loss=True,
        event_loss=True, optimize_nerf=True, optimize_pose=True, optimize_trans=False,
        optimize_rgb_crf=False, optimize_event_crf=False, decay_rate=0.1,
        decay_rate_pose=0.1, decay_rate_transform=0.1, decay_rate_rgb_crf=0.1,
        decay_rate_event_crf=0.1, lrate_decay=200)
    d.update(over)
    return types.SimpleNamespace(**d)

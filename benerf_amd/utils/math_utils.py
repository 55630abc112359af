"""Brightness logarithms with the reference's public names (utils/math_utils.py:4-23).

Element-wise glue for reference-style drivers that assemble the event loss themselves
(train.py:207-292); the fused training step evaluates the same two curves inside kernel K6
(benerf_amd/csrc/loss.hip: bright_log / bright_log_grad).
"""
import math

import torch

import os as _os

_FUSED = _os.environ.get("BENERF_LOSS_GLUE", "hip") != "torch"      # "torch": the element-wise torch operators below instead of the fused launches
_EPS = 1e-9
_SAFELOG_DATASETS = ("BeNeRF_Blender", "BeNeRF_Unreal")
_LINLOG_DATASETS = ("E2NeRF_Synthetic", "E2NeRF_Real")


def safe_log(x, eps=_EPS):
    """log(x + eps)"""
    return (x + eps).log()


def lin_log(color, linlog_thres=20):
    """log of 8-bit brightness, linear below `linlog_thres` with a slope that makes the curve continuous."""
    c255 = 255 * color
    slope = math.log(float(linlog_thres)) / linlog_thres      # the reference's +1e-9 vanishes in float32
    return torch.where(c255 < linlog_thres, c255 * slope, safe_log(c255))


log_func = {"safelog": safe_log, "linlog": lin_log}


class _BrightLog(torch.autograd.Function):
    """Either curve as ONE launch each way (benerf_bright_log_fwd / bwd) instead of 2 (safelog) or 6 (linlog) element-wise torch
    operators and as many autograd nodes - the reference's loop calls this four times per iteration between the render and its backward,
    with the device idle."""

    @staticmethod
    def forward(ctx, x, linlog):
        from .. import kernels as K
        xc = x.detach().contiguous()
        ctx.save_for_backward(xc)
        ctx.linlog = linlog
        return K.bright_log_fwd(xc, linlog)

    @staticmethod
    def backward(ctx, g):
        from .. import kernels as K
        (xc,) = ctx.saved_tensors
        return K.bright_log_bwd(xc, g.contiguous(), ctx.linlog), None


def rgb2brightlog(rgb, dataset_type):
    if dataset_type not in _SAFELOG_DATASETS and dataset_type not in _LINLOG_DATASETS:
        raise ValueError("unknown dataset type %r" % (dataset_type,))
    linlog = dataset_type in _LINLOG_DATASETS
    # the lin-log curve is six element-wise torch operators (and eight autograd nodes): one launch each way instead - 0.7-0.85 ms per
    # iteration of the reference-shaped loop at C4 / C5; log(x + eps) is two operators, cheaper than a Python autograd node (measured at
    # C2: profiles/r06_loss_glue_ab.log), and stays torch
    if linlog and torch.is_tensor(rgb) and rgb.is_cuda and rgb.dtype == torch.float32 and _FUSED:
        return _BrightLog.apply(rgb, True)
    return lin_log(rgb) if linlog else safe_log(rgb)

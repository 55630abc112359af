"""Brightness logarithms with the reference's public names (utils/math_utils.py:4-23).

Element-wise glue for reference-style drivers that assemble the event loss themselves
(train.py:207-292); the fused training step evaluates the same two curves inside kernel K6
(benerf_amd/csrc/loss.hip: bright_log / bright_log_grad).
"""
import math

import torch

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


def rgb2brightlog(rgb, dataset_type):
    if dataset_type in _SAFELOG_DATASETS:
        return safe_log(rgb)
    if dataset_type in _LINLOG_DATASETS:
        return lin_log(rgb)
    raise ValueError("unknown dataset type %r" % (dataset_type,))

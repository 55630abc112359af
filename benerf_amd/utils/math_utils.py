"""Host-side mirror of utils/math_utils.py (brightness logarithms).  Element-wise glue for
reference-style drivers that assemble the loss themselves; the fused training step evaluates
the same formulas inside kernel K6 (benerf_amd/csrc/loss.hip)."""
import torch


def safe_log(x, eps=1e-9):
    return torch.log(x + eps)


def lin_log(color, linlog_thres=20):
    color = color * 255
    lin_slope = safe_log(torch.tensor(linlog_thres, device=color.device)) / linlog_thres
    return torch.where(color < linlog_thres, lin_slope * color, safe_log(color))


log_func = {"safelog": safe_log, "linlog": lin_log}


def rgb2brightlog(rgb, dataset_type):
    """(utils/math_utils.py:18-23)"""
    if dataset_type in ["BeNeRF_Blender", "BeNeRF_Unreal"]:
        return log_func["safelog"](rgb)
    elif dataset_type in ["E2NeRF_Synthetic", "E2NeRF_Real"]:
        return log_func["linlog"](rgb)
    raise ValueError("unknown dataset type %r" % (dataset_type,))

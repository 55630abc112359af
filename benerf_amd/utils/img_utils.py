"""The parts of the reference's utils/img_utils.py that sit on the path: the luma conversion used by
the event loss for colour scenes (utils/img_utils.py:7-16) and the 8-bit quantiser used when test
renders are written (utils/img_utils.py:19-20).  Kernel K6 applies the same luma weights."""
import numpy as np
import torch

import os as _os

LUMA = (0.299, 0.587, 0.114)
_FUSED = _os.environ.get("BENERF_LOSS_GLUE", "hip") != "torch"


class _Luma(torch.autograd.Function):
    """[n,3] -> [n,1] in one launch each way (benerf_rgb2gray_fwd / bwd) instead of three slices, two products and two sums."""

    @staticmethod
    def forward(ctx, rgb):
        from .. import kernels as K
        ctx.n = rgb.shape[0]
        return K.rgb2gray_fwd(rgb.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        from .. import kernels as K
        return K.rgb2gray_bwd(g.contiguous().view(-1), ctx.n)


class RGB2Gray:
    """[n,3] colour -> [n,1] luma, summed left to right like the reference."""

    def __init__(self) -> None:
        self.rgb_weight = torch.tensor(LUMA)

    def __call__(self, rgb):
        if rgb.is_cuda and rgb.dtype == torch.float32 and rgb.dim() == 2 and rgb.shape[1] == 3 and _FUSED:
            return _Luma.apply(rgb)
        w = self.rgb_weight.to(device=rgb.device, dtype=rgb.dtype)
        luma = (rgb[:, 0] * w[0] + rgb[:, 1] * w[1]) + rgb[:, 2] * w[2]
        return luma.unsqueeze(-1)


def to8bit(x) -> np.ndarray:
    return np.asarray(np.clip(x, 0, 1) * 255).astype(np.uint8)


def rgb2gray(x) -> np.ndarray:
    return (x @ np.asarray(LUMA)).astype(np.uint8)

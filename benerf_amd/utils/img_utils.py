"""The parts of the reference's utils/img_utils.py that sit on the path: the luma conversion used by
the event loss for colour scenes (utils/img_utils.py:7-16) and the 8-bit quantiser used when test
renders are written (utils/img_utils.py:19-20).  Kernel K6 applies the same luma weights."""
import numpy as np
import torch

LUMA = (0.299, 0.587, 0.114)


class RGB2Gray:
    """[n,3] colour -> [n,1] luma, summed left to right like the reference."""

    def __init__(self) -> None:
        self.rgb_weight = torch.tensor(LUMA)

    def __call__(self, rgb):
        w = self.rgb_weight.to(device=rgb.device, dtype=rgb.dtype)
        luma = (rgb[:, 0] * w[0] + rgb[:, 1] * w[1]) + rgb[:, 2] * w[2]
        return luma.unsqueeze(-1)


def to8bit(x) -> np.ndarray:
    return np.asarray(np.clip(x, 0, 1) * 255).astype(np.uint8)


def rgb2gray(x) -> np.ndarray:
    return (x @ np.asarray(LUMA)).astype(np.uint8)

"""Host-side mirror of utils/img_utils.py (the parts on the path: RGB2Gray, to8bit)."""
import numpy as np
import torch


class RGB2Gray:
    """0.299 r + 0.587 g + 0.114 b -> [n,1]  (utils/img_utils.py:7-16)."""

    def __init__(self) -> None:
        self.rgb_weight = torch.tensor([0.299, 0.587, 0.114])

    def __call__(self, rgb):
        x = torch.sum(rgb * self.rgb_weight.to(rgb.device)[None, :], axis=-1)
        return x.reshape(x.shape[0], 1)


def to8bit(x) -> np.ndarray:
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def rgb2gray(x) -> np.ndarray:
    return np.sum(x * np.array((0.299, 0.587, 0.114)), axis=-1).astype(np.uint8)

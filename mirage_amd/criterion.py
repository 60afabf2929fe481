"""Masked reconstruction losses with the reference's constructor and call signature
(mirage/criterion.py: MaskedCrossEntropyLoss :11, MaskedMSELoss :70): forward(input, target, mask=None).
input is the image-layout prediction [B, C, H, W] (bf16 or fp32, straight from the output adapter);
mask is the per-token task mask [B, (H/P)*(W/P)] with 1 = masked = counted in the loss.
"""
from typing import Union

import torch
import torch.nn as nn

from . import _lib as L
from .engine import MaskedLossFn


class MaskedCrossEntropyLoss(nn.Module):
    def __init__(self, patch_size: Union[tuple, list], stride: int = 1, label_smoothing: float = 0.0):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.scale_factor = patch_size[0] // stride
        self.label_smoothing = label_smoothing
        self.epoch = 0
        if not (0.0 <= float(label_smoothing) < 1.0):
            raise ValueError(f"MaskedCrossEntropyLoss: label_smoothing = {label_smoothing!r} must be in [0, 1)")

    def forward(self, input, target, mask=None):
        return MaskedLossFn.apply(input, target, mask, L.MRG_LOSS_CE, self.scale_factor, float(self.label_smoothing))


class MaskedMSELoss(nn.Module):
    def __init__(self, patch_size: int = 16, stride: int = 1, norm_pix=False):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        if isinstance(patch_size, int):
            self.scale_factor = patch_size // stride
        elif isinstance(patch_size, (list, tuple)):
            self.scale_factor = patch_size[0] // stride
        self.norm_pix = norm_pix
        self.epoch = 0

    def forward(self, input, target, mask=None):
        return MaskedLossFn.apply(input, target, mask, L.MRG_LOSS_MSE, self.scale_factor, 0.0, bool(self.norm_pix))

"""2-D fusion layers in front of the lift -- mirror of
mmdet3d/models/semantic_net/layers.py (LayerNorm :11-31, AddFusionLift :109-152,
CatFusionLift :154-199, build_fusion_layer_lift :202-208).  detectron2's
``Conv2d`` is used there without norm / activation, i.e. as ``nn.Conv2d``;
parameter names are the same (``input_proj_1.0.weight`` = LayerNorm,
``input_proj_1.1.weight`` = conv).  Pinned by vectors generated from the
reference file itself (oracle/tools/gen_golden_align_net.py)."""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm(nn.Module):
    """Channel LayerNorm for (B, C, H, W) inputs (ConvNeXt style)."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


def _proj(cin, cout):
    return nn.Sequential(LayerNorm(cin), nn.Conv2d(cin, cout, kernel_size=1))


class AddFusionLift(nn.Module):
    def __init__(self, in_channels_1, in_channels_2, out_channels):
        super().__init__()
        self.input_proj_1 = _proj(in_channels_1, out_channels)
        self.input_proj_2 = _proj(in_channels_2, out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, y, spatial_shape: Tuple[int, int]):
        x = self.input_proj_1(x)
        y = F.interpolate(self.input_proj_2(y.contiguous()), size=spatial_shape,
                          mode='bilinear', align_corners=False)
        return self.relu(x + y)


class CatFusionLift(nn.Module):
    """LN + 1x1 conv of cat(x1, x2) -> C/4 channels, LN + 1x1 conv of x2 ->
    3C/4, concatenated, ReLU; both inputs are first resized to the lift's
    feature-map shape."""

    def __init__(self, in_channels_1, in_channels_2, out_channels):
        super().__init__()
        p1 = out_channels // 4
        self.input_proj_1 = _proj(in_channels_1 + in_channels_2, p1)
        self.input_proj_2 = _proj(in_channels_2, out_channels - p1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x1, x2, spatial_shape: Tuple[int, int]):
        spatial_shape = tuple(spatial_shape)
        if tuple(x2.shape[-2:]) != spatial_shape:
            x2 = F.interpolate(x2.contiguous(), size=spatial_shape, mode='bilinear',
                               align_corners=False)
        if tuple(x1.shape[-2:]) != spatial_shape:
            x1 = F.interpolate(x1.contiguous(), size=spatial_shape, mode='bilinear',
                               align_corners=False)
        y1 = self.input_proj_1(torch.cat([x1, x2], dim=1))
        y2 = self.input_proj_2(x2)
        return self.relu(torch.cat([y1, y2], dim=1))


def build_fusion_layer_lift(fusion_type, in_channels_1, in_channels_2, out_channels):
    if fusion_type == 'add_fusion':
        return AddFusionLift(in_channels_1, in_channels_2, out_channels)
    if fusion_type == 'cat_fusion':
        return CatFusionLift(in_channels_1, in_channels_2, out_channels)
    raise ValueError('Unknown fusion type: {}'.format(fusion_type))

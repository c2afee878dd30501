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

from ... import vit_ops
from .._native_cache import NativeCacheMixin
from ... import half as _half


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


class CatFusionLift(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip',)

    """LN + 1x1 conv of cat(x1, x2) -> C/4 channels, LN + 1x1 conv of x2 ->
    3C/4, concatenated, ReLU; both inputs are first resized to the lift's
    feature-map shape."""

    def __init__(self, in_channels_1, in_channels_2, out_channels):
        super().__init__()
        p1 = out_channels // 4
        self.input_proj_1 = _proj(in_channels_1 + in_channels_2, p1)
        self.input_proj_2 = _proj(in_channels_2, out_channels - p1)
        self.relu = nn.ReLU(inplace=True)
        # veon_amd extension: torch.bfloat16 = at inference run the two
        # LN + 1x1-conv branches token-wise on the MFMA kernels (LayerNorm rows ->
        # GEMM with bias + ReLU) and return a channels-last bf16 map (as an NCHW
        # view), which the lift consumes as half-precision feature rows without
        # any copy.  None = the reference's fp32 PyTorch formulation.
        self.hip_dtype = None

    def _hip_ok(self, x1, x2):
        c1 = self.input_proj_1[1]
        c2 = self.input_proj_2[1]
        return (self.hip_dtype == _half.dtype() and x1.is_cuda and not self.training
                and not torch.is_grad_enabled() and c1.in_channels % 64 == 0
                and c2.in_channels % 64 == 0 and c1.out_channels % 4 == 0
                and c2.out_channels % 4 == 0 and c1.in_channels <= 2048)

    def _hip_forward(self, x1, x2):
        """x1, x2 already at the lift's map size, (N, C, H, W) fp32."""
        if '_hip' not in self.__dict__:
            def pack(seq):
                ln, conv = seq
                return (ln.weight.detach().float().contiguous(),
                        ln.bias.detach().float().contiguous(), ln.eps,
                        vit_ops.to_bf16(conv.weight.detach().float()
                                        .view(conv.out_channels, -1)),
                        conv.bias.detach().float().contiguous())
            self.__dict__['_hip'] = (pack(self.input_proj_1), pack(self.input_proj_2))
        (g1, b1, e1, w1, c1), (g2, b2, e2, w2, c2) = self.__dict__['_hip']
        N, _, H, W = x2.shape
        r2 = x2.float().permute(0, 2, 3, 1).reshape(N * H * W, -1)
        r1 = x1.float().permute(0, 2, 3, 1).reshape(N * H * W, -1)
        rc = torch.cat([r1, r2], dim=1)
        y1 = vit_ops.linear(vit_ops.layernorm(rc, g1, b1, e1), w1, c1,
                            vit_ops.EPI_AFFINE_RELU)
        y2 = vit_ops.linear(vit_ops.layernorm(r2.contiguous(), g2, b2, e2), w2, c2,
                            vit_ops.EPI_AFFINE_RELU)
        out = torch.cat([y1, y2], dim=1).view(N, H, W, -1)       # NHWC bf16
        return out.permute(0, 3, 1, 2)                            # NCHW view of it

    @staticmethod
    def _resize(x, size):
        # same op as the reference's F.interpolate(x.contiguous(), ...); on a GPU
        # the channels-last kernel is ~15x faster for these many-channel, tiny
        # maps (1.3 ms -> 0.09 ms for 6 x 768 x 8 x 22) and its output is already
        # the token layout the MFMA path wants
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        else:
            x = x.contiguous()
        return F.interpolate(x, size=size, mode='bilinear', align_corners=False)

    def forward(self, x1, x2, spatial_shape: Tuple[int, int]):
        spatial_shape = tuple(spatial_shape)
        if tuple(x2.shape[-2:]) != spatial_shape:
            x2 = self._resize(x2, spatial_shape)
        if tuple(x1.shape[-2:]) != spatial_shape:
            x1 = self._resize(x1, spatial_shape)
        if self._hip_ok(x1, x2):
            return self._hip_forward(x1, x2)
        y1 = self.input_proj_1(torch.cat([x1, x2], dim=1))
        y2 = self.input_proj_2(x2)
        return self.relu(torch.cat([y1, y2], dim=1))


def build_fusion_layer_lift(fusion_type, in_channels_1, in_channels_2, out_channels):
    if fusion_type == 'add_fusion':
        return AddFusionLift(in_channels_1, in_channels_2, out_channels)
    if fusion_type == 'cat_fusion':
        return CatFusionLift(in_channels_1, in_channels_2, out_channels)
    raise ValueError('Unknown fusion type: {}'.format(fusion_type))

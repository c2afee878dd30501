"""DepthAnythingV2 depth estimator -- mirror of
mmdet3d/models/depth_anything/dpt.py (DPTHead :39-150, DepthAnythingV2Adaptor
:226-263) and util/blocks.py.  The encoder's dense contractions run on MFMA
(``dinov2.py``); the DPT convolution head stays plain PyTorch / MIOpen
(SURVEY 2 #6).  Same module / parameter names as the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..builder import register_neck
from .dinov2 import DINOv2Adaptor


class ResidualConvUnit(nn.Module):
    def __init__(self, features, bn=False):
        super().__init__()
        self.bn = bn
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        if bn:
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)

    def forward(self, x):
        out = self.conv1(F.relu(x))
        if self.bn:
            out = self.bn1(out)
        out = self.conv2(F.relu(out))
        if self.bn:
            out = self.bn2(out)
        return out + x


class FeatureFusionBlock(nn.Module):
    """util/blocks.py:86-148 (expand=False, align_corners=True)."""

    def __init__(self, features, bn=False, size=None):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features, bn)
        self.resConfUnit2 = ResidualConvUnit(features, bn)
        self.size = size

    def forward(self, *xs, size=None):
        out = xs[0]
        if len(xs) == 2:
            out = out + self.resConfUnit1(xs[1])
        out = self.resConfUnit2(out)
        if size is None and self.size is None:
            kw = dict(scale_factor=2)
        else:
            kw = dict(size=self.size if size is None else size)
        out = F.interpolate(out, **kw, mode='bilinear', align_corners=True)
        return self.out_conv(out)


def _make_scratch(in_shape, out_shape):
    scratch = nn.Module()
    for i, c in enumerate(in_shape, 1):
        setattr(scratch, 'layer%d_rn' % i,
                nn.Conv2d(c, out_shape, 3, 1, 1, bias=False))
    return scratch


class DPTHead(nn.Module):
    def __init__(self, in_channels, features=256, use_bn=False,
                 out_channels=[256, 512, 1024, 1024], use_clstoken=False):
        super().__init__()
        self.use_clstoken = use_clstoken
        self.projects = nn.ModuleList(
            [nn.Conv2d(in_channels, oc, 1, 1, 0) for oc in out_channels])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(out_channels[0], out_channels[0], 4, 4, 0),
            nn.ConvTranspose2d(out_channels[1], out_channels[1], 2, 2, 0),
            nn.Identity(),
            nn.Conv2d(out_channels[3], out_channels[3], 3, 2, 1)])
        if use_clstoken:
            self.readout_projects = nn.ModuleList([
                nn.Sequential(nn.Linear(2 * in_channels, in_channels), nn.GELU())
                for _ in out_channels])
        self.scratch = _make_scratch(out_channels, features)
        self.scratch.stem_transpose = None
        for i in (1, 2, 3, 4):
            setattr(self.scratch, 'refinenet%d' % i,
                    FeatureFusionBlock(features, use_bn))
        self.scratch.output_conv1 = nn.Conv2d(features, features // 2, 3, 1, 1)
        self.scratch.output_conv2 = nn.Sequential(
            nn.Conv2d(features // 2, 32, 3, 1, 1), nn.ReLU(True),
            nn.Conv2d(32, 1, 1, 1, 0), nn.Sigmoid())

    def forward(self, out_features, patch_h, patch_w):
        out = []
        for i, x in enumerate(out_features):
            if self.use_clstoken:
                x, cls_token = x[0], x[1]
                readout = cls_token.unsqueeze(1).expand_as(x)
                x = self.readout_projects[i](torch.cat((x, readout), -1))
            else:
                x = x[0]
            x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], patch_h, patch_w)
            out.append(self.resize_layers[i](self.projects[i](x)))
        l1, l2, l3, l4 = out
        s = self.scratch
        l1, l2, l3, l4 = s.layer1_rn(l1), s.layer2_rn(l2), s.layer3_rn(l3), s.layer4_rn(l4)
        p4 = s.refinenet4(l4, size=l3.shape[2:])
        p3 = s.refinenet3(p4, l3, size=l2.shape[2:])
        p2 = s.refinenet2(p3, l2, size=l1.shape[2:])
        p1 = s.refinenet1(p2, l1)
        out = s.output_conv1(p1)
        out = F.interpolate(out, (int(patch_h * 14), int(patch_w * 14)),
                            mode='bilinear', align_corners=True)
        return s.output_conv2(out)


_TAPS = {'vits': [2, 5, 8, 11], 'vitb': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}


@register_neck()
class DepthAnythingV2Adaptor(nn.Module):
    """``forward(x (B,3,H,W)) -> {'metric_depth': (B,H,W)}`` (dpt.py:256-263);
    kwargs as configs/veon/*: encoder, features, out_channels, max_depth,
    use_lora, lora_r."""

    def __init__(self, encoder='vitl', features=256,
                 out_channels=[256, 512, 1024, 1024], use_bn=False,
                 use_clstoken=False, max_depth=20.0, use_lora=True, lora_r=8):
        super().__init__()
        self.intermediate_layer_idx = dict(_TAPS)
        self.max_depth = max_depth
        self.encoder = encoder
        self.pretrained = DINOv2Adaptor(encoder, lora_r=lora_r if use_lora else -1)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn,
                                  out_channels=out_channels,
                                  use_clstoken=use_clstoken)
        # veon_amd extension: run the DPT head's convolutions (PyTorch/MIOpen)
        # under autocast at inference, e.g. torch.bfloat16 for BASELINE config 3
        # ("bf16").  None = fp32, the reference's numerics.
        self.head_dtype = None

    def encode(self, x):
        """The four intermediate (patch tokens, class token) pairs the head
        consumes (dpt.py:258-259)."""
        return self.pretrained.get_intermediate_layers(
            x, self.intermediate_layer_idx[self.encoder], return_class_token=True)

    def decode(self, feats, patch_h, patch_w):
        if (self.head_dtype is not None and feats[0][0].is_cuda
                and not torch.is_grad_enabled()):
            with torch.autocast('cuda', dtype=self.head_dtype):
                depth = self.depth_head(feats, patch_h, patch_w)
            depth = depth.float()
        else:
            depth = self.depth_head(feats, patch_h, patch_w)
        return depth * self.max_depth

    def forward(self, x):
        patch_h, patch_w = x.shape[-2] // 14, x.shape[-1] // 14
        depth = self.decode(self.encode(x), patch_h, patch_w)
        return {'metric_depth': depth.squeeze(1)}

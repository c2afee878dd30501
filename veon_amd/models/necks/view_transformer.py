"""BEVDet-style view transformers -- mirror of
mmdet3d/models/necks/view_transformer.py.

Plugin classes (same names / constructor kwargs / forward contracts):
``LSSViewTransformer`` (:15-318), ``LSSViewTransformerBEVDepth`` (:694-791),
``LSSViewTransformerBEVStereo`` (:794-801).  The lift itself (geometry, index
preparation, bev_pool_v2) comes from ``LSSCore`` and runs on the HIP kernels;
the ``DepthNet`` convolution stack stays plain PyTorch (MIOpen) -- SURVEY 2 #5
marks it out of scope beyond "stays torch".
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ..builder import register_neck
from .lss_core import LSSCore

try:  # pragma: no cover - absent from the build image
    from mmcv.cnn import build_conv_layer
except Exception:
    build_conv_layer = None

try:  # pragma: no cover
    from mmdet.models.backbones.resnet import BasicBlock
except Exception:

    class BasicBlock(nn.Module):
        """Two 3x3 conv + BN residual block with mmdet's parameter names
        (conv1/bn1/conv2/bn2/downsample) so checkpoints load unchanged."""

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            identity = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + identity)


@register_neck()
class LSSViewTransformer(LSSCore):
    r"""Lift-Splat-Shoot view transformer on BEVPoolv2
    (`LSS <https://arxiv.org/abs/2008.05711>`_,
    `BEVPoolv2 <https://arxiv.org/abs/2211.17111>`_).

    Args: as the reference (view_transformer.py:23-38): ``grid_config``
    (x/y/z/depth -> (lower, upper, step)), ``input_size`` (H, W),
    ``downsample``, ``in_channels``, ``out_channels``, ``accelerate`` (cache the
    ranks; calibration must then be constant), ``sid``, ``collapse_z``.
    """

    def __init__(self, grid_config, input_size, downsample=16, in_channels=512,
                 out_channels=64, accelerate=False, sid=False, collapse_z=True):
        super().__init__()
        self._init_lss(grid_config, input_size, downsample, out_channels,
                       accelerate, sid, collapse_z)
        self.in_channels = in_channels
        self.depth_net = nn.Conv2d(in_channels, self.D + self.out_channels,
                                   kernel_size=1, padding=0)

    def forward(self, input):
        """input = (img_feat (B,N,C,H,W), sensor2ego, ego2global, intrins,
        post_rots, post_trans, bda) -> (bev_feat, depth)  (:297-315)."""
        x = input[0]
        B, N, C, H, W = x.shape
        x = self.depth_net(x.view(B * N, C, H, W))
        depth = x[:, :self.D, ...].softmax(dim=1)
        tran_feat = x[:, self.D:self.D + self.out_channels, ...]
        return self.view_transform(input, depth, tran_feat)

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda):
        return None


class _ASPPModule(nn.Module):

    def __init__(self, inplanes, planes, kernel_size, padding, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size, 1, padding,
                                     dilation, bias=False)
        self.bn = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()
        nn.init.kaiming_normal_(self.atrous_conv.weight)

    def forward(self, x):
        return self.relu(self.bn(self.atrous_conv(x)))


class ASPP(nn.Module):
    """Atrous spatial pyramid (view_transformer.py:354-428), dilations 1/6/12/18
    + image pooling, 1x1 fuse, dropout 0.5."""

    def __init__(self, inplanes, mid_channels=256):
        super().__init__()
        self.aspp1 = _ASPPModule(inplanes, mid_channels, 1, 0, 1)
        self.aspp2 = _ASPPModule(inplanes, mid_channels, 3, 6, 6)
        self.aspp3 = _ASPPModule(inplanes, mid_channels, 3, 12, 12)
        self.aspp4 = _ASPPModule(inplanes, mid_channels, 3, 18, 18)
        self.global_avg_pool = nn.Sequential(
            nn.AdaptiveAvgPool2d((1, 1)),
            nn.Conv2d(inplanes, mid_channels, 1, stride=1, bias=False),
            nn.BatchNorm2d(mid_channels), nn.ReLU())
        self.conv1 = nn.Conv2d(mid_channels * 5, inplanes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        for m in (self.global_avg_pool[1], self.conv1):
            nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        pooled = F.interpolate(self.global_avg_pool(x), size=x.shape[2:],
                               mode='bilinear', align_corners=True)
        x = torch.cat((self.aspp1(x), self.aspp2(x), self.aspp3(x),
                       self.aspp4(x), pooled), dim=1)
        return self.dropout(self.relu(self.bn1(self.conv1(x))))


class Mlp(nn.Module):

    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.ReLU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class SELayer(nn.Module):

    def __init__(self, channels, act_layer=nn.ReLU, gate_layer=nn.Sigmoid):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = act_layer()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = gate_layer()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class DepthNet(nn.Module):
    """Camera-aware depth/context head (view_transformer.py:470-630): same
    sub-module names, so BEVDepth / BEVStereo checkpoints map one to one.
    ``use_dcn`` needs mmcv's deformable conv.  The stereo cost volume
    (``stereo=True``, :500-512, :543-601) is plain PyTorch (grid_sample), off
    VEON's path."""

    def __init__(self, in_channels, mid_channels, context_channels,
                 depth_channels, use_dcn=True, use_aspp=True, with_cp=False,
                 stereo=False, bias=0.0, aspp_mid_channels=-1):
        super().__init__()
        self.reduce_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, 3, 1, 1),
            nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.context_conv = nn.Conv2d(mid_channels, context_channels, 1, 1, 0)
        self.bn = nn.BatchNorm1d(27)
        self.depth_mlp = Mlp(27, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        self.context_mlp = Mlp(27, mid_channels, mid_channels)
        self.context_se = SELayer(mid_channels)
        depth_in = mid_channels
        downsample = None
        if stereo:
            depth_in += depth_channels
            downsample = nn.Conv2d(depth_in, mid_channels, 1, 1, 0)
            cv = []
            for _ in range(2):
                cv += [nn.Conv2d(depth_channels, depth_channels, 3, 2, 1),
                       nn.BatchNorm2d(depth_channels)]
            self.cost_volumn_net = nn.Sequential(*cv)
            self.bias = bias
        layers = [BasicBlock(depth_in, mid_channels, downsample=downsample),
                  BasicBlock(mid_channels, mid_channels),
                  BasicBlock(mid_channels, mid_channels)]
        if use_aspp:
            layers.append(ASPP(mid_channels, mid_channels
                               if aspp_mid_channels < 0 else aspp_mid_channels))
        if use_dcn:
            if build_conv_layer is None:
                raise RuntimeError(
                    'DepthNet(use_dcn=True) needs mmcv.ops deformable conv; '
                    'pass depthnet_cfg=dict(use_dcn=False) without mmcv')
            layers.append(build_conv_layer(cfg=dict(
                type='DCN', in_channels=mid_channels, out_channels=mid_channels,
                kernel_size=3, padding=1, groups=4, im2col_step=128)))
        layers.append(nn.Conv2d(mid_channels, depth_channels, 1, 1, 0))
        self.depth_conv = nn.Sequential(*layers)
        self.with_cp = with_cp
        self.depth_channels = depth_channels

    def gen_grid(self, metas, B, N, D, H, W, hi, wi):
        """Sampling grid of the previous frame for every (depth, pixel) of the
        current one (:543-571): un-augment, lift with the candidate depth, move
        key -> sweep sensor, project, re-augment, normalise to [-1, 1]."""
        frustum = metas['frustum']
        points = frustum - metas['post_trans'].view(B, N, 1, 1, 1, 3)
        points = torch.inverse(metas['post_rots']).view(B, N, 1, 1, 1, 3, 3) \
            .matmul(points.unsqueeze(-1))
        points = torch.cat(
            (points[..., :2, :] * points[..., 2:3, :], points[..., 2:3, :]), 5)
        rots = metas['k2s_sensor'][:, :, :3, :3].contiguous()
        trans = metas['k2s_sensor'][:, :, :3, 3].contiguous()
        combine = rots.matmul(torch.inverse(metas['intrins']))
        points = combine.view(B, N, 1, 1, 1, 3, 3).matmul(points)
        points = points + trans.view(B, N, 1, 1, 1, 3, 1)
        neg_mask = points[..., 2, 0] < 1e-3
        points = metas['intrins'].view(B, N, 1, 1, 1, 3, 3).matmul(points)
        points = points[..., :2, :] / points[..., 2:3, :]
        points = metas['post_rots'][..., :2, :2].view(B, N, 1, 1, 1, 2, 2) \
            .matmul(points).squeeze(-1)
        points = points + metas['post_trans'][..., :2].view(B, N, 1, 1, 1, 2)
        px = points[..., 0] / (wi - 1.0) * 2.0 - 1.0
        py = points[..., 1] / (hi - 1.0) * 2.0 - 1.0
        px = torch.where(neg_mask, torch.full_like(px, -2), px)
        py = torch.where(neg_mask, torch.full_like(py, -2), py)
        return torch.stack([px, py], dim=-1).view(B * N, D * H, W, 2)

    def calculate_cost_volumn(self, metas):
        """Group-wise L1 matching cost between the current features and the
        previous frame's warped to every depth hypothesis, softmax over depth
        (:573-601)."""
        prev, curr = metas['cv_feat_list']
        group_size = 4
        _, c, hf, wf = curr.shape
        hi, wi = hf * 4, wf * 4
        B, N, _ = metas['post_trans'].shape
        D, H, W, _ = metas['frustum'].shape
        grid = self.gen_grid(metas, B, N, D, H, W, hi, wi).to(curr.dtype)
        prev = prev.view(B * N, -1, H, W)
        curr = curr.view(B * N, -1, H, W)
        cost = 0
        wrap_prev = None
        for fid in range(curr.shape[1] // group_size):
            sl = slice(fid * group_size, (fid + 1) * group_size)
            wrap_prev = F.grid_sample(prev[:, sl], grid, align_corners=True,
                                      padding_mode='zeros')
            diff = curr[:, sl].unsqueeze(2) - wrap_prev.view(B * N, -1, D, H, W)
            cost = cost + diff.abs().sum(dim=1)
        if not self.bias == 0:
            invalid = wrap_prev[:, 0, ...].view(B * N, D, H, W) == 0
            cost = torch.where(invalid, cost + self.bias, cost)
        return (-cost).softmax(dim=1)

    def forward(self, x, mlp_input, stereo_metas=None):
        mlp_input = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        x = self.reduce_conv(x)
        context = self.context_se(x, self.context_mlp(mlp_input)[..., None, None])
        context = self.context_conv(context)
        depth = self.depth_se(x, self.depth_mlp(mlp_input)[..., None, None])
        if stereo_metas is not None:
            if stereo_metas['cv_feat_list'][0] is None:
                BN, _, H, W = x.shape
                sf = float(stereo_metas['downsample']) / stereo_metas['cv_downsample']
                cost_volumn = torch.zeros((BN, self.depth_channels, int(H * sf),
                                           int(W * sf))).to(x)
            else:
                with torch.no_grad():
                    cost_volumn = self.calculate_cost_volumn(stereo_metas)
            depth = torch.cat([depth, self.cost_volumn_net(cost_volumn)], dim=1)
        if self.with_cp:
            depth = checkpoint(self.depth_conv, depth)
        else:
            depth = self.depth_conv(depth)
        return torch.cat([depth, context], dim=1)


@register_neck()
class LSSViewTransformerBEVDepth(LSSViewTransformer):
    """view_transformer.py:694-791: DepthNet + 27-d camera-aware MLP input."""

    def __init__(self, loss_depth_weight=3.0, depthnet_cfg=dict(), **kwargs):
        super().__init__(**kwargs)
        self.loss_depth_weight = loss_depth_weight
        self.depth_net = DepthNet(self.in_channels, self.in_channels,
                                  self.out_channels, self.D, **depthnet_cfg)

    def get_mlp_input(self, sensor2ego, ego2global, intrin, post_rot,
                      post_tran, bda):
        """27 numbers per camera: 15 intrinsic / augmentation entries + the 12
        of sensor2ego[:3, :] (:703-724)."""
        B, N, _, _ = sensor2ego.shape
        bda = bda.view(B, 1, 3, 3).repeat(1, N, 1, 1)
        parts = [
            intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2],
            intrin[:, :, 1, 2], post_rot[:, :, 0, 0], post_rot[:, :, 0, 1],
            post_tran[:, :, 0], post_rot[:, :, 1, 0], post_rot[:, :, 1, 1],
            post_tran[:, :, 1], bda[:, :, 0, 0], bda[:, :, 0, 1],
            bda[:, :, 1, 0], bda[:, :, 1, 1], bda[:, :, 2, 2]]
        mlp_input = torch.stack(parts, dim=-1)
        return torch.cat([mlp_input, sensor2ego[:, :, :3, :].reshape(B, N, -1)],
                         dim=-1)

    def get_downsampled_gt_depth(self, gt_depths):
        """(B,N,H,W) metric depth -> (B*N*h*w, D) one-hot of the nearest
        non-zero depth per downsample block (:726-760)."""
        B, N, H, W = gt_depths.shape
        ds = self.downsample
        g = gt_depths.view(B * N, H // ds, ds, W // ds, ds)
        g = g.permute(0, 1, 3, 2, 4).reshape(-1, ds * ds)
        g = torch.where(g == 0.0, torch.full_like(g, 1e5), g).min(dim=-1).values
        lo, hi, step = self.grid_config['depth']
        if not self.sid:
            g = (g - (lo - step)) / step
        else:
            g = torch.log(g) - torch.log(torch.tensor(lo).float())
            g = g * (self.D - 1) / torch.log(torch.tensor(hi - 1.).float() / lo)
            g = g + 1.
        g = torch.where((g < self.D + 1) & (g >= 0.0), g, torch.zeros_like(g))
        return F.one_hot(g.long(), num_classes=self.D + 1)[:, 1:].float()

    def get_depth_loss(self, depth_labels, depth_preds):
        """BCE between predicted depth distribution and one-hot GT (:762-778)."""
        depth_labels = self.get_downsampled_gt_depth(depth_labels.float())
        depth_preds = depth_preds.float().permute(0, 2, 3, 1).contiguous() \
            .view(-1, self.D)
        fg = depth_labels.max(dim=1).values > 0.0
        loss = F.binary_cross_entropy(depth_preds[fg], depth_labels[fg],
                                      reduction='none').sum() / max(1.0, fg.sum())
        return self.loss_depth_weight * loss

    def forward(self, input, stereo_metas=None):
        """input = (x, sensor2ego, ego2global, intrins, post_rots, post_trans,
        bda, mlp_input) -> (bev_feat, depth)  (:780-791)."""
        x, mlp_input = input[0], input[7]
        B, N, C, H, W = x.shape
        x = self.depth_net(x.view(B * N, C, H, W), mlp_input, stereo_metas)
        depth = x[:, :self.D, ...].softmax(dim=1)
        tran_feat = x[:, self.D:self.D + self.out_channels, ...]
        return self.view_transform(input, depth, tran_feat)


@register_neck()
class LSSViewTransformerBEVStereo(LSSViewTransformerBEVDepth):
    """view_transformer.py:794-801: adds the 4x-downsample cost-volume frustum."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.cv_frustum = self.create_frustum(kwargs['grid_config']['depth'],
                                              kwargs['input_size'],
                                              downsample=4)

"""VEON's lift -- mirror of mmdet3d/models/necks/view_transformer_raw.py.

``LSSViewTransformerRaw`` takes an externally estimated depth distribution
(from DepthAnythingV2 via ``get_two_hot_depth``) instead of predicting one,
pools CLIP features into the 200x200x16 voxel grid and max-pools 2x2x2
(``ds_feat``).  Constructor kwargs, attributes and method contracts follow the
reference class (:17-72, :393-429, :537-555); SAN calls it as
``self.lss_view_transformer([feats_2d] + img_metas, depth)`` and reads
``.mode``, ``.downsample_depth``, ``.get_two_hot_depth``
(align_net_occ3d.py:245,313,321,325).
"""
import torch
import torch.nn.functional as F

from ..builder import register_neck
from .lss_core import LSSCore
from ... import depth_ops


@register_neck()
class LSSViewTransformerRaw(LSSCore):
    r"""Args as the reference (:22-40) plus ``mode`` ('nuscenes'),
    ``loss_depth_weight`` and ``ds_feat`` = [dz, dh, dw] max-pool factors."""

    _core_returns_depth = False

    def __init__(self, grid_config, input_size, downsample=16,
                 out_channels=256, accelerate=False, sid=False, collapse_z=True,
                 mode='nuscenes', loss_depth_weight=0.05, ds_feat=[2, 2, 2]):
        super().__init__()
        self._init_lss(grid_config, input_size, downsample, out_channels,
                       accelerate, sid, collapse_z)
        self.loss_depth_weight = loss_depth_weight
        self.mode = mode
        self.ds = ds_feat
        assert len(self.ds) == 3
        self.use_ds = any(x != 1 for x in self.ds)
        assert self.mode in ['nuscenes']
        # veon_amd extension: fold the ds_feat max-pool into the pool kernel at
        # inference (bit-equal result; the full-resolution volume is never
        # written).  Set False to run the reference's two-step structure.
        self.fuse_ds = True

    # ------------------------------------------------------------ depth prep
    def downsample_depth(self, depths, downsample):
        """(B,N,H,W) -> (B,N,H/ds,W/ds): min over the non-zero pixels of each
        block, zeros counted as 1e5 (:393-404)."""
        return depth_ops.downsample_depth(depths, downsample)

    def get_two_hot_depth(self, depths, gamma=4, downsample=False):
        """Metric depth (B,N,H,W) -> soft two-hot distribution (B,N,D,H,W):
        softmax over D+1 bin centres of -gamma*|d - c_k| clamped at -16, last
        bin dropped (:406-429)."""
        lo, _, step = self.grid_config['depth']
        if downsample:
            return depth_ops.two_hot_depth_fused(depths, self.downsample, self.D,
                                                 lo, step, gamma)
        return depth_ops.two_hot_depth(depths, self.D, lo, step, gamma)

    def get_two_hot_windows(self, depths, gamma=4, downsample=0, eps=0.0):
        """veon_amd extension: ``get_two_hot_depth`` in compact,
        exact form -- a ``depth_ops.TwoHotWindows`` that ``forward`` takes in place of
        the (B,N,D,H,W) tensor, which is then never written.  ``downsample`` fuses
        ``downsample_depth``; ``eps`` > 0 additionally drops the points whose weight is
        below it (every pooled sum then moves by at most eps * sum|feat| of the
        dropped points of its voxel); eps = 0 is the dense lift to the bit."""
        lo, _, step = self.grid_config['depth']
        return depth_ops.two_hot_windows(depths, self.D, lo, step, gamma, eps,
                                         int(downsample))

    def get_one_hot_depth(self, depths, downsample=False):
        """Hard nearest-bin assignment (:431-456)."""
        if downsample:
            depths = self.downsample_depth(depths, self.downsample)
        B, N, H, W = depths.shape
        lo, _, step = self.grid_config['depth']
        centers = torch.arange(self.D + 1, device=depths.device) * step + \
            (lo + step / 2)
        d = depths.clamp_max(500).reshape(B * N, H, W, 1)
        idx = (-(d - centers.view(1, 1, 1, -1)).abs()).max(-1, keepdim=True)[1]
        hot = torch.zeros(B * N, H, W, self.D + 1, device=depths.device,
                          dtype=depths.dtype).scatter_(-1, idx, 1.0)
        return hot[..., :-1].view(B, N, H, W, self.D).permute(0, 1, 4, 2, 3)

    def get_downsampled_gt_depth(self, gt_depths):
        """Block-min one-hot GT for the depth loss (:339-376)."""
        B, N, H, W = gt_depths.shape
        g = self.downsample_depth(gt_depths, self.downsample).view(-1)
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
        """BCE depth loss (:479-497), training only."""
        depth_labels = self.get_downsampled_gt_depth(depth_labels.float())
        if depth_preds.dim() == 5:
            depth_preds = depth_preds.reshape(-1, *depth_preds.shape[2:])
        depth_preds = depth_preds.float().permute(0, 2, 3, 1).contiguous() \
            .view(-1, self.D)
        fg = depth_labels.max(dim=1).values > 0.0
        loss = F.binary_cross_entropy(depth_preds[fg], depth_labels[fg],
                                      reduction='none').sum() / max(1.0, fg.sum())
        return self.loss_depth_weight * loss

    def _can_fuse_ds(self, feat):
        if not (self.use_ds and self.fuse_ds and feat.is_cuda
                and not self.collapse_z and not torch.is_grad_enabled()):
            return False
        x, y, z = (int(v) for v in self.grid_size)
        return z % self.ds[0] == 0 and y % self.ds[1] == 0 and x % self.ds[2] == 0

    # ---------------------------------------------------------------- forward
    def forward(self, input, depth, stereo_metas=None, out_volume=None):
        """input = (tran_feat (B,N,C,Hf,Wf), sensor2ego, ego2global, intrins,
        post_rots, post_trans, bda); depth (B,N,D,Hf,Wf).  Returns the pooled
        volume (B,C,Z,Y,X), max-pooled by ``ds_feat`` when any factor != 1
        (:537-555).  veon_amd extension: ``out_volume`` (a
        ``conv3d_ops.PaddedVolume``) receives the max-pooled volume in the
        Conv3d body's padded bf16 layout and is returned instead (fused
        inference path only)."""
        tran_feat = input[0]
        B, N, C, H, W = tran_feat.shape
        windows = isinstance(depth, depth_ops.TwoHotWindows)
        if windows and tuple(depth.shape) != (B, N, self.D, H, W):
            raise ValueError('two-hot windows of shape %r for a (%d,%d,%d,%d,%d) lift'
                             % (tuple(depth.shape), B, N, self.D, H, W))
        if out_volume is not None and not self._can_fuse_ds(tran_feat):
            raise ValueError('out_volume needs the fused inference max-pool path '
                             '(ROCm tensors, no grad, ds_feat dividing the grid)')
        if self._can_fuse_ds(tran_feat):
            out = self._lift_maxpool(input,
                                     depth if windows else depth.view(B, N, self.D, H, W),
                                     tran_feat, self.ds, out_volume=out_volume)
            if out is not None:
                return out
            if out_volume is not None:   # empty grid: the volume is all zeros
                out_volume.rows.zero_()
                return out_volume
        tran_feat = tran_feat.view(B * N, C, H, W)
        if windows and not (self.sync_free and not self.accelerate and tran_feat.is_cuda
                            and not torch.is_grad_enabled()):
            # no compact path here (CPU, training, cached ranks): the dense tensor,
            # zero where the threshold drops a point
            depth, windows = depth.dense(thresholded=depth.eps > 0), False
        if not windows:
            depth = depth.view(B * N, depth.shape[2], H, W)
        bev_feat = self.view_transform(input, depth, tran_feat)
        if self.use_ds:
            dz, dh, dw = self.ds
            b, c, z, h, w = bev_feat.shape
            bev_feat = bev_feat.view(b, c, z // dz, dz, h // dh, dh, w // dw, dw) \
                .amax(dim=(3, 5, 7))
        return bev_feat

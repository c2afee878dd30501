"""``AlignNetOcc3D`` -- the 3-D occupancy decoder that owns the lift: mirror of
mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:207-361, with the
temporal fusion of SURVEY 8 row f4 (``num_temporal > 1``: ``temporal_fusion`` is a
``TemporalFusionMultiFrame``, applied after the lift of block ``tf_layers`` = 0 to
the current volume and the aligned volumes of the past frames).

Same constructor, sub-module / parameter names (``fusion_layers.layer_N``,
``layers_3d_body.N``, ``occupancy_pred``, ``feat_pred``) and method contracts
(``forward``, ``forward_early``, ``fuse``, ``prepare_depth``, ``prepare_meta``,
``prepare_feat_for_lifting``); the view transformer is injected exactly as the
reference does it (``prepare_lss`` sets ``lss_view_transformer`` /
``num_frame`` / ``num_camera``, san_in_veon_temporal.py:275-279).

At inference on a ROCm device with the usual single lifting layer
(``LIFTING_LAYERS = ['12->0->0']``) the whole decoder is one chain of native
kernels: fused depth prep -> fusion layer (PyTorch) -> lift + 2x2x2 max-pool
written straight into the body's padded bf16 volume -> ResBlock3D body ->
prediction heads as GEMMs on the same rows.  Otherwise the PyTorch definition
of every module runs (training, CPU tensors for everything but the lift op).
"""
from typing import List

import torch
import torch.nn as nn

from ... import conv3d_ops
from .align_net_body import AlignBody3D, PredHead3DOcc, PredHead3DSem, ResBlock3D
from .fusion_layers import build_fusion_layer_lift
from .temporal_fusion import TemporalFusionMultiFrame


class AlignNetOcc3D(nn.Module):
    def __init__(self, clip_dim=1024, hsa_dim=240, embed_dim=384, clip_outdim=768,
                 layer_lifting_map=None, fusion_type='add', layer_depth=5,
                 num_temporal=1):
        super().__init__()
        self.fusion_map = {int(k): (int(i), int(j)) for i, j, k in
                           [x.split('->') for x in layer_lifting_map]}
        self.fusion_layers = nn.ModuleDict({
            'layer_%d' % tgt: build_fusion_layer_lift(fusion_type, hsa_dim, clip_dim,
                                                      embed_dim)
            for tgt in self.fusion_map})
        self.layers_3d_body = nn.ModuleList(
            [ResBlock3D(embed_dim, embed_dim) for _ in range(layer_depth)])
        self.occupancy_pred = PredHead3DOcc(embed_dim, 2)
        self.feat_pred = PredHead3DSem(embed_dim, clip_outdim)
        self.tf_layers = 0
        self.temporal_fusion = TemporalFusionMultiFrame(
            channels=embed_dim, seqs=num_temporal - 1) if num_temporal > 1 else None
        self.use_hip = True
        # the body runner shares the ModuleList (not registered twice)
        body = AlignBody3D.__new__(AlignBody3D)
        nn.Module.__init__(body)
        body.layers_3d_body = self.layers_3d_body
        body.use_hip, body._hip, body._bufs = True, None, {}
        self.__dict__['_body'] = body
        self.__dict__['_lifted'] = {}

    def train(self, mode=True):
        self.__dict__['_body'].train(mode)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__['_body'].invalidate_hip_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .half()
        self.__dict__['_body'].invalidate_hip_cache()
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------ preparation
    # veon_amd extension: None = the reference's dense (B,N,D,Hf,Wf)
    # two-hot tensor; a float eps >= 0 = the two-hot lift by construction -- at
    # inference on a ROCm device ``prepare_depth`` returns ``depth_ops.TwoHotWindows``
    # (compact exact weights; points with weight < eps dropped before the sort; 0 =
    # the dense lift to the bit) and the D-wide tensor is never written.
    two_hot_eps = None

    def prepare_depth(self, depth):
        vt = self.lss_view_transformer
        if (self.two_hot_eps is not None and depth.is_cuda and not torch.is_grad_enabled()
                and getattr(vt, 'sync_free', False) and not vt.accelerate
                and hasattr(vt, 'get_two_hot_windows')):
            return vt.get_two_hot_windows(depth, downsample=8, eps=float(self.two_hot_eps))
        depth_ds = vt.downsample_depth(depth, downsample=8)
        return vt.get_two_hot_depth(depth_ds)

    def prepare_meta(self, img_metas):
        N = self.num_camera
        sensor2egos, ego2globals, intrins, post_rots, post_trans, bda = img_metas
        if (self.num_frame == 1 and sensor2egos.is_cuda and not torch.is_grad_enabled()):
            # one frame at inference: the 4x4 algebra in ONE launch
            from ... import lss_prepare_hip
            s2k = lss_prepare_hip.sensor2keyego(sensor2egos.reshape(-1, N, 4, 4),
                                                ego2globals.reshape(-1, N, 4, 4))
            return [s2k, ego2globals.view(-1, N, 4, 4), intrins.view(-1, N, 3, 3),
                    post_rots.view(-1, N, 3, 3), post_trans.view(-1, N, 3), bda[0]]
        sensor2egos = sensor2egos.view(-1, self.num_frame, N, 4, 4)
        ego2globals = ego2globals.view(-1, self.num_frame, N, 4, 4)
        keyego2global = ego2globals[:, 0, 0, ...].unsqueeze(1).unsqueeze(1)
        # same LU routine as torch.inverse, minus its host-side check of `info`
        # (a device->host sync in front of the whole decoder)
        global2keyego = torch.linalg.inv_ex(keyego2global.double(), check_errors=False)[0]
        sensor2keyegos = (global2keyego @ ego2globals.double()
                          @ sensor2egos.double()).float()
        extra = [sensor2keyegos, ego2globals,
                 intrins.view(-1, self.num_frame, N, 3, 3),
                 post_rots.view(-1, self.num_frame, N, 3, 3),
                 post_trans.view(-1, self.num_frame, N, 3)]
        extra = [[p.squeeze(1) for p in torch.split(t, 1, 1)] for t in extra]
        s2k, e2g, intr, pr, pt = extra
        return [s2k[0], e2g[0], intr[0], pr[0], pt[0], bda[0]]

    def prepare_feat_for_lifting(self, feats_2d):
        _, C, H, W = feats_2d.shape
        feats_2d = feats_2d.view(-1, self.num_camera, self.num_frame, C, H, W)
        return [t.squeeze(2) for t in torch.split(feats_2d, 1, dim=2)][0]

    # ------------------------------------------------------------------ fusion
    def fuse_2d(self, block_idx, clip_features, supp_features, lift_shape):
        """The 2-D half of ``fuse``: the fusion layer on the CLIP / HSA maps (needs no
        depth, so a caller can issue it at the end of the semantic branch)."""
        src_clip, src_ec = self.fusion_map[block_idx]
        return self.fusion_layers['layer_%d' % block_idx](
            supp_features[src_ec], clip_features[src_clip], lift_shape)

    def fuse(self, block_idx, x, clip_features, supp_features, depth, img_metas,
             clip_shape, lift_shape, out_volume=None, fused=None):
        if block_idx in self.fusion_map:
            if fused is None:
                fused = self.fuse_2d(block_idx, clip_features, supp_features, lift_shape)
            feats_2d = self.prepare_feat_for_lifting(fused)
            if out_volume is not None:
                return self.lss_view_transformer([feats_2d] + img_metas, depth,
                                                 out_volume=out_volume)
            lifted = self.lss_view_transformer([feats_2d] + img_metas, depth)
            x = lifted if x is None else x + lifted
        return x

    def _fast_path(self, sem_feat):
        vt = self.lss_view_transformer
        return (self.use_hip and sem_feat.is_cuda and not self.training
                and not torch.is_grad_enabled() and set(self.fusion_map) == {0}
                and getattr(vt, 'use_ds', False) and not vt.collapse_z
                and all(b.hip_supported() for b in self.layers_3d_body))

    def _lift_volume(self, B, C, device):
        vt = self.lss_view_transformer
        x, y, z = (int(v) for v in vt.grid_size)
        dz, dy, dx = vt.ds
        key = (B, C, z // dz, y // dy, x // dx, str(device))
        cache = self.__dict__['_lifted']
        if key not in cache:
            cache[key] = conv3d_ops.PaddedVolume(*key[:5], device)
        return cache[key]

    def forward(self, sem_feat, clip_features: List, supp_features: List, depth,
                img_metas: List, occ_feat_prevs: List = None):
        if occ_feat_prevs is not None and len(occ_feat_prevs) == 0:
            occ_feat_prevs = None
        if occ_feat_prevs is not None and self.temporal_fusion is None:
            raise ValueError('occ_feat_prevs given but the decoder was built with '
                             'num_temporal=1')
        depth = self.prepare_depth(depth)
        if self.lss_view_transformer.mode == 'nuscenes':
            img_metas = self.prepare_meta(img_metas)
        h, w = clip_features[1].shape[2:]
        H, W = sem_feat.shape[2:]
        if self._fast_path(sem_feat):
            embed = self.layers_3d_body[0].conv1.conv.in_channels
            vol = self._lift_volume(depth.shape[0], embed, sem_feat.device)
            if self.lss_view_transformer._can_fuse_ds(sem_feat):
                x = self.fuse(0, None, clip_features, supp_features, depth, img_metas,
                              (h, w), (H, W), out_volume=vol)
                if occ_feat_prevs is not None:
                    x = self._temporal(x, occ_feat_prevs)
                x = self.__dict__['_body'](x, return_volume=True)
                return {'bin_occ': self.occupancy_pred(x), 'feat_occ': self.feat_pred(x)}
        x = None
        for idx, layer_3d in enumerate(self.layers_3d_body):
            x = self.fuse(idx, x, clip_features, supp_features, depth, img_metas,
                          (h, w), (H, W))
            if idx == self.tf_layers and occ_feat_prevs is not None:
                x = self._temporal(x, occ_feat_prevs)
            x = layer_3d(x)
        return {'bin_occ': self.occupancy_pred(x), 'feat_occ': self.feat_pred(x)}

    def _temporal(self, x, prevs):
        """``temporal_fusion`` on whatever the two sides are: PaddedVolumes stay on
        the MFMA path (fp32 ROCm tensors are packed when it is available),
        otherwise the PyTorch definition runs on fp32 tensors."""
        tf = self.temporal_fusion
        padded = conv3d_ops.PaddedVolume
        if self.use_hip and tf.hip_ok(x) and all(
                isinstance(p, padded) or p.is_cuda for p in prevs):
            vols = [v if isinstance(v, padded) else conv3d_ops.pack(v)
                    for v in [x] + list(prevs)]
            out = tf(vols[0], vols[1:])
            return out if isinstance(x, padded) else conv3d_ops.unpack(out)
        vals = [conv3d_ops.unpack(v) if isinstance(v, padded) else v
                for v in [x] + list(prevs)]
        out = tf(vals[0], vals[1:])
        return conv3d_ops.pack(out) if isinstance(x, padded) else out

    def forward_early(self, sem_feat, clip_features, supp_features, depth, img_metas,
                      out_volume=None):
        """The lifted volume of a (past) frame, before any 3-D layer
        (align_net_occ3d.py:268-280).  ``out_volume`` (a PaddedVolume of the
        max-pooled shape): have the lift write there and return it."""
        depth = self.prepare_depth(depth)
        if self.lss_view_transformer.mode == 'nuscenes':
            img_metas = self.prepare_meta(img_metas)
        h, w = clip_features[1].shape[2:]
        H, W = sem_feat.shape[2:]
        if out_volume is not None and not (
                self._fast_path(sem_feat)
                and self.lss_view_transformer._can_fuse_ds(sem_feat)):
            raise ValueError('out_volume needs the fused lift + max-pool path')
        return self.fuse(0, None, clip_features, supp_features, depth, img_metas,
                         (h, w), (H, W), out_volume=out_volume)

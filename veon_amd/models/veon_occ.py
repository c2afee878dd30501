"""The 3-D occupancy path of ``VeonTemporal.simple_test`` assembled from this
package's mirrors (single frame, inference):

    images (B, N, 3, H, W) + calibration
      depth    : DepthAnythingV2Adaptor at 252x700 -> metric depth at (H/2, W/2)
                 (veon_temporal.py:209-214)
      semantic : CLIP trunk on the half-resolution image, first K blocks
                 (FeatureExtractor) -> HSA network on the full image (attention
                 biases + supp features) -> CLIP tail blocks with the biases
                 (RecWithAttnbiasHead.update_remaining_clip_feats)
                 (san_in_veon_temporal.py:118-123, 189-191)
      3-D      : AlignNetOcc3D (fusion layer -> lift -> Conv3d body -> heads),
                 open-vocabulary classifier, trilinear upsampling, arg-max
                 (san_in_veon_temporal.py:193-211, veon_temporal.py:216-227)

NOT included (third-party / not rebuilt): the timm side-adapter ViT, its mask
decoder and the 2-D segmentation outputs (``sem_embed_ds`` only supplies a shape
to the decoder), the text encoder (class embeddings are an input).  Weights are
whatever the sub-modules hold (random unless loaded).

Temporal (``num_temporal > 1``, san_in_veon_temporal.py:158-173): ``lift_frame``
gives a past frame's lifted volume (``forward_early``), ``align`` warps it into
the current ego frame (``align_after_lss``), and ``forward(...,
prev_volumes=[...])`` fuses them after the current lift.  The reference recomputes
the past frames' encoders at every step; a streaming deployment keeps each frame's
lifted volume and only re-warps it -- same numbers, both are possible with this
interface.

The two encoder branches are independent and neither fills the chip, so they run
on two HIP streams (``two_streams=True``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .builder import build_neck
from .semantic_net import (AlignNetOcc3D, ClipRecHead, ClipVisualTrunk,
                           classifier_logits_low, semantic_inference_3d_fused)
from .semantic_net.hsa_network import HighresSideAdaptorNetwork
from .. import half as _half


class VeonOccupancyPath(nn.Module):
    # VEON-L (configs[3]/[4]): SAN on CLIP ViT-L/14-336 + DepthAnythingV2 ViT-L
    # (semantic_net/configs/san_clip_vit_large_res4_coco_temporal.yaml:5-6,15:
    # K = 18 of 24 blocks, 16 heads, HSA fusion map 0->3->6 / 1->9->12 / 2->15->18,
    # lifting layer 24, CLIP projection 768; clip_utils/visual.py:31-52)
    VEON_L = dict(encoder='vitl', clip_width=1024, clip_layers=24, clip_heads=16,
                  clip_first_tail=18, clip_proj_dim=768, clip_patch=14, clip_image=336,
                  hsa_fusion_map=('0->3->6', '1->9->12', '2->15->18'))

    def __init__(self, input_size=(256, 704), grid_config=None, num_cam=6,
                 encoder='vitb', n_classes=17, clip_width=768, clip_layers=12,
                 clip_heads=12, clip_first_tail=9, clip_proj_dim=512, embed_dim=256,
                 occ_size=(16, 200, 200), bf16_heads=True, two_streams=True,
                 hsa_dim=384, hsa_fusion_map=('0->3->3', '1->6->6', '2->9->9'),
                 num_temporal=1, clip_patch=16, clip_image=224, side_adapter=None,
                 sparse_lift_eps=1e-6):
        super().__init__()
        from .. import synthetic
        grid_config = grid_config or synthetic.GRID_VEON
        dav2_cfg = {'vitb': dict(encoder='vitb', features=128,
                                 out_channels=[96, 192, 384, 768]),
                    'vitl': dict(encoder='vitl', features=256,
                                 out_channels=[256, 512, 1024, 1024])}[encoder]
        self.depth_model = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0,
                                           use_lora=True, lora_r=16, **dav2_cfg))
        self.clip_trunk = ClipVisualTrunk(clip_image, clip_patch, clip_width, clip_layers,
                                          clip_heads)
        self.clip_first_tail = clip_first_tail
        self.ln_post = nn.LayerNorm(clip_width)
        self.clip_proj = nn.Parameter(torch.randn(clip_width, clip_proj_dim)
                                      * clip_width ** -0.5)
        self.clip_rec_head = ClipRecHead(self.clip_trunk.resblocks, self.ln_post,
                                         self.clip_proj, first_layer_idx=clip_first_tail)
        # the 2-D mask-proposal branch (side_adapter/side_adaptor_in_veon.py) is
        # optional: the occupancy path takes only a shape from it.  ``side_adapter``:
        # True for the SAN defaults or a dict of RegionwiseSideAdapterNetwork.build
        # arguments.
        self.side_adapter_network = None
        if side_adapter:
            from .semantic_net.side_adapter import RegionwiseSideAdapterNetwork
            kw = dict(side_adapter) if isinstance(side_adapter, dict) else {}
            kw.setdefault('attn_heads', clip_heads)
            self.side_adapter_network = RegionwiseSideAdapterNetwork.build(
                clip_dim=clip_width, **kw)
            self.clip_rec_head.sos_token_num = self.side_adapter_network.num_queries
        self.hsa = HighresSideAdaptorNetwork.build(
            dim=hsa_dim, clip_dim=clip_width, mlp_dim=hsa_dim, input_size=input_size,
            fusion_map=hsa_fusion_map, manip_supp_dim=hsa_dim, num_heads=clip_heads,
            manip_attn_layers=max(clip_layers - clip_first_tail, 1))
        # its attention biases feed ClipRecHead.update_remaining_clip_feats only: let them
        # come out already bordered for the class token (inference)
        self.hsa.rear_block.pad_class_token = True
        self.view_transformer = build_neck(dict(
            type='LSSViewTransformerRaw', grid_config=grid_config, input_size=input_size,
            out_channels=embed_dim, collapse_z=False, ds_feat=[2, 2, 2]))
        self.view_transformer.sync_free = True
        # the two-hot lift by construction (default): the depth map
        # goes into the lift as per-pixel windows of the two-hot distribution
        # (depth_ops.TwoHotWindows), the (B,6,D,Hf,Wf) tensor is never written, and
        # frustum points whose weight is below ``sparse_lift_eps`` are dropped before
        # the sort (pooled sums move by <= eps * sum|feat| of the dropped points; 0.0 =
        # the dense lift to the bit).  None = the reference's dense two-hot tensor.
        self.view_transformer.sparse_depth_eps = None
        self.occ_decoder = AlignNetOcc3D(
            clip_dim=clip_width, hsa_dim=hsa_dim, embed_dim=embed_dim,
            clip_outdim=clip_proj_dim, layer_lifting_map=['%d->0->0' % clip_layers],
            fusion_type='cat_fusion', layer_depth=4, num_temporal=num_temporal)
        self.occ_decoder.lss_view_transformer = self.view_transformer
        self.occ_decoder.two_hot_eps = sparse_lift_eps
        self.occ_decoder.num_frame, self.occ_decoder.num_camera = 1, num_cam
        self.ov_classifier_weight = nn.Parameter(torch.randn(n_classes, clip_proj_dim))
        self.input_size, self.num_cam, self.occ_size = input_size, num_cam, occ_size
        self.two_streams = two_streams
        if bf16_heads:
            self.depth_model.head_dtype = _half.dtype()
            self.hsa.set_conv_dtype(_half.dtype())
            for layer in self.occ_decoder.fusion_layers.values():
                layer.hip_dtype = _half.dtype()
        self.__dict__['_side'] = None

    # ------------------------------------------------------------- branches
    def estimate_depth(self, img, num_cam=None):
        """(B*N,3,H,W) -> metric depth (B, N, H/2, W/2) (veon_temporal.py:209-214;
        the reference resizes to 252x700 for any of its input sizes)."""
        num_cam = num_cam or self.num_cam
        H, W = img.shape[-2:]
        x = F.interpolate(img, (252, 700), mode='bilinear', align_corners=False)
        d = self.depth_model(x)['metric_depth']
        d = F.interpolate(d[:, None], (H // 2, W // 2), mode='bilinear',
                          align_corners=True)[:, 0]
        return d.view(-1, num_cam, H // 2, W // 2)

    def clip_features(self, img):
        """FeatureExtractor on the half-resolution image, HSA on the full one,
        then the CLIP tail with the HSA attention biases -> the reference's
        ``ClipOutput`` dict (+ 'supp')."""
        x = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False)
        outs, hw = self.clip_trunk(x, last_layer_idx=self.clip_first_tail,
                                   taps=self._clip_taps())
        feats = {}
        for i, t in enumerate(outs):
            if t is not None:
                ClipRecHead._save(feats, i, t, hw)
        _, attns, supp = self.hsa(img, feats)
        last = self.clip_first_tail + len(self.clip_rec_head.resblocks)
        feats = self.clip_rec_head.update_remaining_clip_feats(
            feats, None, attns, keep_layers={last} | self._lift_layers())
        return feats, supp

    def _lift_layers(self):
        return {src for src, _ in self.occ_decoder.fusion_map.values()}

    def _clip_taps(self):
        """The CLIP layers anything downstream reads: the HSA network's cross-attention
        / add sources, the side adapter's fusion sources, the decoder's lifting sources,
        the recognition head's first layer (and layer 1, whose SHAPE the HSA network
        and the decoder read).  The trunk copies only those out of its stream."""
        taps = {0, 1, self.clip_first_tail} | self._lift_layers()
        for a, b in self.hsa.cr_map.values():
            taps |= {a, b}
        if self.side_adapter_network is not None:
            taps |= set(self.side_adapter_network.fusion_map.values())
        return {t for t in taps if 0 <= t <= self.clip_first_tail}

    def forward_2d(self, images):
        """The 2-D open-vocabulary segmentation branch (san_in_veon_temporal.py:
        123-139, 176-186): images (B, N, 3, H, W) -> dict with ``mask_preds``,
        ``mask_embs``, ``mask_logits``, ``sem_seg_ds``, ``sem_embed_ds``, ``sem_seg``.
        Needs ``side_adapter=True`` at construction."""
        if self.side_adapter_network is None:
            raise RuntimeError('VeonOccupancyPath was built without side_adapter=True')
        from .semantic_net.side_adapter import semantic_branch_2d
        img = images.flatten(0, 1)
        x = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False)
        outs, hw = self.clip_trunk(x, last_layer_idx=self.clip_first_tail)
        feats = {}
        for i, t in enumerate(outs):
            ClipRecHead._save(feats, i, t, hw)
        return semantic_branch_2d(self.side_adapter_network, self.clip_rec_head,
                                  self.ov_classifier_weight, img, feats)

    def _branches(self, images, depth=None, metas=None):
        """Both encoder branches of one frame -> (CLIP feature dict, supp, depth).
        ``depth`` (B, N, H/2, W/2): cached metric depth (veon_amd/depth_cache.py)
        used instead of the depth encoder, as the reference's ``use_depth_pred``
        data pipeline does.  ``metas``: the decoder's camera tensors; their 4x4 algebra
        (``prepare_meta``, ~14 tiny launches that depend on nothing else) is then issued
        in front of the shorter semantic branch instead of on the critical path after
        both branches; so is the decoder's 2-D fusion layer, which needs no depth.  Both are
        then returned as a fourth and fifth value."""
        img = images.flatten(0, 1)
        n_cam = images.shape[1]
        dec = self.occ_decoder
        hf, wf = self.input_size[0] // 16, self.input_size[1] // 16

        def prep():   # camera algebra, then the lift's (depth-independent) prepare
            if metas is None:
                return None
            m2 = dec.prepare_meta(metas)
            if dec.two_hot_eps is None:   # (the two-hot lift's prepare needs the depth)
                self.view_transformer.prepare_lift(m2)
            return m2

        def sem():   # CLIP -> HSA -> CLIP tail, then (fast path) the 2-D fusion layer
            feats, supp = self.clip_features(img)
            fused = dec.fuse_2d(0, feats, [supp], (hf, wf)) if metas is not None else None
            return feats, supp, fused
        if depth is not None:
            metas2 = prep()
            feats, supp, fused = sem()
            out = (feats, supp, depth.to(img.device, torch.float32))
        elif self.two_streams and img.is_cuda:
            if self.__dict__['_side'] is None:
                self.__dict__['_side'] = torch.cuda.Stream()
            side, cur = self.__dict__['_side'], torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                metas2 = prep()
                feats, supp, fused = sem()
            depth = self.estimate_depth(img, n_cam)
            cur.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                # (a captured forward keeps every tensor of the capture alive)
                for t in (list(feats.values()) + [supp] + list(metas2 or [])
                          + ([fused] if fused is not None else [])):
                    t.record_stream(cur)
            out = (feats, supp, depth)
        else:
            metas2 = prep()
            feats, supp, fused = sem()
            out = (feats, supp, self.estimate_depth(img, n_cam))
        return out if metas is None else out + (metas2, fused)

    def lift_frame(self, images, img_metas, out_volume=None, depth=None):
        """Lifted, max-pooled volume of one frame before any 3-D layer
        (``occ_decoder.forward_early``, san_in_veon_temporal.py:166-168): what a
        later step needs of this frame.  On the native path the result is a
        PaddedVolume (a new one unless ``out_volume`` is given)."""
        from .. import conv3d_ops
        B, N = images.shape[:2]
        feats, supp, depth = self._branches(images, depth)
        hf, wf = self.input_size[0] // 16, self.input_size[1] // 16
        sem_embed_ds = images.new_zeros((B * N, 1, hf, wf))
        metas = list(img_metas[:5]) + [img_metas[5][None]]
        dec = self.occ_decoder
        if dec._fast_path(sem_embed_ds) and self.view_transformer._can_fuse_ds(sem_embed_ds):
            if out_volume is None:
                like = dec._lift_volume(B, dec.layers_3d_body[0].conv1.conv.in_channels,
                                        images.device)
                out_volume = like.like()
            return dec.forward_early(sem_embed_ds, feats, [supp], depth, metas,
                                     out_volume=out_volume)
        return dec.forward_early(sem_embed_ds, feats, [supp], depth, metas)

    def align(self, volume, adj_metas):
        """``align_after_lss``: a past frame's volume resampled in the current ego
        frame; ``adj_metas`` = [lidarego2global (B,1,4,4), lidaregoprev2global]."""
        from .semantic_net.temporal_fusion import align_after_lss
        vt = self.view_transformer
        return align_after_lss(volume, adj_metas, vt.grid_config, tuple(vt.ds))

    def _tail(self, x, prev_volumes=None):
        """Conv3d body -> heads -> classifier -> upsampling -> arg-max on a lifted
        PaddedVolume (san_in_veon_temporal.py:196-211, veon_temporal.py:219-227)."""
        dec = self.occ_decoder
        if prev_volumes:
            x = dec._temporal(x, prev_volumes)
        x = dec.__dict__['_body'](x, return_volume=True)
        bin_occ = dec.occupancy_pred(x)
        feat = dec.feat_pred(x, return_volume=True)
        return self._classify(bin_occ, feat)

    def _classify(self, bin_occ, feat):
        from .. import conv3d_ops
        W = self.ov_classifier_weight
        if bin_occ.is_cuda and not torch.is_grad_enabled() and bin_occ.shape[1] == 2:
            # upsampling x2, both softmaxes, arg-max and the label volume: one kernel
            low = classifier_logits_low(W, feat)
            sem_occ, bin_up, occ = conv3d_ops.occ_classify(low.float(), bin_occ.float(),
                                                           self.occ_size)
            return {'bin_occ': bin_up, 'sem_occ': sem_occ, 'occ_pred_cls': occ}
        sem_occ = semantic_inference_3d_fused(self.ov_classifier_weight, feat, self.occ_size)
        bin_occ = F.interpolate(bin_occ, size=tuple(self.occ_size), mode='trilinear',
                                align_corners=False)
        # veon_temporal.py:219-227
        score, cls = torch.softmax(sem_occ, dim=1).max(dim=1)
        keep = (score > 0.0) & (torch.softmax(bin_occ, dim=1)[:, 0] > 0.5)
        occ = torch.where(keep, cls, torch.full_like(cls, sem_occ.shape[1]))
        return {'bin_occ': bin_occ, 'sem_occ': sem_occ,
                'occ_pred_cls': occ.permute(0, 3, 2, 1).contiguous()}

    def lift_cameras(self, images, img_metas, lo, hi, depth=None):
        """The UN-POOLED lifted volume (B, C, Z, Y, X) fp32 of cameras [lo, hi) of
        the rig: both encoder branches, the HSA network and the fusion layer run on
        those cameras only, then the lift (V = sum over cameras of V_cam, so the
        volumes of disjoint camera sets add up to the full one; the 2x2x2 max-pool
        does not commute with that sum and comes after it).  ``img_metas`` is the
        calibration of the WHOLE rig: the key-ego frame is camera 0's
        (align_net_occ3d.py:328-352)."""
        B, N = images.shape[:2]
        dec, vt = self.occ_decoder, self.view_transformer
        metas = dec.prepare_meta(list(img_metas[:5]) + [img_metas[5][None]])
        C = dec.layers_3d_body[0].conv1.conv.in_channels
        x, y, z = (int(v) for v in vt.grid_size)
        if hi <= lo:
            return torch.zeros((B, C, z, y, x), dtype=torch.float32, device=images.device)
        n = hi - lo
        sub = None if depth is None else depth[:, lo:hi]
        feats, supp, d = self._branches(images[:, lo:hi], sub)
        depth2 = dec.prepare_depth(d)                      # (B, n, D, Hf, Wf)
        hf, wf = self.input_size[0] // 16, self.input_size[1] // 16
        src_clip, src_ec = dec.fusion_map[0]
        fused = dec.fusion_layers['layer_0']([supp][src_ec], feats[src_clip], (hf, wf))
        feats_2d = fused.reshape(B, n, fused.shape[1], hf, wf)
        local = [t[:, lo:hi].contiguous() for t in metas[:5]] + [metas[5]]
        from .. import depth_ops
        vol = vt.view_transform([feats_2d] + local,
                                depth2 if isinstance(depth2, depth_ops.TwoHotWindows)
                                else depth2.reshape(B * n, -1, hf, wf),
                                feats_2d.reshape(B * n, -1, hf, wf).float())
        return vol if vol.dim() == 5 else vol.view(B, C, z, y, x)

    def forward_camera_sharded(self, images, img_metas, group=None, reduce_dtype=None,
                               depth=None, reduce='allreduce'):
        """BASELINE configs[3]: the cameras of ONE sample sharded over the ranks of
        ``group`` -- every rank runs the encoders / HSA / fusion / lift of its cameras
        (``lift_cameras``), the full-resolution voxel feature volumes are summed over
        the ranks (RCCL over xGMI; ``reduce_dtype=torch.bfloat16`` halves the
        message), then max-pool, Conv3d body, heads and classifier.  Ranks beyond the
        camera count contribute zeros.  Same outputs as ``forward`` up to the
        summation order of the reduction.

        ``reduce='allreduce'``: ONE all-reduce(SUM) of the un-pooled volume
        (2 (n-1)/n S bytes per rank on a ring), everything after it replicated.
        ``reduce='scatter'``: the 2x2x2 max-pool is per channel, so the volume is
        reduce-scattered over CHANNEL slices ((n-1)/n S), every rank max-pools its
        C/n channels, and the 8x smaller pooled slices are all-gathered
        ((n-1)/n S/8): 1.78x fewer bytes on the links than the all-reduce, and the
        pool itself is sharded.  Needs C % n == 0 (else falls back to the all-reduce).
        With B = 1 the channel slices of (B, C, Z, Y, X) are contiguous: no copy."""
        import torch.distributed as dist
        from .. import sharding
        if reduce not in ('allreduce', 'scatter'):
            raise ValueError("reduce must be 'allreduce' or 'scatter', got %r" % (reduce,))
        active = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if active else 1
        rank = dist.get_rank(group) if active else 0
        B, N = images.shape[:2]
        lo, hi = sharding.camera_slices(N, world)[rank]
        vol = self.lift_cameras(images, img_metas, lo, hi, depth)
        C = vol.shape[1]
        if world > 1 and reduce == 'scatter' and C % world == 0:
            return self.from_pooled(self._scatter_pool_gather(vol, world, group, reduce_dtype))
        if world > 1:
            if reduce_dtype is not None and reduce_dtype != vol.dtype:
                buf = vol.to(reduce_dtype)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                vol = buf.float()
            else:
                dist.all_reduce(vol, op=dist.ReduceOp.SUM, group=group)
        return self.from_volume(vol)

    def _scatter_pool_gather(self, vol, world, group, reduce_dtype):
        """reduce-scatter over channel slices -> max-pool of the own slice ->
        all-gather of the pooled slices; returns the pooled volume (B, C, Z/dz, Y/dy,
        X/dx) fp32, identical on every rank."""
        import torch.distributed as dist
        B, C, Z, Y, X = vol.shape
        cs = C // world
        buf = vol if reduce_dtype is None or reduce_dtype == vol.dtype else vol.to(reduce_dtype)
        if B > 1:   # rank-major: (world, B, C/world, Z, Y, X)
            buf = buf.view(B, world, cs, Z, Y, X).transpose(0, 1)
        buf = buf.contiguous()
        mine = torch.empty((B, cs, Z, Y, X), dtype=buf.dtype, device=buf.device)
        # flat views: gloo wants input.shape[0] == world * output.shape[0]
        dist.reduce_scatter_tensor(mine.view(-1), buf.view(-1), op=dist.ReduceOp.SUM,
                                   group=group)
        part = self._max_pool(mine.float())
        if reduce_dtype is not None:   # rounding is monotonic: max commutes with it,
            part = part.to(reduce_dtype)   # and the body packs to half precision anyway
        part = part.contiguous()
        allp = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(allp.view(-1), part.view(-1), group=group)
        # (world, B, C/world, ...) -> (B, C, ...)
        return allp.transpose(0, 1).reshape(B, C, *part.shape[2:]).float()

    def _max_pool(self, vol):
        """ds_feat block max of an un-pooled (B, C, Z, Y, X) volume
        (view_transformer_raw.py:549-553); one native streaming pass on a ROCm device
        for the 2x2x2 case (torch's view + amax takes 1.2 ms on the 655 MB volume)."""
        dz, dy, dx = self.view_transformer.ds
        b, c, z, y, x = vol.shape
        if (vol.is_cuda and (dz, dy, dx) == (2, 2, 2) and vol.dtype == torch.float32
                and not torch.is_grad_enabled() and z % 2 == 0 and y % 2 == 0 and x % 2 == 0):
            from .. import _lib
            vol = vol.contiguous()
            out = torch.empty((b, c, z // 2, y // 2, x // 2), dtype=torch.float32,
                              device=vol.device)
            with _lib.on_device(vol.device):
                st = _lib.lib().veon_volume_maxpool2_f32(
                    _lib.ptr(vol), _lib.ptr(out), b * c, z, y, x, _lib.stream_ptr(vol.device))
            _lib.check(st, 'veon_volume_maxpool2_f32')
            return out
        return vol.view(b, c, z // dz, dz, y // dy, dy, x // dx, dx).amax(dim=(3, 5, 7))

    def from_volume(self, vol):
        """Everything after the (reduced) un-pooled lifted volume (B, C, Z, Y, X):
        ds_feat max-pool, Conv3d body, heads, classifier, arg-max."""
        return self.from_pooled(self._max_pool(vol))

    def from_pooled(self, pooled):
        """Conv3d body, heads, classifier, arg-max on the pooled volume
        (B, C, Z/dz, Y/dy, X/dx)."""
        from .. import conv3d_ops
        b, c = pooled.shape[:2]
        dec = self.occ_decoder
        if pooled.is_cuda and dec._fast_path(pooled[:, :1, 0]):
            lifted = dec._lift_volume(b, c, pooled.device)
            return self._tail(conv3d_ops.pack(pooled, out=lifted))
        xx = pooled
        for layer_3d in dec.layers_3d_body:
            xx = layer_3d(xx)
        return self._classify(dec.occupancy_pred(xx), dec.feat_pred(xx))

    def forward(self, images, img_metas, prev_volumes=None, depth=None, with_2d=False):
        """images (B, N, 3, H, W); img_metas = (sensor2egos, ego2globals, intrins,
        post_rots, post_trans, bda) as the reference's ``img[1:7]``;
        ``prev_volumes``: aligned lifted volumes of the past frames, newest first
        (the reference's ``occ_feat_prevs``).  Returns ``bin_occ`` / ``sem_occ`` at
        ``occ_size`` and ``occ_pred_cls``.  ``with_2d`` (needs ``side_adapter=True``):
        also the 2-D mask branch the reference always runs beside the 3-D one
        (san_in_veon_temporal.py:123-139), on the SAME CLIP features; its outputs come
        back under ``2d_*`` keys."""
        if with_2d and self.side_adapter_network is None:
            raise RuntimeError('VeonOccupancyPath was built without side_adapter=True')
        B, N = images.shape[:2]
        hf, wf = self.input_size[0] // 16, self.input_size[1] // 16
        sem_embed_ds = images.new_zeros((B * N, 1, hf, wf))   # shape carrier only
        metas = list(img_metas[:5]) + [img_metas[5][None]]
        dec = self.occ_decoder
        fast = (dec._fast_path(sem_embed_ds)
                and self.view_transformer._can_fuse_ds(sem_embed_ds))
        if fast:   # keep the sem head's output as a padded volume for the classifier
            feats, supp, depth, metas2, fused = self._branches(images, depth, metas)
            depth2 = dec.prepare_depth(depth)
            vol = dec._lift_volume(depth2.shape[0], dec.layers_3d_body[0].conv1.conv.in_channels,
                                   images.device)
            x = dec.fuse(0, None, feats, [supp], depth2, metas2, None, (hf, wf),
                         out_volume=vol, fused=fused)
            out = self._tail(x, prev_volumes)
            return self._add_2d(out, images, feats) if with_2d else out
        feats, supp, depth = self._branches(images, depth)
        out = dec(sem_embed_ds, feats, [supp], depth, metas, prev_volumes)
        out = self._classify(out['bin_occ'], out['feat_occ'])
        return self._add_2d(out, images, feats) if with_2d else out

    def _add_2d(self, out, images, feats):
        from .semantic_net.side_adapter import semantic_branch_2d
        two_d = semantic_branch_2d(self.side_adapter_network, self.clip_rec_head,
                                   self.ov_classifier_weight, images.flatten(0, 1), feats)
        out = dict(out)
        for k in ('mask_preds', 'mask_logits', 'sem_seg_ds', 'sem_embed_ds', 'sem_seg'):
            out['2d_' + k] = two_d[k]
        return out


class CameraShardedStep:
    """``VeonOccupancyPath.forward_camera_sharded`` for a fixed shape as hipGraph
    segments around the (eager) collectives: everything a rank computes before the
    exchange of the voxel volume replays from one graph, everything after it from
    another (``reduce='scatter'``: a third one for the sharded max-pool between the
    reduce-scatter and the all-gather).  The collectives themselves stay outside the
    graphs: they are two or three calls per step, and a captured RCCL call pins its
    communicator and buffers into the graph for no gain.  The collectives write
    straight into the next segment's static input, so no copy is added.

    ``step(images)`` -> the output dict of the tail (static tensors, overwritten by
    the next call).  Works without a process group (one rank, all cameras) and
    through a 1-rank group, which is how the GPU tests exercise it."""

    def __init__(self, net, images, img_metas, group=None, reduce_dtype=None,
                 reduce='allreduce'):
        import torch.distributed as dist
        from .. import sharding
        from ..graphs import GraphedCallable
        if reduce not in ('allreduce', 'scatter'):
            raise ValueError("reduce must be 'allreduce' or 'scatter', got %r" % (reduce,))
        self.dist, self.group = dist, group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        rank = dist.get_rank(group) if self.active else 0
        B, N = images.shape[:2]
        if B != 1:
            raise ValueError('CameraShardedStep shards the cameras of ONE sample (B = 1)')
        lo, hi = sharding.camera_slices(N, self.world)[rank]
        self.cameras = (lo, hi)
        C = net.occ_decoder.layers_3d_body[0].conv1.conv.in_channels
        self.scatter = reduce == 'scatter' and C % self.world == 0
        rd = reduce_dtype

        def pre(im):
            vol = net.lift_cameras(im, img_metas, lo, hi)
            return (vol if rd is None or rd == vol.dtype else vol.to(rd)).contiguous()

        def mid(mine):   # own channel slice, summed over the ranks -> pooled slice
            part = net._max_pool(mine.float())
            return (part if rd is None else part.to(rd)).contiguous()

        def post_pooled(allp):   # (world, B, C/world, z, y, x) -> (B, C, z, y, x)
            return net.from_pooled(allp.transpose(0, 1).reshape(B, C, *allp.shape[3:]).float())

        def post_volume(buf):
            return net.from_volume(buf.float())

        def wrap(f, ex):   # the example tensors themselves are the static inputs
            return GraphedCallable(f, ex, clone=False)
        with torch.no_grad():
            self.pre = wrap(pre, (images.clone(),))
            buf = self.pre.static_out
            if self.scatter:
                cs = C // self.world
                mine = torch.zeros((B, cs) + tuple(buf.shape[2:]), dtype=buf.dtype,
                                   device=buf.device)
                self.mid = wrap(mid, (mine,))
                part = self.mid.static_out
                allp = torch.zeros((self.world,) + tuple(part.shape), dtype=part.dtype,
                                   device=part.device)
                self.post = wrap(post_pooled, (allp,))
            else:
                self.mid = None
                self.post = wrap(post_volume, (buf,))
        self.launch = '%d hipGraph segments around the eager collectives' % (
            3 if self.scatter else 2)

    def __call__(self, images):
        dist = self.dist
        buf = self.pre(images)
        if self.scatter:
            mine = self.mid.static_in[0]
            if self.active:
                dist.reduce_scatter_tensor(mine.view(-1), buf.view(-1),
                                           op=dist.ReduceOp.SUM, group=self.group)
            else:
                mine.copy_(buf)
            part = self.mid.run()
            allp = self.post.static_in[0]
            if self.active:
                dist.all_gather_into_tensor(allp.view(-1), part.view(-1), group=self.group)
            else:
                allp[0].copy_(part)
            return self.post.run()
        if self.active:   # in place on the first segment's output = the tail's input
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return self.post.run()

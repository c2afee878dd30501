"""Shared Lift-Splat machinery of the view transformers.

The reference carries this code twice, byte-identical
(mmdet3d/models/necks/view_transformer.py:66-295 and
view_transformer_raw.py:74-337); here it lives once and both plugin classes
derive from it.  Method names, argument order, attributes and return values
are the reference's: ``create_grid_infos``, ``create_frustum``,
``get_lidar_coor``, ``init_acceleration_v2``, ``voxel_pooling_prepare_v2``,
``voxel_pooling_v2``, ``pre_compute``, ``view_transform_core``,
``view_transform``; ``D``, ``frustum``, ``grid_lower_bound``,
``grid_interval``, ``grid_size``, ``accelerate``, ``initial_flag``,
``ranks_*``, ``interval_*``.
"""
import torch
import torch.nn as nn

from ...ops.bev_pool_v2 import bev_pool as _bp
from ...ops.bev_pool_v2.bev_pool import bev_pool_v2
from ... import lss_prepare as _prep

try:  # pragma: no cover - mmcv is absent from the build image
    from mmcv.runner import BaseModule as _Base
except Exception:
    _Base = nn.Module


class LSSCore(_Base):
    """Geometry + index preparation + pooling; subclasses add the depth net."""

    # view_transformer.py returns (bev_feat, depth) from view_transform_core
    # (:284), view_transformer_raw.py only bev_feat (:332).
    _core_returns_depth = True

    def _init_lss(self, grid_config, input_size, downsample, out_channels,
                  accelerate, sid, collapse_z):
        self.grid_config = grid_config
        self.downsample = downsample
        self.create_grid_infos(**grid_config)
        self.sid = sid
        self.frustum = self.create_frustum(grid_config['depth'], input_size,
                                           downsample)
        self.out_channels = out_channels
        self.accelerate = accelerate
        self.initial_flag = True
        self.collapse_z = collapse_z
        # veon_amd extension (default off = the reference's exact behaviour):
        # run the per-call prepare without any host sync (inference only; an
        # empty grid then yields zeros of the regular shape instead of the
        # reference's odd-shaped dummy).  Makes accelerate=False capturable in
        # a hipGraph.
        self.sync_free = False
        # veon_amd extension (default off): with cached ranks (accelerate=True)
        # at inference, keep ONE output volume alive across calls and return it
        # every time (the caller must consume it before the next call -- what a
        # hipGraph replay implies anyway).  The buffer is picked once among a
        # few candidate allocations by timing the pool kernel on each
        # (veon_amd/placement.py: the same launch is ~15 % faster into a
        # well-placed allocation).
        self.persistent_output = False
        self._out_buf = None
        self.placement_info = None

    # ------------------------------------------------------------------ grid
    def create_grid_infos(self, x, y, z, **kwargs):
        """view_transformer_raw.py:74-89: float32 tensors, size = (hi-lo)/step
        evaluated in Python doubles first."""
        axes = (x, y, z)
        self.grid_lower_bound = torch.Tensor([a[0] for a in axes])
        self.grid_interval = torch.Tensor([a[2] for a in axes])
        self.grid_size = torch.Tensor([(a[1] - a[0]) / a[2] for a in axes])

    def create_frustum(self, depth_cfg, input_size, downsample):
        """view_transformer_raw.py:91-119 -> (D, Hf, Wf, 3) = (x_pix, y_pix, d)."""
        h_in, w_in = input_size
        hf, wf = h_in // downsample, w_in // downsample
        d = torch.arange(*depth_cfg, dtype=torch.float)
        self.D = d.shape[0]
        if self.sid:
            # spacing-increasing discretisation (:105-110)
            k = torch.arange(self.D).float()
            cfg = torch.tensor(depth_cfg).float()
            d = torch.exp(torch.log(cfg[0]) + k / (self.D - 1) *
                          torch.log((cfg[1] - 1) / cfg[0]))
        xs = torch.linspace(0, w_in - 1, wf, dtype=torch.float)
        ys = torch.linspace(0, h_in - 1, hf, dtype=torch.float)
        fr = torch.empty(self.D, hf, wf, 3, dtype=torch.float)
        fr[..., 0] = xs.view(1, 1, wf)
        fr[..., 1] = ys.view(1, hf, 1)
        fr[..., 2] = d.view(self.D, 1, 1)
        return fr

    # -------------------------------------------------------------- geometry
    def get_lidar_coor(self, sensor2ego, ego2global, cam2imgs, post_rots,
                       post_trans, bda):
        """Frustum points in the ego frame, (B, N, D, Hf, Wf, 3)
        (view_transformer_raw.py:121-158).  ``ego2global`` is accepted and
        unused, as in the reference."""
        return _prep.get_lidar_coor(self.frustum, sensor2ego, cam2imgs,
                                    post_rots, post_trans, bda)

    # ----------------------------------------------------------- index half
    def voxel_pooling_prepare_v2(self, coor):
        """view_transformer_raw.py:244-302 -> (ranks_bev, ranks_depth,
        ranks_feat, interval_starts, interval_lengths) int32, or 5 x None."""
        return _prep.voxel_pooling_prepare_v2(
            coor, self.grid_lower_bound, self.grid_interval, self.grid_size)

    def init_acceleration_v2(self, coor):
        """view_transformer_raw.py:196-215: cache the five rank tensors (plus,
        here, the fused kernels' tile plan and row table)."""
        ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths \
            = self.voxel_pooling_prepare_v2(coor)
        if ranks_bev is None:
            raise RuntimeError('accelerate=True but no frustum point falls '
                               'inside the grid')
        self.ranks_bev = ranks_bev.int().contiguous()
        self.ranks_feat = ranks_feat.int().contiguous()
        self.ranks_depth = ranks_depth.int().contiguous()
        self.interval_lengths = interval_lengths.int().contiguous()
        starts = interval_starts.int().contiguous()
        tag = getattr(interval_starts, '_veon_sorted', None)
        if starts is not interval_starts and tag is not None and tag[0] \
                and tag[3] == interval_starts._version:
            _bp.mark_sorted(starts, tag[1], tag[2])
        self.interval_starts = starts
        B = coor.shape[0]
        vpb = int(self.grid_size[2]) * int(self.grid_size[1]) * \
            int(self.grid_size[0])
        _bp.build_plan(self.ranks_bev, self.interval_starts, B, vpb)
        _bp.build_row_table(self.ranks_bev, self.interval_starts, B, vpb,
                            int(self.grid_size[0]))

    def _bev_feat_shape(self, B, C):
        zyx = self.__dict__.get('_grid_zyx')
        if zyx is None or zyx[0] is not self.grid_size:   # grid_size tensor -> ints, once
            gs = self.grid_size
            zyx = (gs, int(gs[2]), int(gs[1]), int(gs[0]))
            self.__dict__['_grid_zyx'] = zyx
        return (B, zyx[1], zyx[2], zyx[3], C)  # (B, Z, Y, X, C)

    def voxel_pooling_v2(self, coor, depth, feat):
        """view_transformer_raw.py:217-242."""
        ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths \
            = self.voxel_pooling_prepare_v2(coor)
        if ranks_feat is None:
            print('warning ---> no points within the predefined '
                  'bev receptive field')
            # the reference's dummy, including its (Z, X, Y) axis order (:224-231)
            dummy = torch.zeros(size=[
                feat.shape[0], feat.shape[2], int(self.grid_size[2]),
                int(self.grid_size[0]), int(self.grid_size[1])]).to(feat)
            return torch.cat(dummy.unbind(dim=2), 1)
        feat = feat.permute(0, 1, 3, 4, 2)
        bev_feat = bev_pool_v2(
            depth, feat, ranks_depth, ranks_feat, ranks_bev,
            self._bev_feat_shape(depth.shape[0], feat.shape[-1]),
            interval_starts, interval_lengths)
        if self.collapse_z:
            bev_feat = torch.cat(bev_feat.unbind(dim=2), 1)
        return bev_feat

    def _depth_table(self, depth):
        """-> (the float table the pool kernels index with ``ranks_depth``, extra
        arguments of the prepare).  A dense (B,N,D,H,W) tensor is its own table; a
        ``depth_ops.TwoHotWindows`` (the two-hot lift by construction) hands its
        compact weight table to the pool and its windows to the prepare."""
        from ... import depth_ops
        if isinstance(depth, depth_ops.TwoHotWindows):
            return depth.wts, dict(twohot=depth)
        return depth, self._sparse_args(depth)

    def _sparse_args(self, depth):
        """``sparse_depth_eps`` (opt-in, default None = every frustum point, the
        reference's sums to the bit): in the sync-free lift, points whose depth
        weight is below it are dropped before the sort -- VEON's soft two-hot depth
        puts ~1e-7 on all but a few of a pixel's D bins (include/veon_hip.h
        ``veon_lss_prepare_cameras_sparse``).  Pooled sums then differ by at most
        eps * sum|feat| over the dropped points."""
        eps = getattr(self, 'sparse_depth_eps', None)
        if not eps:
            return {}
        return dict(depth_weights=depth.contiguous().float(), depth_eps=float(eps))

    def _rows_beside_prepare(self, feat_l, depth):
        """Inside a hipGraph capture the feature layout change (NCHW -> pixel rows)
        is put on a forked stream, so the graph runs it BESIDE the prepare kernels
        it does not depend on; eager calls keep one stream (the fork would cost more
        host time than the 9 us kernel).  -> (rows, join) ; call join() before the
        pool launch."""
        if not (feat_l.is_cuda and torch.cuda.is_current_stream_capturing()):
            return feat_l, lambda: None
        cur = torch.cuda.current_stream(feat_l.device)
        side = self.__dict__.get('_fork_stream')
        if side is None or side.device != feat_l.device:
            side = self.__dict__['_fork_stream'] = torch.cuda.Stream(feat_l.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            rows = _bp._inference_feat(feat_l, depth)
        return rows, lambda: cur.wait_stream(side)

    def _lift_sync_free(self, input, depth, feat):
        self._drop_prepared()
        sensor2ego, _, cam2imgs, post_rots, post_trans, bda = input[1:7]
        depth, extra = self._depth_table(depth)
        feat_l, join = self._rows_beside_prepare(feat.permute(0, 1, 3, 4, 2), depth)
        pre = _prep._HIP_PREPARE.prepare_cameras(
            self.frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda,
            self.grid_lower_bound, self.grid_interval, self.grid_size,
            owner=self, **extra)
        join()
        shape = self._bev_feat_shape(depth.shape[0], feat.shape[2])
        out = None
        if self.persistent_output:
            out = self._persistent_volume(
                depth, feat, shape,
                probe=lambda o: _bp.bev_pool_v2_prepared(depth, feat_l, pre, shape,
                                                         out=o))
        bev_feat = _bp.bev_pool_v2_prepared(depth, feat_l, pre, shape, out=out)
        if self.collapse_z:
            bev_feat = torch.cat(bev_feat.unbind(dim=2), 1)
        return bev_feat

    def prepare_lift(self, metas):
        """Issue the per-call prepare (geometry + counting sort) of the sync-free lift
        NOW, on the current stream: it depends on the camera tensors only, so a caller
        that has them early (``VeonOccupancyPath``: before the encoders finish) can
        take it off the depth -> lift critical path.  ``metas`` = ``input[1:7]`` of the
        later ``forward`` call -- the SAME tensor objects; the stash is consumed by that
        call and ignored by any other.  No-op (None) when the lift is not sync-free or
        the sparse lift needs the depth."""
        if (self.accelerate or not self.sync_free or getattr(self, 'sparse_depth_eps', None)
                or not metas[0].is_cuda or torch.is_grad_enabled()):
            return None
        sensor2ego, _, cam2imgs, post_rots, post_trans, bda = metas[:6]
        pre = _prep._HIP_PREPARE.prepare_cameras(
            self.frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda,
            self.grid_lower_bound, self.grid_interval, self.grid_size, owner=self)
        # the tensors themselves are kept (not their ids: a freed tensor's id can be
        # reused by a new one), compared with `is` by the consumer
        self.__dict__['_prepared'] = (
            (sensor2ego, cam2imgs, post_rots, post_trans, bda), pre)
        return pre

    def _take_prepared(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        """The stash of ``prepare_lift`` if it was made from exactly these tensor
        objects; the stash is dropped either way (every lift entry point calls this
        or ``_drop_prepared`` first, so a stale one never survives a forward)."""
        st = self.__dict__.pop('_prepared', None)
        if st is not None and all(a is b for a, b in zip(
                st[0], (sensor2ego, cam2imgs, post_rots, post_trans, bda))):
            return st[1]
        return None

    def _drop_prepared(self):
        self.__dict__.pop('_prepared', None)

    def _lift_maxpool(self, input, depth, feat, ds, out_volume=None):
        """forward's pool + (dz,dy,dx) block max in one kernel (inference).
        depth (B,N,D,H,W), feat (B,N,C,H,W) -> (B,C,Z/dz,Y/dy,X/dx), or into
        ``out_volume`` (the Conv3d body's padded bf16 input)."""
        B = depth.shape[0]
        shape = self._bev_feat_shape(B, feat.shape[2])
        feat = feat.permute(0, 1, 3, 4, 2)
        if self.accelerate or not self.sync_free:
            self._drop_prepared()
        if self.accelerate:
            self.pre_compute(input)
            return _bp.bev_pool_v2_maxpool(
                depth, feat, self.ranks_depth, self.ranks_feat, self.ranks_bev,
                shape, self.interval_starts, self.interval_lengths, ds,
                out_volume=out_volume)
        sensor2ego, _, cam2imgs, post_rots, post_trans, bda = input[1:7]
        if self.sync_free:
            depth, extra = self._depth_table(depth)
            pre = None if extra else self._take_prepared(sensor2ego, cam2imgs, post_rots,
                                                         post_trans, bda)
            if extra:
                self._drop_prepared()
            if pre is None:
                feat, join = self._rows_beside_prepare(feat, depth)
                pre = _prep._HIP_PREPARE.prepare_cameras(
                    self.frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda,
                    self.grid_lower_bound, self.grid_interval, self.grid_size,
                    owner=self, **extra)
                join()
            return _bp.bev_pool_v2_maxpool(
                depth, feat, pre.ranks_depth, pre.ranks_feat, pre.ranks_bev,
                shape, pre.interval_starts, pre.interval_lengths, ds,
                counts=pre.counts, out_volume=out_volume, vstart=pre.vstart)
        pri, comb, trans = _prep.camera_matrices(sensor2ego, cam2imgs, post_rots)
        ranks = _prep.prepare_from_matrices(
            self.frustum, pri, post_trans, comb, trans, bda,
            self.grid_lower_bound, self.grid_interval, self.grid_size)
        if ranks[0] is None:
            return None
        rb, rd, rf, st, ln = ranks
        return _bp.bev_pool_v2_maxpool(depth, feat, rd, rf, rb, shape, st, ln, ds,
                                       out_volume=out_volume)

    # ---------------------------------------------------------- entry points
    def pre_compute(self, input):
        if self.initial_flag:
            coor = self.get_lidar_coor(*input[1:7])
            self.init_acceleration_v2(coor)
            self.initial_flag = False

    def view_transform_core(self, input, depth, tran_feat):
        B, N, C, H, W = input[0].shape
        if self.accelerate:
            feat = tran_feat.view(B, N, self.out_channels, H, W)
            feat = feat.permute(0, 1, 3, 4, 2)
            depth = depth.view(B, N, self.D, H, W)
            shape = self._bev_feat_shape(B, feat.shape[-1])
            out = None
            if (self.persistent_output and feat.is_cuda
                    and not torch.is_grad_enabled()):
                out = self._persistent_volume(depth, feat, shape)
            bev_feat = bev_pool_v2(
                depth, feat, self.ranks_depth, self.ranks_feat, self.ranks_bev,
                shape, self.interval_starts, self.interval_lengths, out=out)
            bev_feat = bev_feat.squeeze(2)
        elif (self.sync_free and tran_feat.is_cuda
              and not torch.is_grad_enabled()):
            from ... import depth_ops
            bev_feat = self._lift_sync_free(
                input, depth if isinstance(depth, depth_ops.TwoHotWindows)
                else depth.view(B, N, self.D, H, W),
                tran_feat.view(B, N, self.out_channels, H, W))
        else:
            coor = self.get_lidar_coor(*input[1:7])
            bev_feat = self.voxel_pooling_v2(
                coor, depth.view(B, N, self.D, H, W),
                tran_feat.view(B, N, self.out_channels, H, W))
        if self._core_returns_depth:
            return bev_feat, depth
        return bev_feat

    def _persistent_volume(self, depth, feat, shape, probe=None):
        Bv, Z, Y, X, C = (int(v) for v in shape)
        want = (Bv, C, Z, Y, X)
        if self._out_buf is None or tuple(self._out_buf.shape) != want \
                or self._out_buf.device != feat.device:
            from ... import placement

            if probe is None:
                def probe(o):
                    bev_pool_v2(depth, feat, self.ranks_depth, self.ranks_feat,
                                self.ranks_bev, shape, self.interval_starts,
                                self.interval_lengths, out=o)
            self._out_buf, self.placement_info = placement.best_placed(
                probe, want, torch.float32, feat.device)
        return self._out_buf

    def view_transform(self, input, depth, tran_feat):
        if self.accelerate:
            self.pre_compute(input)
        return self.view_transform_core(input, depth, tran_feat)

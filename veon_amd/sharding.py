"""Multi-GPU forms of the lift (one process per GPU, ``torch.distributed`` over
RCCL/xGMI; nothing here exists in the reference, whose only parallelism is
one-sample-per-GPU DDP -- SURVEY 8e).

* replicas      : samples are independent -> no data-path collective
                  (``replica_slice``).  This is what ``bench.py --gpus N`` runs.
* camera shards : the pooled volume is a SUM over frustum points and every point
                  belongs to one camera, so V = sum_cam V_cam: each rank lifts
                  its cameras into a full-size volume, then ONE all-reduce(SUM).
                  The ds_feat max-pool must come after the all-reduce (max does
                  not commute with the cross-camera sum).  Worth it only when
                  the per-camera encoders dominate: the message is the whole
                  volume (205 MB for S2, 655 MB for SV in fp32).
"""
import torch
import torch.distributed as dist


def replica_slice(n_items, rank, world):
    """Contiguous balanced slice [lo, hi) of ``n_items`` for ``rank``."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def camera_slices(n_cams, world):
    """Per-rank camera ranges; ranks beyond the camera count get (k, k)."""
    return [replica_slice(n_cams, r, world) for r in range(world)]


def slice_cameras(input, depth, lo, hi):
    """Restrict a view-transformer ``input`` list (feat, sensor2ego, ego2global,
    intrins, post_rots, post_trans, bda[, ...]) and ``depth`` (B,N,D,H,W) to
    cameras [lo, hi).  ``bda`` is per sample, not per camera."""
    out = [input[0][:, lo:hi].contiguous()]
    for t in input[1:6]:
        out.append(t[:, lo:hi].contiguous())
    out.extend(input[6:])
    return out, depth[:, lo:hi].contiguous()


class CameraShardedLift(torch.nn.Module):
    """Wraps a view transformer so that each rank lifts only its cameras and the
    full volume is obtained by all-reduce.  ``lift_fn(vt, input, depth)`` must
    return the *un-pooled* (B,C,Z,Y,X) volume of the given cameras; the default
    runs the transformer with its max-pool disabled."""

    def __init__(self, view_transformer, group=None, lift_fn=None,
                 reduce_dtype=None):
        super().__init__()
        self.vt = view_transformer
        self.group = group
        self.lift_fn = lift_fn or self._default_lift
        self.reduce_dtype = reduce_dtype

    @staticmethod
    def _default_lift(vt, input, depth):
        B, N, C, H, W = input[0].shape
        out = vt.view_transform(input, depth.reshape(B * N, -1, H, W),
                                input[0].reshape(B * N, C, H, W))
        return out[0] if isinstance(out, tuple) else out

    def forward(self, input, depth):
        active = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(self.group) if active else 1
        rank = dist.get_rank(self.group) if active else 0
        n_cams = input[0].shape[1]
        lo, hi = camera_slices(n_cams, world)[rank]
        vol = None
        if hi > lo:
            local_in, local_depth = slice_cameras(input, depth, lo, hi)
            vol = self.lift_fn(self.vt, local_in, local_depth)
        shape = self._volume_shape(input)
        if vol is None:  # more ranks than cameras: contribute zeros
            vol = torch.zeros(shape, dtype=torch.float32, device=input[0].device)
        vol = self._as_volume(vol, shape)
        if world > 1:
            if self.reduce_dtype is not None and self.reduce_dtype != vol.dtype:
                buf = vol.to(self.reduce_dtype)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                vol = buf.to(torch.float32)
            else:
                dist.all_reduce(vol, op=dist.ReduceOp.SUM, group=self.group)
        ds = getattr(self.vt, 'ds', None)
        if ds is not None and getattr(self.vt, 'use_ds', False):
            dz, dy, dx = ds
            b, c, z, y, x = vol.shape
            vol = vol.view(b, c, z // dz, dz, y // dy, dy, x // dx, dx) \
                .amax(dim=(3, 5, 7))
        return vol

    @staticmethod
    def _as_volume(vol, shape):
        """The lift's result as (B, C, Z, Y, X).  The view transformers may return
        the volume with Z folded into the channels (``collapse_z``: index z*C + c,
        view_transformer_raw.py:240-241), with a unit Z squeezed (the accelerate
        branch, :325) or -- when no frustum point falls inside the grid -- the
        reference's all-zero dummy of shape (B, C*Z, X, Y) (:221-231).  Anything else
        is an error: silently contributing zeros would corrupt the reduced volume."""
        B, C, Z, Y, X = shape
        if tuple(vol.shape) == tuple(shape):
            return vol
        if vol.dim() == 4 and tuple(vol.shape) == (B, C * Z, Y, X):
            return vol.view(B, Z, C, Y, X).permute(0, 2, 1, 3, 4).contiguous()
        if vol.dim() == 4 and tuple(vol.shape) == (B, C * Z, X, Y) and not bool(vol.any()):
            return vol.new_zeros(shape)     # the empty-grid dummy (X != Y)
        raise ValueError('camera-sharded lift: the view transformer returned shape %r, '
                         'expected %r (or its collapse_z / squeezed / empty-grid forms)'
                         % (tuple(vol.shape), tuple(shape)))

    def _volume_shape(self, input):
        vt = self.vt
        B = input[0].shape[0]
        C = input[0].shape[2]
        return (B, C, int(vt.grid_size[2]), int(vt.grid_size[1]),
                int(vt.grid_size[0]))

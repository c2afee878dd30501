"""HIP implementation of the index half (csrc/lss_prepare.hip) behind
``veon_amd.lss_prepare``; imported for its side effect of registering itself as
the device path.  Raises if libveon_hip.so is missing (no fallback)."""
import ctypes

import torch

from . import _lib, lss_prepare
from .ops.bev_pool_v2 import bev_pool as _bp
from .ops.bev_pool_v2.bev_pool import mark_sorted

_F3 = ctypes.c_float * 3


def _axes(frustum, device):
    """Device copies of the frustum axes xs[W], ys[H], ds[D], cached on the
    (CPU) frustum tensor object."""
    cache = getattr(frustum, '_veon_axes', None)
    if cache is None:
        cache = {}
        frustum._veon_axes = cache
    key = str(device)
    if key not in cache:
        fr = frustum.detach().float().cpu()
        cache[key] = (fr[0, 0, :, 0].contiguous().to(device),
                      fr[0, :, 0, 1].contiguous().to(device),
                      fr[:, 0, 0, 2].contiguous().to(device))
    return cache[key]


def _f32c(t):
    return t.contiguous().float()


def lidar_coor_from_matrices(frustum, post_rots_inv, post_trans, combine, trans,
                             bda):
    dev = _lib.require_device(post_rots_inv, post_trans, combine, trans, bda)
    B, N = combine.shape[:2]
    D, H, W, _ = frustum.shape
    xs, ys, ds = _axes(frustum, dev)
    pri, pt, cb, tr, bd = (_f32c(t) for t in (post_rots_inv, post_trans, combine,
                                              trans, bda))
    coor = torch.empty((B, N, D, H, W, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_lidar_coor(
            B, N, D, H, W, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ds),
            _lib.ptr(pri), _lib.ptr(pt), _lib.ptr(cb), _lib.ptr(tr),
            _lib.ptr(bd), _lib.ptr(coor), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_lidar_coor')
    return coor


def camera_matrices(sensor2ego, cam2imgs, post_rots):
    """Stream-capturable camera algebra (csrc k_camera_matrices)."""
    dev = _lib.require_device(sensor2ego, cam2imgs, post_rots)
    B, N = sensor2ego.shape[:2]
    s2e, k, pr = _f32c(sensor2ego), _f32c(cam2imgs), _f32c(post_rots)
    pri = torch.empty((B, N, 3, 3), dtype=torch.float32, device=dev)
    comb = torch.empty((B, N, 3, 3), dtype=torch.float32, device=dev)
    trans = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_camera_matrices(
            B * N, _lib.ptr(s2e), _lib.ptr(k), _lib.ptr(pr), _lib.ptr(pri),
            _lib.ptr(comb), _lib.ptr(trans), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_camera_matrices')
    return pri, comb, trans


def sensor2keyego(sensor2ego, ego2global):
    """(B,N,4,4) x 2 -> (B,N,4,4): inverse(ego2global[:, 0]) @ ego2global @ sensor2ego in
    double precision, one launch (csrc k_sensor2keyego)."""
    dev = _lib.require_device(sensor2ego, ego2global)
    B, N = sensor2ego.shape[:2]
    s2e, e2g = _f32c(sensor2ego), _f32c(ego2global)
    out = torch.empty((B, N, 4, 4), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        st = _lib.lib().veon_sensor2keyego(B, N, _lib.ptr(s2e), _lib.ptr(e2g), _lib.ptr(out),
                                           _lib.stream_ptr(dev))
    _lib.check(st, 'veon_sensor2keyego')
    return out


class Prepared:
    """Full-capacity device buffers of one prepare call (no host sync yet).
    ``vstart``: the dense voxel table of the row pool kernels (None when the call
    did not produce it)."""
    __slots__ = ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts',
                 'interval_lengths', 'plan', 'counts', 'batch', 'vpb', 'vstart')


class LiftWorkspace(Prepared):
    """Every device buffer of the per-call, sync-free lift for one problem size,
    allocated ONCE (and the histogram zeroed once): a steady-state call allocates
    nothing, so a hipGraph capture of it owns no memory, and it needs no memset
    node.  The buffers are overwritten by the next call on the same stream."""
    __slots__ = ('ws', 'ws_bytes', 'dims', 'dirty')

    def __init__(self, dims, vpb, device):
        B, N, D, H, W = dims
        L = _lib.lib()
        P = B * N * D * H * W
        self.dims, self.batch, self.vpb = tuple(dims), B, vpb
        # the histogram is zero between calls (every call re-zeroes it); a call that
        # fails part-way leaves it dirty, and the next call then takes the memset path
        self.dirty = False
        self.ws_bytes = L.veon_lss_prepare_workspace_bytes(P, vpb * B)
        self.ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.ranks_bev = torch.empty(P, **i32)
        self.ranks_depth = torch.empty(P, **i32)
        self.ranks_feat = torch.empty(P, **i32)
        self.interval_starts = torch.empty(P, **i32)
        self.interval_lengths = torch.empty(P, **i32)
        self.counts = torch.zeros(2, **i32)
        self.vstart = torch.zeros(L.veon_bev_pool_voxel_table_ints(B, vpb), **i32)
        self.plan = None
        if vpb % 64 == 0:
            self.plan = torch.zeros(L.veon_bev_pool_plan_ints(B, vpb), **i32)


_WORKSPACES = {}


def lift_workspace(dims, vpb, device, owner=None):
    """The LiftWorkspace of (problem size, device, current stream): two streams
    lifting at once never share buffers.  ``owner`` (a view transformer, a graphed
    callable ...) holds its workspaces itself: they die with it and two owners never
    share buffers, also when they capture on one stream; owner-less calls use a
    module-level table that ``clear_workspaces()`` empties."""
    key = (tuple(dims), int(vpb), str(device),
           int(torch._C._cuda_getCurrentRawStream(
               device.index if device.index is not None else torch.cuda.current_device())))
    table = _WORKSPACES
    if owner is not None:
        table = owner.__dict__.setdefault('_veon_lift_workspaces', {})
    ws = table.get(key)
    if ws is None:
        with torch.cuda.device(device):
            ws = LiftWorkspace(dims, int(vpb), device)
        table[key] = ws
    return ws


def clear_workspaces(owner=None):
    """Drop the cached workspaces (of ``owner``, or the module-level ones).  A live
    hipGraph that captured a workspace keeps its buffers alive through torch's
    graph pool only if they were allocated in the capture -- they are not: do not
    clear while such a graph is still replayed."""
    if owner is not None:
        owner.__dict__.pop('_veon_lift_workspaces', None)
    else:
        _WORKSPACES.clear()


def _grid_host(lower, interval, gsize):
    return (_F3(*[float(v) for v in lower.tolist()]),
            _F3(*[float(v) for v in interval.tolist()]),
            _F3(*[float(v) for v in gsize.tolist()]))


def _vpb(gsize):
    return int(gsize[2]) * int(gsize[1]) * int(gsize[0])


def prepare_device(dims, coor, geometry, lower, interval, gsize, device):
    """Launch the prepare pipeline; returns ``Prepared`` (capacity-sized
    buffers + device counts).  ``geometry`` = (frustum, pri, post_trans,
    combine, trans, bda) when ``coor`` is None."""
    B, N, D, H, W = dims
    L = _lib.lib()
    P = B * N * D * H * W
    vpb = _vpb(gsize)
    ws_bytes = L.veon_lss_prepare_workspace_bytes(P, vpb * B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    out = Prepared()
    out.vstart = None
    out.batch, out.vpb = B, vpb
    out.ranks_bev = torch.empty(P, dtype=torch.int32, device=device)
    out.ranks_depth = torch.empty(P, dtype=torch.int32, device=device)
    out.ranks_feat = torch.empty(P, dtype=torch.int32, device=device)
    out.interval_starts = torch.empty(P, dtype=torch.int32, device=device)
    out.interval_lengths = torch.empty(P, dtype=torch.int32, device=device)
    out.counts = torch.empty(2, dtype=torch.int32, device=device)
    out.plan = None
    if vpb % 64 == 0:
        out.plan = torch.empty(L.veon_bev_pool_plan_ints(B, vpb),
                               dtype=torch.int32, device=device)
    glo, gstep, gsz = _grid_host(lower, interval, gsize)
    null = ctypes.c_void_p(0)
    if coor is not None:
        geo = [null] * 8
        coor_p = _lib.ptr(coor)
        keep = (coor,)
    else:
        frustum, pri, pt, cb, tr, bd = geometry
        xs, ys, ds = _axes(frustum, device)
        keep = tuple(_f32c(t) for t in (pri, pt, cb, tr, bd)) + (xs, ys, ds)
        geo = [_lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ds)] + \
            [_lib.ptr(t) for t in keep[:5]]
        coor_p = null
    with torch.cuda.device(device):
        st = L.veon_lss_prepare(
            B, N, D, H, W, coor_p, *geo,
            ctypes.cast(glo, ctypes.c_void_p), ctypes.cast(gstep, ctypes.c_void_p),
            ctypes.cast(gsz, ctypes.c_void_p), vpb, _lib.ptr(ws), ws_bytes,
            _lib.ptr(out.ranks_bev), _lib.ptr(out.ranks_depth),
            _lib.ptr(out.ranks_feat), _lib.ptr(out.interval_starts),
            _lib.ptr(out.interval_lengths), _lib.ptr(out.plan),
            _lib.ptr(out.counts), _lib.stream_ptr(device))
    _lib.check(st, 'veon_lss_prepare')
    del keep
    return out


def prepare_cameras(frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda, lower,
                    interval, gsize, depth_weights=None, depth_eps=0.0, owner=None,
                    twohot=None):
    """The sync-free per-call prepare from the reference's camera tensors
    (get_lidar_coor's arguments) into the static LiftWorkspace: five launches, no
    allocation, no memset, no host sync.  Returns the workspace (a ``Prepared``
    with ``vstart``)."""
    dev = _lib.require_device(sensor2ego, cam2imgs, post_rots, post_trans, bda)
    B, N = sensor2ego.shape[:2]
    D, H, W, _ = frustum.shape
    vpb = _vpb(gsize)
    ws = lift_workspace((B, N, D, H, W), vpb, dev, owner)
    xs, ys, ds = _axes(frustum, dev)
    s2e, k, pr, pt, bd = (_f32c(t) for t in (sensor2ego, cam2imgs, post_rots, post_trans,
                                              bda))
    glo, gstep, gsz = _grid_host(lower, interval, gsize)
    args = (B, N, D, H, W, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ds), _lib.ptr(s2e),
            _lib.ptr(k), _lib.ptr(pr), _lib.ptr(pt), _lib.ptr(bd),
            ctypes.cast(glo, ctypes.c_void_p), ctypes.cast(gstep, ctypes.c_void_p),
            ctypes.cast(gsz, ctypes.c_void_p), vpb, _lib.ptr(ws.ws), ws.ws_bytes,
            0 if ws.dirty else 1,   # hist_is_zero
            _lib.ptr(ws.ranks_bev), _lib.ptr(ws.ranks_depth), _lib.ptr(ws.ranks_feat),
            _lib.ptr(ws.interval_starts), _lib.ptr(ws.interval_lengths),
            _lib.ptr(ws.plan), _lib.ptr(ws.vstart), _lib.ptr(ws.counts))
    ws.dirty = True   # cleared below once every launch was accepted
    with _lib.on_device(dev):
        if twohot is not None:
            # two-hot lift by construction: ``twohot`` = depth_ops.TwoHotWindows; the
            # returned ranks_depth index its compact weight table ``twohot.wts``
            if (tuple(twohot.shape) != (B, N, D, H, W) or twohot.device != dev
                    or twohot.win.dtype != torch.int32 or not twohot.win.is_contiguous()
                    or not twohot.wts.is_contiguous()):
                raise _lib.VeonHipError(
                    'two-hot windows of shape %r on %s do not match the frustum %r on %s'
                    % (tuple(twohot.shape), twohot.device, (B, N, D, H, W), dev))
            st = _lib.lib().veon_lss_prepare_cameras_twohot(
                *args, _lib.ptr(twohot.win), int(twohot.K), _lib.stream_ptr(dev))
            _lib.check(st, 'veon_lss_prepare_cameras_twohot')
        elif depth_weights is not None and depth_eps > 0.0:
            # sparse lift: points whose depth weight is below depth_eps are not sorted
            if (depth_weights.dtype != torch.float32 or not depth_weights.is_contiguous()
                    or depth_weights.numel() != B * N * D * H * W
                    or depth_weights.device != dev):
                raise _lib.VeonHipError('depth_weights must be a contiguous fp32 '
                                        '(B,N,D,H,W) tensor on the rig\'s device')
            st = _lib.lib().veon_lss_prepare_cameras_sparse(
                *args, _lib.ptr(depth_weights), float(depth_eps), _lib.stream_ptr(dev))
            _lib.check(st, 'veon_lss_prepare_cameras_sparse')
        else:
            st = _lib.lib().veon_lss_prepare_cameras(*args, _lib.stream_ptr(dev))
            _lib.check(st, 'veon_lss_prepare_cameras')
    # a call captured into a hipGraph while the workspace was dirty has the memset in
    # the graph; every replay then leaves a zero histogram like any other call
    ws.dirty = False
    return ws


def _finish(pre):
    """The reference contract: exactly-sized tensors (one host sync, as the
    reference's own torch.where / len() calls), or 5 x None when empty."""
    kept, n_int = (int(v) for v in pre.counts.tolist())
    if kept == 0 or n_int == 0:
        return None, None, None, None, None
    rb = pre.ranks_bev[:kept]
    starts = pre.interval_starts[:n_int]
    # first / last rank are only needed for the bounds tag: they are inside the
    # grid by construction
    mark_sorted(starts, 0, pre.batch * pre.vpb - 1)
    if pre.plan is not None:
        _bp._cache_put(starts, rb, '_veon_plan', (pre.batch, pre.vpb), pre.plan)
    return (rb, pre.ranks_depth[:kept], pre.ranks_feat[:kept], starts,
            pre.interval_lengths[:n_int])


def voxel_pooling_prepare_v2(coor, lower, interval, gsize):
    dev = _lib.require_device(coor)
    B, N, D, H, W, _ = coor.shape
    pre = prepare_device((B, N, D, H, W), _f32c(coor), None, lower, interval,
                         gsize, dev)
    return _finish(pre)


def prepare_from_matrices(frustum, post_rots_inv, post_trans, combine, trans,
                          bda, lower, interval, gsize, sync=True):
    """Geometry fused into the prepare (coordinates never materialised)."""
    dev = _lib.require_device(post_rots_inv, post_trans, combine, trans, bda)
    B, N = combine.shape[:2]
    D, H, W, _ = frustum.shape
    pre = prepare_device((B, N, D, H, W), None,
                         (frustum, post_rots_inv, post_trans, combine, trans, bda),
                         lower, interval, gsize, dev)
    return _finish(pre) if sync else pre


import sys  # noqa: E402

lss_prepare._HIP_PREPARE = sys.modules[__name__]

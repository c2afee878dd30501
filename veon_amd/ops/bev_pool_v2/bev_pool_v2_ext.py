"""Mirror of the reference's pybind11 module ``bev_pool_v2_ext``
(mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:106-111): the same two entry points,
same argument order -- note ``(interval_lengths, interval_starts)``, swapped
with respect to the Python op (bev_pool.cpp:37-38, 83-84) -- lowered onto the
C ABI of libveon_hip.so.  Unlike the reference it validates dtype, device and
contiguity, launches on the current stream and raises on failure.
"""
import torch

from ... import _lib


def _chk(t, name, dtype):
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)


def _common(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
            interval_starts):
    _chk(depth, 'depth', torch.float32)
    _chk(feat, 'feat', torch.float32)
    for t, n in ((ranks_depth, 'ranks_depth'), (ranks_feat, 'ranks_feat'),
                 (ranks_bev, 'ranks_bev'), (interval_lengths, 'interval_lengths'),
                 (interval_starts, 'interval_starts')):
        _chk(t, n, torch.int32)
    if interval_lengths.numel() != interval_starts.numel():
        raise ValueError('interval_lengths / interval_starts size mismatch')


def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                        interval_lengths, interval_starts):
    """bev_pool.cpp:30-57.  ``out`` (B,Z,Y,X,C) zero-filled by the caller."""
    _common(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
            interval_starts)
    _chk(out, 'out', torch.float32)
    dev = _lib.require_device(depth, feat, out, ranks_depth, ranks_feat,
                              ranks_bev, interval_lengths, interval_starts)
    c = feat.size(4) if feat.dim() == 5 else feat.size(-1)
    n_intervals = interval_lengths.size(0)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_bev_pool_v2_fwd(
            c, n_intervals, _lib.ptr(depth), _lib.ptr(feat),
            _lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev),
            _lib.ptr(interval_starts), _lib.ptr(interval_lengths),
            _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_bev_pool_v2_fwd')


def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat,
                         ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                         interval_starts):
    """bev_pool.cpp:74-104.  Grad buffers zero-filled by the caller."""
    _common(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
            interval_starts)
    for t, n in ((out_grad, 'out_grad'), (depth_grad, 'depth_grad'),
                 (feat_grad, 'feat_grad')):
        _chk(t, n, torch.float32)
    dev = _lib.require_device(out_grad, depth_grad, feat_grad, depth, feat,
                              ranks_depth, ranks_feat, ranks_bev,
                              interval_lengths, interval_starts)
    c = out_grad.size(4) if out_grad.dim() == 5 else out_grad.size(-1)
    n_intervals = interval_lengths.size(0)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_bev_pool_v2_bwd(
            c, n_intervals, _lib.ptr(out_grad), _lib.ptr(depth), _lib.ptr(feat),
            _lib.ptr(ranks_depth), _lib.ptr(ranks_feat), _lib.ptr(ranks_bev),
            _lib.ptr(interval_starts), _lib.ptr(interval_lengths),
            _lib.ptr(depth_grad), _lib.ptr(feat_grad), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_bev_pool_v2_bwd')

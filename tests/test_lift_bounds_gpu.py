"""Out-of-bounds WRITE check of every kernel of the per-call lift: all buffers the
kernels write are carved out of one arena with canary bands around them; after
the prepare + pool (+ max-pool) calls the bands must be untouched.  (A lift graph
once faulted with "write access to a read-only page" when another graph was alive
-- an overrun past the end of an allocation is exactly what only shows once the
neighbouring pages change owner.)  Shapes are chosen so that no size is a multiple
of the kernels' block sizes."""
import ctypes

import numpy as np
import pytest
import torch

from veon_amd import _lib, conv3d_ops, synthetic
from veon_amd.models import build_neck

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
CANARY = 0x5A
BAND = 4096


class Arena:
    def __init__(self, nbytes):
        self.buf = torch.full((nbytes,), CANARY, dtype=torch.uint8, device=DEV)
        self.off = BAND
        self.used = []

    def take(self, n, dtype, zero=False):
        nb = n * torch.empty((), dtype=dtype).element_size()
        start = (self.off + 255) // 256 * 256
        t = self.buf[start:start + nb].view(dtype)
        if zero:
            t.zero_()
        self.used.append((start, start + nb))
        self.off = start + nb + BAND
        assert self.off + BAND < self.buf.numel()
        return t

    def check(self):
        mask = torch.ones(self.buf.numel(), dtype=torch.bool, device=DEV)
        for a, b in self.used:
            mask[a:b] = False
        bad = (self.buf != CANARY) & mask
        assert not bool(bad.any()), 'canary overwritten at byte offsets %s (buffers %s)' % (
            torch.nonzero(bad).flatten()[:8].tolist(), self.used)


@pytest.mark.parametrize('C', [24, 256])
def test_per_call_lift_writes_stay_inside_their_buffers(C):
    grid = {'x': [-11.0, 11.0, 1.0], 'y': [-9.0, 9.0, 1.0], 'z': [-1.0, 5.0, 1.0],
            'depth': [1.0, 14.0, 1.0]}          # 22 x 18 x 6 voxels, D = 13
    size, cams = (80, 208), 3                    # 5 x 13 feature maps
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=grid, input_size=size,
                         out_channels=C, collapse_z=False, ds_feat=[2, 2, 2])).to(DEV).eval()
    B, N, D = 2, cams, vt.D
    hf, wf = size[0] // 16, size[1] // 16
    X, Y, Z = (int(v) for v in vt.grid_size)
    vpb = X * Y * Z
    P = B * N * D * hf * wf
    rig = synthetic.make_rig(B, cams, size)
    s2e, e2g, k, pr, pt, bda = (t.to(DEV).float().contiguous()
                                for t in synthetic.rig_inputs(rig))
    L = _lib.lib()
    ar = Arena(64 << 20)
    ws_bytes = L.veon_lss_prepare_workspace_bytes(P, vpb * B)
    ws = ar.take(ws_bytes, torch.uint8, zero=True)
    rb, rd, rf, st, ln = (ar.take(P, torch.int32) for _ in range(5))
    counts = ar.take(2, torch.int32)
    vstart = ar.take(B * vpb + 1, torch.int32)
    fr = vt.frustum
    xs = fr[0, 0, :, 0].contiguous().to(DEV)
    ys = fr[0, :, 0, 1].contiguous().to(DEV)
    ds = fr[:, 0, 0, 2].contiguous().to(DEV)
    f3 = ctypes.c_float * 3
    glo, gst, gsz = (f3(*[float(v) for v in t.tolist()]) for t in
                     (vt.grid_lower_bound, vt.grid_interval, vt.grid_size))
    s = _lib.stream_ptr(DEV)
    st_ = L.veon_lss_prepare_cameras(
        B, N, D, hf, wf, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ds), _lib.ptr(s2e),
        _lib.ptr(k), _lib.ptr(pr), _lib.ptr(pt), _lib.ptr(bda),
        ctypes.cast(glo, ctypes.c_void_p), ctypes.cast(gst, ctypes.c_void_p),
        ctypes.cast(gsz, ctypes.c_void_p), vpb, _lib.ptr(ws), ws_bytes, 1,
        _lib.ptr(rb), _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(st), _lib.ptr(ln), None,
        _lib.ptr(vstart), _lib.ptr(counts), s)
    assert st_ == 0
    torch.cuda.synchronize()
    ar.check()
    kept, n_int = counts.tolist()
    assert 0 < kept <= P and 0 < n_int <= kept and int(vstart[-1]) == kept
    # the histogram is left zeroed for the next call
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B, N, D, hf, wf, generator=g).to(DEV)
    feat = torch.randn(B, N, hf, wf, C, generator=g).to(DEV)
    out = ar.take(B * C * vpb, torch.float32)
    if C % 2 == 0:
        st_ = L.veon_bev_pool_v2_fwd_rows(C, B, vpb, _lib.ptr(depth), _lib.ptr(feat), 0,
                                          _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(vstart),
                                          _lib.ptr(out), 0, feat.numel(), 0, s)
        assert st_ == 0
    mp = ar.take(B * C * (vpb // 8), torch.float32)
    st_ = L.veon_bev_pool_v2_fwd_rows_maxpool(
        C, B, Z, Y, X, 2, 2, 2, _lib.ptr(depth), _lib.ptr(feat), 0, _lib.ptr(rd),
        _lib.ptr(rf), _lib.ptr(vstart), _lib.ptr(mp), 0, feat.numel(), s)
    assert st_ == 0
    # the slab kernels on the same ranks (plan / row table built per call)
    plan = ar.take(L.veon_bev_pool_plan_ints(B, vpb), torch.int32)
    assert L.veon_bev_pool_plan(n_int, kept, B, vpb, _lib.ptr(rb), _lib.ptr(st),
                                _lib.ptr(counts), _lib.ptr(plan), s) == 0
    out2 = ar.take(B * C * vpb, torch.float32)
    assert L.veon_bev_pool_v2_fwd_fused_ex(
        C, n_int, B, vpb, _lib.ptr(depth), _lib.ptr(feat), 0, _lib.ptr(rd), _lib.ptr(rf),
        _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln), _lib.ptr(plan), _lib.ptr(out2),
        _lib.LAYOUT_BCZYX, s) == 0
    rows = ar.take(2 * (B * Z * Y + 1), torch.int32)
    assert L.veon_bev_pool_row_table(n_int, kept, B, vpb, X, _lib.ptr(rb), _lib.ptr(st),
                                     _lib.ptr(counts), _lib.ptr(rows),
                                     _lib.ptr(rows[B * Z * Y + 1:]), s) == 0
    mp2 = ar.take(B * C * (vpb // 8), torch.float32)
    assert L.veon_bev_pool_v2_fwd_maxpool_ex(
        C, n_int, B, Z, Y, X, 2, 2, 2, _lib.ptr(depth), _lib.ptr(feat), 0, _lib.ptr(rd),
        _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln), _lib.ptr(rows),
        _lib.ptr(mp2), s) == 0
    torch.cuda.synchronize()
    ar.check()
    if C % 2 == 0:
        assert torch.equal(out, out2)
    assert torch.equal(mp, mp2)
    # second call into the same workspace: same result (histogram was left zeroed)
    rb_first = rb[:kept].clone()
    assert L.veon_lss_prepare_cameras(
        B, N, D, hf, wf, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(ds), _lib.ptr(s2e),
        _lib.ptr(k), _lib.ptr(pr), _lib.ptr(pt), _lib.ptr(bda),
        ctypes.cast(glo, ctypes.c_void_p), ctypes.cast(gst, ctypes.c_void_p),
        ctypes.cast(gsz, ctypes.c_void_p), vpb, _lib.ptr(ws), ws_bytes, 1,
        _lib.ptr(rb), _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(st), _lib.ptr(ln), None,
        _lib.ptr(vstart), _lib.ptr(counts), s) == 0
    torch.cuda.synchronize()
    assert counts.tolist() == [kept, n_int] and torch.equal(rb[:kept], rb_first)
    ar.check()


def test_depth_prep_and_transpose_writes_stay_inside():
    """downsample_depth / two_hot_depth / feat transpose / padded max-pool output."""
    L = _lib.lib()
    s = _lib.stream_ptr(DEV)
    ar = Arena(32 << 20)
    BN, H, W, dsf, D = 5, 24, 40, 8, 13
    g = torch.Generator().manual_seed(1)
    metric = (1.0 + 12.0 * torch.rand(BN, H, W, generator=g)).to(DEV)
    down = ar.take(BN * (H // dsf) * (W // dsf), torch.float32)
    assert L.veon_downsample_depth(BN, H, W, dsf, _lib.ptr(metric), _lib.ptr(down), s) == 0
    two = ar.take(BN * D * (H // dsf) * (W // dsf), torch.float32)
    assert L.veon_two_hot_depth(BN, H // dsf, W // dsf, 0, D, 1.0, 1.0, 4.0, _lib.ptr(down),
                                _lib.ptr(two), s) == 0
    two2 = ar.take(BN * D * (H // dsf) * (W // dsf), torch.float32)
    assert L.veon_two_hot_depth(BN, H // dsf, W // dsf, dsf, D, 1.0, 1.0, 4.0,
                                _lib.ptr(metric), _lib.ptr(two2), s) == 0
    C, HW = 37, 5 * 13
    nchw = torch.randn(BN, C, HW, generator=g).to(DEV)
    nhwc = ar.take(BN * HW * C, torch.float32)
    assert L.veon_feat_nchw_to_nhwc(_lib.ptr(nchw), _lib.ptr(nhwc), 4, BN, C, HW, s) == 0
    torch.cuda.synchronize()
    ar.check()
    assert torch.equal(nhwc.view(BN, HW, C), nchw.permute(0, 2, 1).contiguous())
    assert torch.allclose(two, two2)

"""GPU parity of the HIP bev_pool_v2 kernels against the CPU oracle.

Every call goes through the C ABI of libveon_hip.so (via the op mirror).
Forward sums are serial fmaf chains in storage order on both sides, so the
comparison is BIT-EXACT (np.array_equal), not tolerance-based.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from tests import helpers
from tests.conftest import load_golden
from veon_amd import _lib, synthetic
from veon_amd.ops.bev_pool_v2 import bev_pool as bp
from veon_amd.ops.bev_pool_v2 import bev_pool_v2_ext as ext
from veon_amd.ops.bev_pool_v2.bev_pool import (QuickCumsumCuda, TRTBEVPoolv2,
                                               bev_pool_v2)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return helpers.t(a, DEV)


def _case_from_golden(name):
    g = load_golden(name)
    B, C = g['feat'].shape[0], g['feat'].shape[2]
    X, Y, Z = (int(v) for v in g['grid_size'])
    feat_nhwc = np.ascontiguousarray(g['feat'].transpose(0, 1, 3, 4, 2))
    ranks = tuple(g[k] for k in ('ranks_bev', 'ranks_depth', 'ranks_feat',
                                 'interval_starts', 'interval_lengths'))
    return g['two_hot'], feat_nhwc, ranks, (B, Z, Y, X, C)


def _synthetic_case(grid, input_size, n_cams, C, batch=1, seed=0):
    ranks, coor, rig, fr, gsize = helpers.oracle_ranks(grid, input_size,
                                                       n_cams, batch)
    D = fr.shape[0]
    hf, wf = input_size[0] // 16, input_size[1] // 16
    depth, feat = synthetic.make_depth_feat(batch, n_cams, D, C, hf, wf, seed)
    feat_nhwc = feat.permute(0, 1, 3, 4, 2).contiguous().numpy()
    shape = (batch, int(gsize[2]), int(gsize[1]), int(gsize[0]), C)
    return depth.numpy(), feat_nhwc, ranks, shape


def _oracle_out(depth, feat_nhwc, ranks, shape):
    rb, rd, rf, st, ln = ranks
    B, Z, Y, X, C = shape
    return c_oracle.bev_pool_v2_fwd(depth, feat_nhwc, rd, rf, rb, st, ln,
                                    B * Z * Y * X).reshape(B, Z, Y, X, C)


def _run_scatter(depth, feat_nhwc, ranks, shape):
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    out = torch.zeros(shape, dtype=torch.float32, device=DEV)
    ext.bev_pool_v2_forward(dev(depth), dev(feat_nhwc), out, rd, rf, rb, ln, st)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _run_fused(depth, feat_nhwc, ranks, shape, layout, table=False):
    """table=True: plan cached on interval_starts (accelerate path);
    False: plan rebuilt inside the call."""
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    if table:
        B, Z, Y, X, C = shape
        bp.build_plan(rb, st, B, Z * Y * X)
    # poison the allocator so "every element written once" is actually tested
    junk = torch.full((int(np.prod(shape)),), float('nan'), device=DEV)
    del junk
    out = bp._fused_forward(dev(depth), dev(feat_nhwc), rd, rf, rb, st, ln,
                            shape, layout)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_library_loaded_and_abi():
    L = _lib.lib()
    assert L.veon_abi_version() == 2
    assert L.veon_status_string(0) == b'ok'


def test_kat_forward_backward_through_op():
    """The reference's own KAT (bev_pool.py:145-176), on the HIP path."""
    g = load_golden('kat_bev_pool_v2')
    depth = dev(g['depth']).requires_grad_()
    feat = dev(g['feat']).requires_grad_()
    out = bev_pool_v2(depth, feat, dev(g['ranks_depth']), dev(g['ranks_feat']),
                      dev(g['ranks_bev']), tuple(int(v) for v in g['bev_feat_shape']),
                      dev(g['interval_starts']), dev(g['interval_lengths']))
    assert out.shape == (1, 2, 1, 2, 2) and out.is_contiguous()
    loss = out.sum()
    loss.backward()
    assert loss.item() == pytest.approx(4.4, abs=1e-6)
    assert np.float32(loss.item()) == g['expect_sum']
    assert torch.allclose(depth.grad.cpu(), torch.from_numpy(g['expect_depth_grad']))
    assert torch.allclose(feat.grad.cpu(), torch.from_numpy(g['expect_feat_grad']))


@pytest.mark.parametrize('name', ['lss_small', 'lss_small_b2', 'lss_mid'])
def test_scatter_and_fused_bit_exact_small(name):
    depth, feat_nhwc, ranks, shape = _case_from_golden(name)
    want = _oracle_out(depth, feat_nhwc, ranks, shape)
    assert np.array_equal(_run_scatter(depth, feat_nhwc, ranks, shape), want)
    for table in (False, True):
        got = _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BZYXC, table)
        assert np.array_equal(got, want)
        got = _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BCZYX, table)
        assert np.array_equal(got, want.transpose(0, 4, 1, 2, 3))


@pytest.mark.parametrize('grid,size,cams,C', [
    (synthetic.GRID_BEVDET, (256, 704), 1, 64),    # BASELINE configs[0]
    (synthetic.GRID_S2, (256, 704), 6, 80),        # BASELINE configs[1]
])
def test_full_size_bit_exact(grid, size, cams, C):
    depth, feat_nhwc, ranks, shape = _synthetic_case(grid, size, cams, C)
    want = _oracle_out(depth, feat_nhwc, ranks, shape)
    assert np.array_equal(_run_scatter(depth, feat_nhwc, ranks, shape), want)
    assert np.array_equal(
        _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BZYXC), want)
    assert np.array_equal(
        _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BCZYX, True),
        want.transpose(0, 4, 1, 2, 3))


def test_veon_shape_c256_bit_exact():
    """SV: 6 cams 512x1408, D=88, C=256 (655 MB volume), channel slabs."""
    depth, feat_nhwc, ranks, shape = _synthetic_case(
        synthetic.GRID_VEON, (512, 1408), 6, 256)
    want = _oracle_out(depth, feat_nhwc, ranks, shape)
    got = _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BCZYX)
    assert np.array_equal(got, want.transpose(0, 4, 1, 2, 3))
    del got
    got = _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BZYXC)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('C', [1, 3, 7, 80, 130])
def test_odd_channel_counts_and_ragged_tiles(C):
    """Scalar path (C % 4 != 0), slab path (C > 128) and a voxel count that is
    not a multiple of the tile (5*7*3 = 105 voxels)."""
    rng = np.random.default_rng(C)
    B, Z, Y, X = 2, 3, 7, 5
    nvox = B * Z * Y * X
    n_pts, n_feat, n_depth = 400, 30, 500
    rb = np.sort(rng.integers(0, nvox, n_pts)).astype(np.int32)
    rd = rng.integers(0, n_depth, n_pts).astype(np.int32)
    rf = rng.integers(0, n_feat, n_pts).astype(np.int32)
    st, ln = helpers.bp_intervals(rb)
    depth = rng.random((1, 1, n_depth, 1, 1), dtype=np.float32)
    feat = rng.standard_normal((1, 1, n_feat, 1, C)).astype(np.float32)
    shape = (B, Z, Y, X, C)
    ranks = (rb, rd, rf, st, ln)
    want = _oracle_out(depth, feat, ranks, shape)
    assert np.array_equal(_run_scatter(depth, feat, ranks, shape), want)
    assert np.array_equal(_run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BZYXC), want)
    assert np.array_equal(_run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BCZYX),
                          want.transpose(0, 4, 1, 2, 3))


def test_empty_and_single_interval():
    shape = (1, 2, 3, 4, 8)
    depth = np.ones((1, 1, 4, 1, 1), np.float32)
    feat = np.ones((1, 1, 2, 1, 8), np.float32)
    empty = tuple(np.zeros(0, np.int32) for _ in range(5))
    for layout in (_lib.LAYOUT_BZYXC, _lib.LAYOUT_BCZYX):
        assert not _run_fused(depth, feat, empty, shape, layout).any()
    one = (np.array([23], np.int32), np.array([1], np.int32),
           np.array([0], np.int32), np.array([0], np.int32),
           np.array([1], np.int32))
    got = _run_fused(depth, feat, one, shape, _lib.LAYOUT_BZYXC).reshape(24, 8)
    assert (got[23] == 1).all() and not got[:23].any()


def test_unsorted_intervals_take_reference_structure():
    """Hand-made inputs whose intervals are not ascending cannot use the fused
    kernel; the op must still return the reference's result."""
    depth, feat_nhwc, ranks, shape = _case_from_golden('lss_small')
    rb, rd, rf, st, ln = ranks
    perm = np.random.default_rng(0).permutation(len(st))
    # rebuild the point arrays interval by interval in shuffled order
    segs = [np.arange(st[i], st[i] + ln[i]) for i in perm]
    order = np.concatenate(segs)
    rb2, rd2, rf2 = rb[order], rd[order], rf[order]
    ln2 = ln[perm]
    st2 = (np.cumsum(ln2) - ln2).astype(np.int32)
    out = bev_pool_v2(dev(depth), dev(feat_nhwc), dev(rd2), dev(rf2), dev(rb2),
                      shape, dev(st2), dev(ln2))
    want = _oracle_out(depth, feat_nhwc, ranks, shape).transpose(0, 4, 1, 2, 3)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize('name', ['lss_small_b2', 'lss_mid'])
def test_backward_bit_exact(name):
    depth, feat_nhwc, ranks, shape = _case_from_golden(name)
    rb, rd, rf, st, ln = ranks
    rng = np.random.default_rng(1)
    og = rng.standard_normal(shape).astype(np.float32)
    want_dg, want_fg = helpers.oracle_backward(og, depth, feat_nhwc, rd, rf, rb)
    d = dev(depth).requires_grad_()
    f = dev(feat_nhwc).requires_grad_()
    out = QuickCumsumCuda.apply(d, f, dev(rd), dev(rf), dev(rb), shape,
                                dev(st), dev(ln))
    assert out.shape == shape
    out.backward(dev(og))
    assert np.array_equal(d.grad.cpu().numpy(), want_dg)
    assert np.array_equal(f.grad.cpu().numpy(), want_fg)
    # and through the channels-first op (out_grad arrives as (B,C,Z,Y,X))
    d2 = dev(depth).requires_grad_()
    f2 = dev(feat_nhwc).requires_grad_()
    out2 = bev_pool_v2(d2, f2, dev(rd), dev(rf), dev(rb), shape, dev(st), dev(ln))
    out2.backward(dev(og.transpose(0, 4, 1, 2, 3)))
    assert np.array_equal(d2.grad.cpu().numpy(), want_dg)
    assert np.array_equal(f2.grad.cpu().numpy(), want_fg)


def test_trt_shim_eager_forward():
    depth, feat_nhwc, ranks, shape = _synthetic_case(
        synthetic.GRID_BEVDET, (256, 704), 1, 64)
    rb, rd, rf, st, ln = ranks
    out = TRTBEVPoolv2.apply(dev(depth[0]), dev(feat_nhwc[0]), dev(rd), dev(rf),
                             dev(rb), dev(st), dev(ln), 128, 128)
    want = _oracle_out(depth, feat_nhwc, ranks, shape)[:, 0]
    assert out.shape == (1, 128, 128, 64)
    assert np.array_equal(out.cpu().numpy(), want)


def test_size_independent_properties_full_size():
    """At BASELINE size: linearity in depth, and mass conservation
    sum(out) == sum_p depth*sum_c feat (fp64 reference)."""
    depth, feat_nhwc, ranks, shape = _synthetic_case(
        synthetic.GRID_S2, (256, 704), 6, 80)
    rb, rd, rf, st, ln = ranks
    a = _run_fused(depth, feat_nhwc, ranks, shape, _lib.LAYOUT_BCZYX)
    b = _run_fused(depth * np.float32(2.0), feat_nhwc, ranks, shape,
                   _lib.LAYOUT_BCZYX)
    assert np.array_equal(b, a * np.float32(2.0))     # exact: power of two
    mass = (depth.reshape(-1)[rd].astype(np.float64) *
            feat_nhwc.reshape(-1, shape[-1])[rf].astype(np.float64).sum(1)).sum()
    assert a.astype(np.float64).sum() == pytest.approx(mass, rel=1e-6, abs=1e-3)
    # untouched voxels are exactly zero
    occ = np.zeros(np.prod(shape[:4]), bool)
    occ[rb] = True
    assert not a.reshape(shape[0], shape[4], -1)[0][:, ~occ[:a[0, 0].size]].any()


def test_cpu_tensors_raise():
    with pytest.raises(_lib.VeonHipError):
        ext.bev_pool_v2_forward(*(torch.zeros(1),) * 3,
                                *(torch.zeros(1, dtype=torch.int32),) * 5)


@pytest.mark.parametrize('C', [8, 80])
def test_very_long_intervals_and_dense_tiles(C):
    """Edge cases of the fused kernels' staging logic: one interval longer than
    the staged window (3000 > 1024 points), a tile whose points need several
    windows, a tile with all 64 voxels occupied (two 32-voxel halves when the
    compact LDS tile has 32 columns) and intervals straddling window ends."""
    rng = np.random.default_rng(7 + C)
    B, Z, Y, X = 1, 2, 8, 32            # 512 voxels = 8 tiles of 64
    nvox = B * Z * Y * X
    lens = np.zeros(nvox, np.int64)
    lens[5] = 3000                       # > window
    lens[64:128] = rng.integers(1, 40, 64)    # dense tile, ~1300 points
    lens[130] = 1024                     # exactly one window
    lens[131] = 1025
    lens[200:260:3] = rng.integers(1, 5, 20)  # sparse, crosses a tile boundary
    lens[511] = 7                        # last voxel
    rb = np.repeat(np.arange(nvox), lens).astype(np.int32)
    n = len(rb)
    n_depth, n_feat = 997, 211
    rd = rng.integers(0, n_depth, n).astype(np.int32)
    rf = rng.integers(0, n_feat, n).astype(np.int32)
    st, ln = helpers.bp_intervals(rb)
    depth = rng.random((1, 1, n_depth, 1, 1), dtype=np.float32)
    feat = rng.standard_normal((1, 1, n_feat, 1, C)).astype(np.float32)
    shape = (B, Z, Y, X, C)
    ranks = (rb, rd, rf, st, ln)
    want = _oracle_out(depth, feat, ranks, shape)
    assert np.array_equal(_run_scatter(depth, feat, ranks, shape), want)
    assert np.array_equal(_run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BZYXC), want)
    assert np.array_equal(_run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BCZYX, True),
                          want.transpose(0, 4, 1, 2, 3))
    # fused max-pool on the same case
    got = bp.bev_pool_v2_maxpool(dev(depth), dev(feat), dev(rd), dev(rf), dev(rb),
                                 shape, dev(st), dev(ln), (2, 2, 2))
    full = want.transpose(0, 4, 1, 2, 3)
    assert np.array_equal(got.cpu().numpy(), c_oracle.maxpool3d(full, (2, 2, 2)))


def _random_case(rng):
    B = int(rng.integers(1, 4))
    Z, Y, X = int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.integers(1, 70))
    C = int(rng.choice([1, 2, 4, 5, 8, 12, 32, 80, 132, 256]))
    nvox = B * Z * Y * X
    occ_p = rng.choice([0.02, 0.3, 0.9])
    heavy = rng.random() < 0.5
    lens = (rng.random(nvox) < occ_p) * rng.integers(1, 6, nvox)
    if heavy:
        hot = rng.integers(0, nvox, size=min(3, nvox))
        lens[hot] = rng.integers(100, 1500, size=len(hot))
    rb = np.repeat(np.arange(nvox), lens).astype(np.int32)
    n = len(rb)
    n_depth, n_feat = int(rng.integers(1, 400)), int(rng.integers(1, 90))
    rd = rng.integers(0, n_depth, n).astype(np.int32)
    rf = rng.integers(0, n_feat, n).astype(np.int32)
    st, ln = helpers.bp_intervals(rb) if n else (np.zeros(0, np.int32),) * 2
    depth = rng.random((1, 1, n_depth, 1, 1), dtype=np.float32)
    feat = rng.standard_normal((1, 1, n_feat, 1, C)).astype(np.float32)
    return depth, feat, (rb, rd, rf, st, ln), (B, Z, Y, X, C)


@pytest.mark.parametrize('seed', range(24))
def test_random_structures_bit_exact(seed):
    """Random grids (ragged tiles, odd channel counts, empty / dense / very long
    intervals, several batch elements): every forward kernel equals the oracle
    bit for bit, and the fused max-pool equals max-pooling that volume."""
    rng = np.random.default_rng(1000 + seed)
    depth, feat, ranks, shape = _random_case(rng)
    want = _oracle_out(depth, feat, ranks, shape)
    if len(ranks[0]):
        assert np.array_equal(_run_scatter(depth, feat, ranks, shape), want)
    assert np.array_equal(_run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BZYXC), want)
    cf = _run_fused(depth, feat, ranks, shape, _lib.LAYOUT_BCZYX, bool(seed & 1))
    assert np.array_equal(cf, want.transpose(0, 4, 1, 2, 3))
    B, Z, Y, X, C = shape
    ds = tuple(int(d) for d, n in ((2, Z), (2, Y), (2, X)))
    ds = tuple(d if n % d == 0 else 1 for d, n in zip(ds, (Z, Y, X)))
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    got = bp.bev_pool_v2_maxpool(dev(depth), dev(feat), rd, rf, rb, shape, st, ln, ds)
    assert np.array_equal(got.cpu().numpy(), c_oracle.maxpool3d(cf, ds))


@pytest.mark.parametrize('seed', range(6))
def test_random_structures_backward_bit_exact(seed):
    rng = np.random.default_rng(2000 + seed)
    depth, feat, ranks, shape = _random_case(rng)
    rb, rd, rf, st, ln = ranks
    if len(rb) == 0:
        pytest.skip('empty case')
    # every frustum point has its own depth element (ranks_depth is a
    # permutation in real data); depth_grad is a STORE per point
    # (bev_pool_cuda.cu:104), so duplicates would race in the reference too
    n = len(rb)
    depth = rng.random((1, 1, n + 7, 1, 1), dtype=np.float32)
    rd = rng.permutation(n + 7)[:n].astype(np.int32)
    og = rng.standard_normal(shape).astype(np.float32)
    want_dg, want_fg = helpers.oracle_backward(og, depth, feat, rd, rf, rb)
    d = dev(depth).requires_grad_()
    f = dev(feat).requires_grad_()
    out = bev_pool_v2(d, f, dev(rd), dev(rf), dev(rb), shape, dev(st), dev(ln))
    out.backward(dev(og.transpose(0, 4, 1, 2, 3)))
    assert np.array_equal(d.grad.cpu().numpy(), want_dg)
    assert np.array_equal(f.grad.cpu().numpy(), want_fg)


# ---------------------------------------------------------------------------
# half-precision feature rows (veon_bev_pool_v2_fwd_fused_ex / _maxpool_ex)
# ---------------------------------------------------------------------------
def _half_feat(feat, dtype):
    """(device half tensor, the same values widened to fp32 on the host): the
    reference's `feat.float()` (bev_pool.py:21) is exact, so the oracle on the
    widened rows is the bit-exact expectation."""
    h = torch.from_numpy(feat).to(dtype)
    return h.to(DEV), h.float().numpy()


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('seed', range(8))
def test_half_feat_random_structures_bit_exact(seed, dtype):
    rng = np.random.default_rng(3000 + seed)
    depth, feat, ranks, shape = _random_case(rng)
    feat_h, feat_w = _half_feat(feat, dtype)
    want = _oracle_out(depth, feat_w, ranks, shape)
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    B, Z, Y, X, C = shape
    before = dict(_lib.CALLS)
    for layout, expect in ((_lib.LAYOUT_BZYXC, want),
                           (_lib.LAYOUT_BCZYX, want.transpose(0, 4, 1, 2, 3))):
        out = bp._fused_forward(dev(depth), feat_h, rd, rf, rb, st, ln, shape, layout)
        assert out.dtype == torch.float32
        assert np.array_equal(out.cpu().numpy(), expect)
    # channels-last always runs the slab kernel; channels-first the row kernel
    # from bp.ROWS_MIN_C channels on
    assert _lib.CALLS['veon_bev_pool_v2_fwd_fused_ex'] - \
        before.get('veon_bev_pool_v2_fwd_fused_ex', 0) == (1 if bp._rows_ok(C) else 2)
    # through the op: inference keeps the half rows, result still fp32
    with torch.no_grad():
        got = bev_pool_v2(dev(depth), feat_h, rd, rf, rb, shape, st, ln)
    assert got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy(), want.transpose(0, 4, 1, 2, 3))
    ds = tuple(2 if n % 2 == 0 else 1 for n in (Z, Y, X))
    mp = bp.bev_pool_v2_maxpool(dev(depth), feat_h, rd, rf, rb, shape, st, ln, ds)
    assert np.array_equal(mp.cpu().numpy(),
                          c_oracle.maxpool3d(want.transpose(0, 4, 1, 2, 3), ds))


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_half_feat_veon_shape_bit_exact(dtype):
    """SV (6 cams 512x1408, D=88, C=256): fused kernel and fused max-pool on
    half rows equal the oracle on the widened rows."""
    depth, feat, ranks, shape = _synthetic_case(synthetic.GRID_VEON, (512, 1408), 6, 256)
    feat_h, feat_w = _half_feat(feat, dtype)
    want = _oracle_out(depth, feat_w, ranks, shape).transpose(0, 4, 1, 2, 3)
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    bp.mark_sorted(st, int(ranks[0][0]), int(ranks[0][-1]))
    with torch.no_grad():
        got = bev_pool_v2(dev(depth), feat_h, rd, rf, rb, shape, st, ln)
    assert np.array_equal(got.cpu().numpy(), want)
    mp = bp.bev_pool_v2_maxpool(dev(depth), feat_h, rd, rf, rb, shape, st, ln, (2, 2, 2))
    assert np.array_equal(mp.cpu().numpy(), c_oracle.maxpool3d(want, (2, 2, 2)))


def test_half_feat_with_grad_takes_fp32_autograd_path():
    """Training: the reference widens and differentiates in fp32
    (bev_pool.py:21, 43-83); half rows that need a gradient do the same."""
    depth, feat, ranks, shape = _case_from_golden('lss_small')
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    f = torch.from_numpy(feat).to(torch.float16).to(DEV).requires_grad_(True)
    d = dev(depth).requires_grad_(True)
    out = bev_pool_v2(d, f, rd, rf, rb, shape, st, ln)
    out.sum().backward()
    assert f.grad is not None and f.grad.dtype == torch.float16
    assert d.grad is not None and torch.isfinite(d.grad).all()
    want = _oracle_out(depth, f.detach().float().cpu().numpy(), ranks, shape)
    assert np.array_equal(out.detach().cpu().numpy(), want.transpose(0, 4, 1, 2, 3))


def test_contiguous_volume_allocation_is_a_usable_tensor():
    """placement.contiguous_tensor: physically contiguous VRAM wrapped as a torch
    tensor (or None when the driver refuses); the pool writes into it like into
    any other buffer, and the block is released with the tensor."""
    import gc
    import torch
    from veon_amd import placement
    t = placement.contiguous_tensor((3, 5, 7), torch.float32, torch.device('cuda:0'))
    if t is None:
        pytest.skip('driver gave no contiguous block')
    assert t.is_cuda and t.shape == (3, 5, 7) and t.is_contiguous()
    t.fill_(2.5)
    assert float(t.sum()) == 2.5 * 105
    u = t.view(-1)[10:20]
    del t
    gc.collect()
    u += 1                      # the view keeps the block alive
    assert float(u.sum()) == 35.0
    del u
    gc.collect()
    torch.cuda.synchronize()


def test_fused_pool_into_padded_planes_equals_contiguous():
    """veon_bev_pool_v2_fwd_fused_strided: channel planes `plane_stride` floats
    apart hold exactly the contiguous (B,C,Z,Y,X) result; the padding between the
    planes is not touched; a stride below the plane size is refused."""
    import ctypes
    import torch
    from tools._inputs import lift_case
    from veon_amd import _lib, synthetic
    from veon_amd.ops.bev_pool_v2 import bev_pool as bp
    dev = 'cuda:0'
    C = 16
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    case = lift_case(grid, (64, 176), 2, C, dev)
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = case['gsize']
    vpb = X * Y * Z
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))
    plan = bp.build_plan(rb, st, 1, vpb)
    L = _lib.lib()
    s = _lib.stream_ptr(torch.device(dev))

    def run(out, stride):
        return L.veon_bev_pool_v2_fwd_fused_strided(
            C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(feat), _lib.FEAT_F32,
            _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
            _lib.ptr(plan), _lib.ptr(out), stride, s)
    want = torch.empty(C, vpb, device=dev)
    assert run(want, vpb) == 0
    pad = 192
    got = torch.full((C, vpb + pad), -7.0, device=dev)
    assert run(got, vpb + pad) == 0
    torch.cuda.synchronize()
    assert float(want.abs().sum()) > 0
    assert torch.equal(got[:, :vpb], want)
    assert bool((got[:, vpb:] == -7.0).all())
    assert run(got, vpb - 1) != 0


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16', 'float16'])
@pytest.mark.parametrize('B,N,C,H,W', [(1, 6, 80, 16, 44), (2, 3, 256, 5, 7), (1, 1, 17, 3, 65)])
def test_feature_rows_transpose_kernel_equals_contiguous(dtype, B, N, C, H, W):
    """bev_pool._rows: the permuted (B,N,C,H,W) -> (B,N,H,W,C) view made contiguous
    by the LDS-tiled kernel is bit-identical to torch's .contiguous(); other
    stride patterns take the torch path."""
    import torch
    from veon_amd import _lib
    from veon_amd.ops.bev_pool_v2 import bev_pool as bp
    dt = getattr(torch, dtype)
    x = torch.randn(B, N, C, H, W, device='cuda:0').to(dt)
    view = x.permute(0, 1, 3, 4, 2)
    before = _lib.CALLS.get('veon_feat_nchw_to_nhwc', 0)
    got = bp._rows(view)
    assert _lib.CALLS.get('veon_feat_nchw_to_nhwc', 0) == before + 1
    assert got.is_contiguous() and torch.equal(got, view.contiguous())
    other = x.permute(0, 1, 4, 3, 2)            # not the lift's pattern
    assert torch.equal(bp._rows(other), other.contiguous())
    assert _lib.CALLS.get('veon_feat_nchw_to_nhwc', 0) == before + 1
    assert bp._rows(got) is got


def test_backward_bit_exact_at_full_s2_size():
    """Row a10 at BASELINE configs[1]'s full size (6 cams 256x704, D = 59, C = 80,
    200x200x16 voxels, 109 k kept points): depth.grad and feat.grad of the HIP backward
    (bev_pool_cuda.cu:67-121) equal the C oracle bit for bit, through the reference-layout
    op and through the channels-first drop-in (out_grad arrives as (B,C,Z,Y,X))."""
    from tools._inputs import lift_case
    cs = lift_case(synthetic.GRID_S2, (256, 704), 6, 80, 'cuda:0')
    X, Y, Z = cs['gsize']
    shape = (1, Z, Y, X, 80)
    depth, feat = cs['depth'], cs['feat_nhwc']
    rb, rd, rf, st, ln = (cs[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    g = torch.Generator().manual_seed(3)
    og = torch.randn(shape, generator=g)
    want_dg, want_fg = helpers.oracle_backward(
        og.numpy(), depth.cpu().numpy(), feat.cpu().numpy(), rd.cpu().numpy(),
        rf.cpu().numpy(), rb.cpu().numpy())
    d = depth.clone().requires_grad_()
    f = feat.clone().requires_grad_()
    out = QuickCumsumCuda.apply(d, f, rd, rf, rb, shape, st, ln)
    out.backward(og.cuda())
    assert np.array_equal(d.grad.cpu().numpy(), want_dg)
    assert np.array_equal(f.grad.cpu().numpy(), want_fg)
    d2 = depth.clone().requires_grad_()
    f2 = feat.clone().requires_grad_()
    out2 = bev_pool_v2(d2, f2, rd, rf, rb, shape, st, ln)
    assert tuple(out2.shape) == (1, 80, Z, Y, X)
    out2.backward(og.permute(0, 4, 1, 2, 3).contiguous().cuda())
    assert np.array_equal(d2.grad.cpu().numpy(), want_dg)
    assert np.array_equal(f2.grad.cpu().numpy(), want_fg)
    # size-independent property: the backward is linear in out_grad
    d3 = depth.clone().requires_grad_()
    f3 = feat.clone().requires_grad_()
    QuickCumsumCuda.apply(d3, f3, rd, rf, rb, shape, st, ln).backward(2.0 * og.cuda())
    assert torch.equal(d3.grad, 2.0 * d.grad) and torch.equal(f3.grad, 2.0 * f.grad)

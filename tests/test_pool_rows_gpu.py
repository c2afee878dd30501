"""GPU parity of the "row" pool kernels (csrc/bev_pool_rows.hip) against the CPU
oracle: dense voxel table, fused (B,C,Z,Y,X) forward, fused pool + 2x2x2 max-pool
(fp32 planar and padded bf16 channels-last outputs).  Sums are serial fmaf chains
in storage order on both sides (bev_pool_cuda.cu:38-43), so everything is
BIT-EXACT.  All calls go through the C ABI of libveon_hip.so.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from tests import helpers
from veon_amd import _lib, conv3d_ops, synthetic
from veon_amd.ops.bev_pool_v2 import bev_pool as bp
from veon_amd.ops.bev_pool_v2.bev_pool import bev_pool_v2

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return helpers.t(a, DEV)


def _oracle_cf(depth, feat_nhwc, ranks, shape):
    rb, rd, rf, st, ln = ranks
    B, Z, Y, X, C = shape
    out = c_oracle.bev_pool_v2_fwd(depth, feat_nhwc, rd, rf, rb, st, ln, B * Z * Y * X)
    return out.reshape(B, Z, Y, X, C).transpose(0, 4, 1, 2, 3)


def _case(rng, C, dims=None, heavy=None):
    B = int(rng.integers(1, 3))
    if dims is None:
        Z, Y, X = 2 * int(rng.integers(1, 3)), 2 * int(rng.integers(1, 5)), \
            2 * int(rng.integers(1, 40))
    else:
        Z, Y, X = dims
    nvox = B * Z * Y * X
    occ_p = rng.choice([0.02, 0.3, 0.9])
    heavy = (rng.random() < 0.5) if heavy is None else heavy
    lens = (rng.random(nvox) < occ_p) * rng.integers(1, 6, nvox)
    if heavy:
        hot = rng.integers(0, nvox, size=min(3, nvox))
        lens[hot] = rng.integers(60, 700, size=len(hot))
    rb = np.repeat(np.arange(nvox), lens).astype(np.int32)
    n = len(rb)
    n_depth, n_feat = int(rng.integers(1, 400)), int(rng.integers(1, 90))
    rd = rng.integers(0, n_depth, n).astype(np.int32)
    rf = rng.integers(0, n_feat, n).astype(np.int32)
    st, ln = helpers.bp_intervals(rb) if n else (np.zeros(0, np.int32),) * 2
    depth = rng.random((1, 1, n_depth, 1, 1), dtype=np.float32)
    feat = rng.standard_normal((1, 1, n_feat, 1, C)).astype(np.float32)
    return depth, feat, (rb, rd, rf, st, ln), (B, Z, Y, X, C)


def _feat_as(feat, dtype):
    """(device rows in `dtype`, the same values widened to fp32 on the host)."""
    h = torch.from_numpy(feat).to(dtype)
    return h.to(DEV), h.float().numpy()


def test_voxel_table_equals_searchsorted():
    rng = np.random.default_rng(5)
    for _ in range(4):
        depth, feat, ranks, shape = _case(rng, 8)
        rb, rd, rf, st, ln = ranks
        B, Z, Y, X, C = shape
        vpb = Z * Y * X
        vs = bp.build_voxel_table(dev(rb), dev(st), B, vpb, attach=False)
        want = np.searchsorted(rb, np.arange(B * vpb + 1), side='left').astype(np.int32)
        assert np.array_equal(vs.cpu().numpy(), want)
    # empty input: every voxel starts at 0
    vs = bp.build_voxel_table(dev(np.zeros(0, np.int32)), dev(np.zeros(0, np.int32)),
                              1, 24, attach=False)
    assert not vs.cpu().numpy().any() and vs.numel() == 25


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('C', [2, 8, 80, 128, 132, 256, 260, 512])
def test_rows_forward_random_structures_bit_exact(C, dtype):
    rng = np.random.default_rng(100 + C)
    for rep in range(3):
        depth, feat, ranks, shape = _case(rng, C)
        feat_d, feat_w = _feat_as(feat, dtype)
        want = _oracle_cf(depth, feat_w, ranks, shape)
        rb, rd, rf, st, ln = (dev(a) for a in ranks)
        B, Z, Y, X, _ = shape
        vs = bp.build_voxel_table(rb, st, B, Z * Y * X, attach=False)
        for variant in range(4):
            junk = torch.full((int(np.prod(shape)),), float('nan'), device=DEV)
            del junk
            got = bp.rows_forward(dev(depth), feat_d, rd, rf, vs, shape, variant=variant)
            assert np.array_equal(got.cpu().numpy(), want), (rep, variant)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [4, 64, 80, 256, 264, 512])
def test_rows_maxpool_random_structures_bit_exact(C, dtype):
    rng = np.random.default_rng(200 + C)
    for rep in range(3):
        depth, feat, ranks, shape = _case(rng, C)
        feat_d, feat_w = _feat_as(feat, dtype)
        full = _oracle_cf(depth, feat_w, ranks, shape)
        want = c_oracle.maxpool3d(np.ascontiguousarray(full), (2, 2, 2))
        rb, rd, rf, st, ln = (dev(a) for a in ranks)
        B, Z, Y, X, _ = shape
        vs = bp.build_voxel_table(rb, st, B, Z * Y * X, attach=False)
        got = bp.rows_maxpool(dev(depth), feat_d, rd, rf, vs, shape, (2, 2, 2))
        assert np.array_equal(got.cpu().numpy(), want), rep
        # padded bf16 channels-last output == bf16 rounding of the fp32 result
        vol = conv3d_ops.PaddedVolume(B, C, Z // 2, Y // 2, X // 2, DEV)
        vol.storage.fill_(7.0)          # interior must be overwritten everywhere
        bp.rows_maxpool(dev(depth), feat_d, rd, rf, vs, shape, (2, 2, 2), out_volume=vol)
        inner = vol.interior().permute(0, 4, 1, 2, 3).float().cpu().numpy()
        assert np.array_equal(inner, torch.from_numpy(want).bfloat16().float().numpy())


def test_negative_sums_against_empty_neighbours():
    """A pooled voxel with fewer than 8 occupied inputs also competes against the
    zeros of its empty inputs; one with all 8 occupied does not."""
    C = 4
    B, Z, Y, X = 1, 2, 2, 4
    # pooled voxel 0: all 8 inputs occupied, all sums negative -> negative max
    # pooled voxel 1: one occupied input with a negative sum -> 0
    vox = [z * Y * X + y * X + x for z in range(2) for y in range(2) for x in range(2)]
    vox += [2]
    rb = np.array(sorted(vox), np.int32)
    n = len(rb)
    rd = np.arange(n, dtype=np.int32)
    rf = np.zeros(n, np.int32)
    st, ln = helpers.bp_intervals(rb)
    depth = np.linspace(1, 2, n, dtype=np.float32).reshape(1, 1, n, 1, 1)
    feat = -np.ones((1, 1, 1, 1, C), np.float32)
    shape = (B, Z, Y, X, C)
    full = _oracle_cf(depth, feat, (rb, rd, rf, st, ln), shape)
    want = c_oracle.maxpool3d(np.ascontiguousarray(full), (2, 2, 2))
    assert want[0, 0, 0, 0, 0] < 0 and want[0, 0, 0, 0, 1] == 0
    vs = bp.build_voxel_table(dev(rb), dev(st), B, Z * Y * X, attach=False)
    got = bp.rows_maxpool(dev(depth), dev(feat), dev(rd), dev(rf), vs, shape, (2, 2, 2))
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_veon_shape_rows_bit_exact(dtype):
    """SV: 6 cams 512x1408, D=88, C=256 into 200x200x16 -- the fused forward
    (655 MB volume), the fused max-pool in both output forms, and the op-level
    dispatch (bev_pool_v2 / bev_pool_v2_maxpool take the row kernels here)."""
    ranks, coor, rig, fr, gsize = helpers.oracle_ranks(synthetic.GRID_VEON, (512, 1408), 6)
    D, C = fr.shape[0], 256
    depth, feat = synthetic.make_depth_feat(1, 6, D, C, 32, 88, 0)
    feat_nhwc = feat.permute(0, 1, 3, 4, 2).contiguous().numpy()
    depth = depth.numpy()
    shape = (1, int(gsize[2]), int(gsize[1]), int(gsize[0]), C)
    feat_d, feat_w = _feat_as(feat_nhwc, dtype)
    want = np.ascontiguousarray(_oracle_cf(depth, feat_w, ranks, shape))
    rb, rd, rf, st, ln = (dev(a) for a in ranks)
    before = dict(_lib.CALLS)
    with torch.no_grad():
        got = bev_pool_v2(dev(depth), feat_d, rd, rf, rb, shape, st, ln)
    assert _lib.CALLS['veon_bev_pool_v2_fwd_rows'] == \
        before.get('veon_bev_pool_v2_fwd_rows', 0) + 1
    assert np.array_equal(got.cpu().numpy(), want)
    del got
    want_mp = c_oracle.maxpool3d(want, (2, 2, 2))
    mp = bp.bev_pool_v2_maxpool(dev(depth), feat_d, rd, rf, rb, shape, st, ln, (2, 2, 2))
    assert _lib.CALLS['veon_bev_pool_v2_fwd_rows_maxpool_ordered'] == \
        before.get('veon_bev_pool_v2_fwd_rows_maxpool_ordered', 0) + 1
    assert np.array_equal(mp.cpu().numpy(), want_mp)
    vol = conv3d_ops.PaddedVolume(1, C, 8, 100, 100, DEV)
    bp.bev_pool_v2_maxpool(dev(depth), feat_d, rd, rf, rb, shape, st, ln, (2, 2, 2),
                           out_volume=vol)
    inner = vol.interior().permute(0, 4, 1, 2, 3).float().cpu().numpy()
    assert np.array_equal(inner, torch.from_numpy(want_mp).bfloat16().float().numpy())
    # the halo stayed zero
    assert float(vol.rows.float().abs().sum()) == pytest.approx(
        float(np.abs(inner.astype(np.float64)).sum()), rel=1e-3)


def test_gappy_intervals_fall_back_to_reference_semantics():
    """sorted keys whose intervals do NOT tile the point arrays
    (lengths shorter than the start differences, points outside any interval).
    The reference kernel honours interval_lengths (bev_pool_cuda.cu:35); the
    fused kernels derive lengths from the next start, so the op must not take
    them -- and must still equal the oracle."""
    rng = np.random.default_rng(11)
    for C in (8, 256):
        depth, feat, ranks, shape = _case(rng, C, dims=(2, 4, 20), heavy=False)
        rb, rd, rf, st, ln = ranks
        ln2 = np.maximum(1, ln - (rng.random(len(ln)) < 0.5)).astype(np.int32)
        assert (ln2 != ln).any()
        ranks2 = (rb, rd, rf, st, ln2)
        want = _oracle_cf(depth, feat, ranks2, shape)
        assert not np.array_equal(want, _oracle_cf(depth, feat, ranks, shape))
        got = bev_pool_v2(dev(depth), dev(feat), dev(rd), dev(rf), dev(rb), shape,
                          dev(st), dev(ln2))
        assert np.array_equal(got.cpu().numpy(), want)


def test_in_place_update_voids_cached_index():
    """the sorted tag / plan / voxel table are cached on the
    interval_starts tensor object; rewriting the tensors in place (buffer reuse)
    must recompute them instead of trusting stale structures."""
    rng = np.random.default_rng(12)
    C = 256
    d1, f1, r1, shape = _case(rng, C, dims=(2, 4, 30), heavy=False)
    rb, rd, rf, st, ln = (dev(a) for a in r1)
    got = bev_pool_v2(dev(d1), dev(f1), rd, rf, rb, shape, st, ln)
    assert np.array_equal(got.cpu().numpy(), _oracle_cf(d1, f1, r1, shape))
    # same buffers, new contents of the same sizes: shift every rank by one voxel
    rb2 = (r1[0] + 1).clip(max=shape[0] * shape[1] * shape[2] * shape[3] - 1)
    st2, ln2 = helpers.bp_intervals(rb2)
    if len(st2) != len(r1[3]):
        pytest.skip('shifted case changed the interval count')
    rb.copy_(dev(rb2))
    st.copy_(dev(st2))
    ln.copy_(dev(ln2))
    r2 = (rb2, r1[1], r1[2], st2, ln2)
    got = bev_pool_v2(dev(d1), dev(f1), rd, rf, rb, shape, st, ln)
    assert np.array_equal(got.cpu().numpy(), _oracle_cf(d1, f1, r2, shape))

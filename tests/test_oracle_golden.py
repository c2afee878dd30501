"""Pin the CPU oracle against the reference's own vectors (no GPU).

* pool fwd/bwd  <- the reference's known-answer test (bev_pool.py:145-176)
* geometry / prepare / depth prep / max-pool <- tests/golden/*.npz, generated
  by running the reference's Python (oracle/tools/gen_golden.py)
* full BASELINE shapes <- hashes of the reference prepare on oracle coordinates
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle, lss_torch
from tests.conftest import load_golden
from tests import helpers
from veon_amd import synthetic

SMALL = ['lss_small', 'lss_small_b2', 'lss_mid']


def test_kat_forward_sum():
    g = load_golden('kat_bev_pool_v2')
    out = c_oracle.bev_pool_v2_fwd(
        g['depth'], g['feat'], g['ranks_depth'], g['ranks_feat'],
        g['ranks_bev'], g['interval_starts'], g['interval_lengths'], 8)
    # voxel 0 = .3+.7, voxel 1 = .4+.8, two channels each
    assert np.float32(out.sum()) == g['expect_sum']
    assert np.allclose(out[:2], [[1.0, 1.0], [1.2, 1.2]])
    assert not out[2:].any()


def test_kat_backward():
    g = load_golden('kat_bev_pool_v2')
    og = np.ones((1, 1, 2, 2, 2), np.float32)  # d(sum)/d(out)
    dg, fg = helpers.oracle_backward(og, g['depth'], g['feat'],
                                     g['ranks_depth'], g['ranks_feat'],
                                     g['ranks_bev'])
    assert np.allclose(dg, g['expect_depth_grad'])
    assert np.allclose(fg, g['expect_feat_grad'])


@pytest.mark.parametrize('name', SMALL)
def test_geometry_matches_reference_bitwise(name):
    g = load_golden(name)
    pri, comb, trans = lss_torch.camera_matrices(
        torch.from_numpy(g['sensor2ego']), torch.from_numpy(g['intrins']),
        torch.from_numpy(g['post_rots']))
    coor = c_oracle.get_lidar_coor(g['frustum'], pri.numpy(), g['post_trans'],
                                   comb.numpy(), trans.numpy(), g['bda'])
    assert coor.shape == g['coor'].shape
    assert np.array_equal(coor, g['coor'])
    # the torch restatement agrees too
    fr = lss_torch.make_frustum(list(g['grid_depth']), tuple(g['input_size']), 16)
    assert np.array_equal(fr.numpy(), g['frustum'])
    c2 = lss_torch.lidar_coor(fr, torch.from_numpy(g['sensor2ego']),
                              torch.from_numpy(g['intrins']),
                              torch.from_numpy(g['post_rots']),
                              torch.from_numpy(g['post_trans']),
                              torch.from_numpy(g['bda']))
    assert np.array_equal(c2.numpy(), g['coor'])


@pytest.mark.parametrize('name', SMALL)
def test_prepare_matches_reference_exactly(name):
    g = load_golden(name)
    rb, rd, rf, st, ln = c_oracle.voxel_prepare(
        g['coor'], g['grid_lower_bound'], g['grid_interval'], g['grid_size'])
    assert np.array_equal(rb, g['ranks_bev'])
    assert np.array_equal(st, g['interval_starts'])
    assert np.array_equal(ln, g['interval_lengths'])
    # canonical (stable) order inside each interval
    assert np.array_equal(rd, g['ranks_depth'])
    assert np.array_equal(rf, g['ranks_feat'])
    # the raw (unstable-argsort) reference output is the same per interval
    for s, l in zip(st[:200], ln[:200]):
        assert sorted(g['ranks_depth_raw'][s:s + l]) == list(rd[s:s + l])
    # torch port
    lower, interval, gsize = (torch.from_numpy(g[k]) for k in
                              ('grid_lower_bound', 'grid_interval', 'grid_size'))
    tr = lss_torch.voxel_prepare(torch.from_numpy(g['coor']), lower, interval, gsize)
    for a, b in zip(tr, (rb, rd, rf, st, ln)):
        assert np.array_equal(a.numpy(), b)


@pytest.mark.parametrize('name', SMALL)
def test_depth_prep_matches_reference(name):
    g = load_golden(name)
    ds = c_oracle.downsample_depth(g['metric_depth'], 8)
    assert np.array_equal(ds, g['ds_depth'])
    lo, _, step = g['grid_depth']
    th = c_oracle.two_hot_depth(g['ds_depth'], int(g['D']), lo, step, 4.0)
    assert th.shape == g['two_hot'].shape
    # libm expf vs torch's vectorised exp: tolerance, not bitwise
    np.testing.assert_allclose(th, g['two_hot'], rtol=2e-6, atol=1e-9)
    th2 = lss_torch.two_hot_depth(torch.from_numpy(g['ds_depth']), int(g['D']),
                                  lo, step)
    np.testing.assert_allclose(th2.numpy(), g['two_hot'], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize('name', SMALL)
def test_pool_wiring_matches_reference(name):
    """pooled / forward_out were produced by the reference's Python with this
    repo's torch pool plugged in (see gen_golden.py): they pin permute,
    collapse_z=False and the 2x2x2 max-pool.  The C oracle's serial fmaf sums
    differ from index_add_ only by summation order."""
    g = load_golden(name)
    B, C = g['feat'].shape[0], g['feat'].shape[2]
    X, Y, Z = (int(v) for v in g['grid_size'])
    feat_nhwc = np.ascontiguousarray(g['feat'].transpose(0, 1, 3, 4, 2))
    out = c_oracle.bev_pool_v2_fwd(
        g['two_hot'], feat_nhwc, g['ranks_depth'], g['ranks_feat'],
        g['ranks_bev'], g['interval_starts'], g['interval_lengths'],
        B * Z * Y * X)
    vol = c_oracle.permute_to_bczyx(out.reshape(B, Z, Y, X, C))
    np.testing.assert_allclose(vol, g['pooled'], rtol=1e-5, atol=1e-6)
    mp = c_oracle.maxpool3d(vol, tuple(int(v) for v in g['ds_feat']))
    np.testing.assert_allclose(mp, g['forward_out'], rtol=1e-5, atol=1e-6)
    mp_t = lss_torch.maxpool(torch.from_numpy(g['pooled']),
                             tuple(int(v) for v in g['ds_feat']))
    assert np.array_equal(mp_t.numpy(), g['forward_out'])


@pytest.mark.parametrize('tag', ['S1', 'S2', 'SV'])
def test_full_shapes_hashes(tag, full_cases):
    e = full_cases[tag]
    ranks, coor, rig, fr, gsize = helpers.oracle_ranks(
        e['grid_config'], tuple(e['input_size']), e['n_cams'])
    assert helpers.sha(coor) == e['sha_oracle_coor']
    # the reference's own coordinates were bit-identical when the fixture was made
    assert e['ref_vs_oracle_coor_n_diff'] == 0
    sub = np.load('tests/golden/coor_sub_%s.npy' % tag)
    assert np.array_equal(coor.reshape(-1, 3)[::997], sub)
    rb, rd, rf, st, ln = ranks
    assert len(rb) == e['P_kept'] and len(st) == e['n_intervals']
    assert int(ln.max()) == e['max_interval']
    assert helpers.sha(rb) == e['sha_ranks_bev']
    assert helpers.sha(rd) == e['sha_ranks_depth']
    assert helpers.sha(rf) == e['sha_ranks_feat']
    assert helpers.sha(st) == e['sha_interval_starts']
    assert helpers.sha(ln) == e['sha_interval_lengths']


def test_torch_port_pool_matches_fixture():
    g = load_golden('lss_mid')
    B, C = g['feat'].shape[0], g['feat'].shape[2]
    X, Y, Z = (int(v) for v in g['grid_size'])
    vol = lss_torch.pool(torch.from_numpy(g['two_hot']),
                         torch.from_numpy(g['feat']).permute(0, 1, 3, 4, 2),
                         torch.from_numpy(g['ranks_depth']),
                         torch.from_numpy(g['ranks_feat']),
                         torch.from_numpy(g['ranks_bev']), (B, Z, Y, X, C))
    # the fixture summed in the reference's (unstable) point order
    np.testing.assert_allclose(vol.numpy(), g['pooled'], rtol=1e-5, atol=1e-6)


def test_truncation_keeps_points_in_minus_one_zero():
    """coor.long() truncates toward zero, so coordinates in (-1, 0) voxel units
    land in voxel 0 and are kept (SURVEY 7 'hard parts')."""
    lower = np.array([0, 0, 0], np.float32)
    interval = np.array([1, 1, 1], np.float32)
    gsize = np.array([4, 4, 2], np.float32)
    coor = np.array([[-0.5, 0.2, 0.1], [-1.0, 0.2, 0.1], [3.99, 3.2, 1.9],
                     [4.0, 0, 0]], np.float32).reshape(1, 1, 4, 1, 1, 3)
    rb, rd, rf, st, ln = c_oracle.voxel_prepare(coor, lower, interval, gsize)
    assert list(rd) == [0, 2] and list(rb) == [0, 16 + 12 + 3]

"""the two-hot lift BY CONSTRUCTION.

``get_two_hot_depth`` (view_transformer_raw.py:406-429) puts distinct weights only on
the window of bins whose logit is not clamped at -16 and one tail value on all others,
so the lift takes the metric depth map as per-pixel windows + a compact weight table
(``depth_ops.TwoHotWindows``); the (B,6,D,Hf,Wf) tensor is never written and points
whose weight is below ``eps`` never enter the sort.

Checked here:
* CPU: the compact form expands to the C oracle's dense two-hot tensor (rtol 1e-5:
  ``exp`` implementations), kept-set == {w >= eps};
* GPU, against the C ORACLE on the SAME thresholded point set (the oracle's own
  prepare on the full frustum, filtered by the kept mask, its serial fmaf pool with
  the device's compact weights scattered to a dense tensor): ranks and the max-pooled
  volume must agree BIT FOR BIT;
* GPU, against the FULL oracle lift (every frustum point, the oracle's own two-hot
  weights): within eps * pooled |feat| of the dropped points + the stated expf
  tolerance;
* eps = 0: the same bits as the dense HIP lift fed the dense two-hot tensor;
* the window set against the oracle's own threshold, up to borderline weights."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, lss_torch
from veon_amd import depth_ops, synthetic
from veon_amd.models import build_neck

GRID = {'x': [-40, 40, 1.6], 'y': [-40, 40, 1.6], 'z': [-1, 5.4, 0.8],
        'depth': [1.0, 45.0, 0.5]}
SIZE, CAMS = (128, 352), 6


def _metric(seed, far_fraction=0.25):
    """Metric depth at (H/2, W/2) like the VEON path: mostly inside the depth range,
    some pixels beyond it (uniform two-hot distribution: the tail is kept), some
    zeros (block-min treats them as missing), one all-zero block (-> 1e5)."""
    g = torch.Generator().manual_seed(seed)
    h, w = SIZE[0] // 2, SIZE[1] // 2
    d = 1.5 + 42.0 * torch.rand(1, CAMS, h, w, generator=g)
    far = torch.rand(1, CAMS, h // 8, w // 8, generator=g) < far_fraction
    far = far.repeat_interleave(8, 2).repeat_interleave(8, 3)
    d = torch.where(far, 46.0 + 30.0 * torch.rand(d.shape, generator=g), d)
    d = torch.where(torch.rand(d.shape, generator=g) < 0.05, torch.zeros(()), d)
    d[0, 0, :8, :8] = 0.0
    return d


def test_windows_expand_to_the_oracle_two_hot_tensor_cpu():
    lo, _, step = GRID['depth']
    D = 88
    ds = c_oracle.downsample_depth(_metric(0).numpy(), 8)
    ref = c_oracle.two_hot_depth(ds, D, lo, step, 4.0)
    for eps in (0.0, 1e-6, 1e-3):
        tw = depth_ops.two_hot_windows(_metric(0), D, lo, step, 4, eps, downsample=8)
        assert tw.K == depth_ops.two_hot_window_slots(D, step, 4) == 19
        np.testing.assert_allclose(tw.dense().numpy(), ref, rtol=1e-5, atol=0)
        kept, want = tw.kept().numpy(), ref >= eps
        border = np.abs(ref - eps) <= 2e-5 * max(eps, 1e-30)
        assert ((kept != want) & ~border).sum() == 0
        assert np.array_equal(tw.dense(thresholded=True).numpy() > 0, kept) or eps == 0.0


def _vt(dev, C, ds):
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=GRID, input_size=SIZE,
                         out_channels=C, collapse_z=False, ds_feat=ds)).to(dev).eval()
    vt.sync_free = True
    return vt


@pytest.mark.gpu
@pytest.mark.parametrize('eps', [1e-6, 1e-3, 0.0])
@pytest.mark.parametrize('C', [32, 256])
def test_two_hot_lift_against_the_oracle_on_the_same_thresholded_points(eps, C):
    dev = 'cuda:0'
    vt = _vt(dev, C, [2, 2, 2])
    hf, wf = SIZE[0] // 16, SIZE[1] // 16
    rig = synthetic.make_rig(1, CAMS, SIZE)
    inp = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(1, CAMS, C, hf, wf, generator=g)
    metric = _metric(2)
    with torch.no_grad():
        tw = vt.get_two_hot_windows(metric.to(dev), downsample=8, eps=eps)
        got = vt([feat.to(dev)] + inp, tw)
        # device ranks of the same call (the workspace the lift just used)
        ws = list(vt.__dict__['_veon_lift_workspaces'].values())[-1]
        kept_n, n_int = ws.counts.tolist()
        rb_d = ws.ranks_bev[:kept_n].cpu().numpy()
        rd_d = ws.ranks_depth[:kept_n].cpu().numpy()
        rf_d = ws.ranks_feat[:kept_n].cpu().numpy()
        coor = vt.get_lidar_coor(*inp).cpu().numpy()
    torch.cuda.synchronize()
    D = vt.D
    assert tw.K == 19 and tuple(tw.shape) == (1, CAMS, D, hf, wf)
    kept = tw.kept().cpu().numpy()                              # (1,N,D,h,w) bool
    dense_w = tw.dense(thresholded=True).cpu().numpy()          # device weights, dropped = 0
    # ---- oracle: its own prepare on the FULL frustum, filtered by the kept mask
    lower, interval, gsize = (t.numpy() for t in lss_torch.grid_infos(GRID))
    rb, rd, rf, _, _ = c_oracle.voxel_prepare(coor, lower, interval, gsize)
    m = kept.reshape(-1)[rd]
    rb, rd, rf = rb[m], rd[m], rf[m]
    assert kept_n == rb.size
    # ranks: same points, same order; ranks_depth is the compact index of the dense id
    wx = tw.win[..., 0].cpu().numpy().reshape(-1)
    k = (rd // (hf * wf)) % D
    j = k - (wx[rf] & 0xffff)
    compact = rf * tw.K + np.where((j >= 0) & (j < (wx[rf] >> 16)), 1 + j, 0)
    assert np.array_equal(rb_d, rb) and np.array_equal(rf_d, rf)
    assert np.array_equal(rd_d, compact)
    # weights the pool reads == the dense view of the compact table, bit for bit
    assert np.array_equal(tw.wts.cpu().numpy().reshape(-1)[rd_d], dense_w.reshape(-1)[rd])
    # ---- oracle pool (serial fmaf chain) on that point set, then permute + max-pool
    change = np.flatnonzero(np.diff(rb)) + 1
    st = np.concatenate(([0], change)).astype(np.int32)
    ln = np.diff(np.concatenate((st, [rb.size]))).astype(np.int32)
    X, Y, Z = (int(v) for v in gsize)
    f_np = feat.permute(0, 1, 3, 4, 2).contiguous().numpy()
    vol = c_oracle.bev_pool_v2_fwd(dense_w, f_np, rd, rf, rb, st, ln, Z * Y * X)
    want = c_oracle.maxpool3d(c_oracle.permute_to_bczyx(vol.reshape(1, Z, Y, X, C)), (2, 2, 2))
    assert np.array_equal(got.cpu().numpy(), want), 'two-hot lift differs from the oracle'
    # ---- against the FULL oracle lift (all points, the oracle's own weights)
    ds_o = c_oracle.downsample_depth(metric.numpy(), 8)
    w_o = c_oracle.two_hot_depth(ds_o, D, float(GRID['depth'][0]), float(GRID['depth'][2]), 4.0)
    rb_f, rd_f, rf_f, st_f, ln_f = c_oracle.voxel_prepare(coor, lower, interval, gsize)
    full = c_oracle.bev_pool_v2_fwd(w_o, f_np, rd_f, rf_f, rb_f, st_f, ln_f, Z * Y * X)
    # bound: eps * sum|feat| over the dropped points of the voxel + expf noise (1e-5 rel)
    drop_w = np.where(kept, 0.0, eps).astype(np.float32)
    absf = np.abs(f_np)
    bound = c_oracle.bev_pool_v2_fwd(drop_w, absf, rd_f, rf_f, rb_f, st_f, ln_f, Z * Y * X)
    mag = c_oracle.bev_pool_v2_fwd(w_o, absf, rd_f, rf_f, rb_f, st_f, ln_f, Z * Y * X)
    err = np.abs(vol - full)
    assert (err <= bound * 1.001 + 2e-5 * mag + 1e-7).all(), float((err - bound).max())
    if eps > 0:
        assert kept_n * 2 < rb_f.size      # far fewer points pooled
    print('eps %g: %d of %d points pooled, max |diff| to the full lift %.2e (max |V| %.2f)'
          % (eps, kept_n, rb_f.size, err.max(), np.abs(full).max()))


@pytest.mark.gpu
def test_eps_zero_equals_the_dense_lift_bit_for_bit():
    dev = 'cuda:0'
    for ds in ([2, 2, 2], [1, 1, 1]):
        vt = _vt(dev, 128, ds)
        hf, wf = SIZE[0] // 16, SIZE[1] // 16
        inp = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, CAMS, SIZE))]
        feat = torch.randn(1, CAMS, 128, hf, wf, device=dev)
        metric = _metric(3).to(dev)
        with torch.no_grad():
            dense = vt.get_two_hot_depth(vt.downsample_depth(metric, 8))
            a = vt([feat] + inp, dense).clone()
            tw = vt.get_two_hot_windows(metric, downsample=8, eps=0.0)
            assert torch.equal(tw.dense(), dense)          # same expf, same bits
            b = vt([feat] + inp, tw)
        assert torch.equal(a, b), ds


@pytest.mark.gpu
def test_window_set_against_the_oracle_threshold():
    dev = 'cuda:0'
    vt = _vt(dev, 32, [2, 2, 2])
    metric = _metric(4)
    lo, _, step = GRID['depth']
    ref = c_oracle.two_hot_depth(c_oracle.downsample_depth(metric.numpy(), 8), vt.D, lo, step, 4.0)
    for eps in (1e-6, 1e-4):
        tw = vt.get_two_hot_windows(metric.to(dev), downsample=8, eps=eps)
        np.testing.assert_allclose(tw.dense().cpu().numpy(), ref, rtol=1e-5, atol=0)
        kept, want = tw.kept().cpu().numpy(), ref >= eps
        border = np.abs(ref - eps) <= 2e-5 * eps
        assert ((kept != want) & ~border).sum() == 0
        # the CPU mirror builds the same windows (up to borderline weights)
        cpu = depth_ops.two_hot_windows(metric, vt.D, lo, step, 4, eps, downsample=8)
        same = (cpu.kept().numpy() == kept) | border
        assert same.all()

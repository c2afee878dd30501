"""GPU parity of the temporal-path kernels (csrc/temporal.hip) and of the MFMA
route through ``TemporalFusionMultiFrame`` against the PyTorch mirrors, which
tests/test_temporal.py pins to vectors from the reference's own classes.

Tolerances: the gather kernels take bf16 operands on both sides and accumulate in
fp32; what differs is fp32 evaluation order, the fast exp, and the bf16 rounding
of the result: |got - want| <= 2^-7 |want| + 4e-3 rms(want).  The whole fusion
(15+ bf16 layers deep, vs fp32 modules) is held to 3 % of rms.
"""
import pytest
import torch

from veon_amd import conv3d_ops
from veon_amd.models.semantic_net import temporal_fusion as tfm

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, want, k=4e-3):
    rms = want.pow(2).mean().sqrt().item() + 1e-12
    err = (got - want).abs()
    bound = want.abs() * 2.0 ** -7 + k * rms
    assert bool((err <= bound).all()), (err.max().item(), rms)


@pytest.mark.parametrize('B,C,heads,Z,Y,X', [(2, 256, 4, 3, 9, 11), (1, 128, 4, 2, 5, 7),
                                            (1, 256, 4, 1, 1, 1), (1, 64, 2, 8, 13, 10),
                                            (2, 256, 8, 2, 5, 6), (1, 32, 1, 3, 4, 5)])
def test_deform_attention_matches_torch(B, C, heads, Z, Y, X):
    g = torch.Generator().manual_seed(C + X)
    kv = _bf(torch.randn(B, 2 * C, Z, Y, X, generator=g)).to(DEV)
    q = _bf(torch.randn(B, C, Z, Y, X, generator=g)).to(DEV)
    noff = heads * 8 * 3
    pad = (noff + 7) // 8 * 8
    off = _bf(torch.randn(B, pad, Z, Y, X, generator=g) * 1.5).to(DEV)
    mod = tfm.TemporalDeformable(C, num_heads=heads).to(DEV)
    want = mod.attend(kv, q, torch.tanh(off[:, :noff]))
    got = conv3d_ops.deform_attention(conv3d_ops.pack(kv), conv3d_ops.pack(q),
                                      conv3d_ops.pack(off), heads)
    _close(conv3d_ops.unpack(got), want)
    halo = got.rows.view(B, Z + 2, Y + 2, X + 2, C).clone()
    halo[:, 1:-1, 1:-1, 1:-1] = 0
    assert float(halo.abs().sum()) == 0.0


def test_deform_attention_rejects_bad_shapes():
    from veon_amd._lib import VeonHipError
    q = conv3d_ops.PaddedVolume(1, 96, 2, 3, 4, DEV)       # head dim 24
    kv = conv3d_ops.PaddedVolume(1, 192, 2, 3, 4, DEV)
    off = conv3d_ops.PaddedVolume(1, 96, 2, 3, 4, DEV)
    with pytest.raises(VeonHipError):
        conv3d_ops.deform_attention(kv, q, off, 4)
    q = conv3d_ops.PaddedVolume(1, 256, 2, 3, 4, DEV)
    kv = conv3d_ops.PaddedVolume(1, 512, 2, 3, 4, DEV)
    with pytest.raises(VeonHipError):                         # too few offset channels
        conv3d_ops.deform_attention(kv, q, conv3d_ops.PaddedVolume(1, 64, 2, 3, 4, DEV), 4)


@pytest.mark.parametrize('shift', [0.3, 2.5, 40.0])
def test_warp_matches_align_after_lss(shift):
    g = torch.Generator().manual_seed(3)
    grid = {'x': [-4.0, 4.0, 0.5], 'y': [-3.0, 3.0, 0.5], 'z': [-1.0, 3.0, 0.5]}
    ds = (2, 2, 2)
    B, C = 2, 64
    occ = _bf(torch.randn(B, C, 4, 6, 8, generator=g)).to(DEV)

    def rigid(a, t):
        m = torch.eye(4)
        ca, sa = torch.cos(torch.tensor(a)), torch.sin(torch.tensor(a))
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = ca, -sa, sa, ca
        m[:3, 3] = torch.tensor(t)
        return m
    cur = torch.stack([rigid(0.2, [3.0, -2.0, 0.1]), rigid(-0.4, [1.0, 5.0, 0.0])])[:, None]
    prev = torch.stack([cur[0, 0] @ rigid(0.1, [shift, 0.2 * shift, 0.05 * shift]),
                        cur[1, 0] @ rigid(-0.05, [-0.5 * shift, shift, 0.0])])[:, None]
    metas = [cur.to(DEV), prev.to(DEV)]
    want = tfm.align_after_lss(occ, metas, grid, ds)
    got = tfm.align_after_lss(conv3d_ops.pack(occ), metas, grid, ds)
    _close(conv3d_ops.unpack(got), want)
    if shift > 30:
        assert float(want.abs().sum()) == 0.0 and float(got.rows.abs().sum()) == 0.0


def test_zero_halo():
    vol = conv3d_ops.PaddedVolume(2, 64, 2, 3, 5, DEV)
    vol.rows.fill_(1.0)
    conv3d_ops.zero_halo(vol)
    grid = vol.rows.view(2, 4, 5, 7, 64).float()
    assert float(grid[:, 1:-1, 1:-1, 1:-1].min()) == 1.0
    assert float(grid.sum()) == 2 * 2 * 3 * 5 * 64


def _randomise(mod, gen):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.2)


@pytest.mark.parametrize('T', [1, 2])
def test_temporal_fusion_mfma_path_matches_module(T):
    gen = torch.Generator().manual_seed(11 + T)
    torch.manual_seed(11 + T)
    C = 256
    net = tfm.TemporalFusionMultiFrame(C, seqs=T).eval()
    _randomise(net, gen)
    with torch.no_grad():
        net.deform_fusion_layer.t_deform.offset_conv[2].weight.mul_(6.0)
    net = net.to(DEV)
    cur = torch.randn(1, C, 3, 10, 12, generator=gen).to(DEV)
    prevs = [torch.randn(1, C, 3, 10, 12, generator=gen).to(DEV) for _ in range(T)]
    with torch.no_grad():
        assert net.hip_ok(cur)
        want = net(cur, prevs)
        got = net.forward_fast(cur, prevs)
    rms = want.pow(2).mean().sqrt().item()
    err = (got - want).abs()
    assert err.max().item() <= 0.12 * rms and err.pow(2).mean().sqrt().item() <= 0.03 * rms, \
        (err.max().item(), err.pow(2).mean().sqrt().item(), rms)


def _decoder(embed, depth_layers, sd=None):
    from veon_amd.models import build_neck
    from veon_amd.models.semantic_net import AlignNetOcc3D
    net = AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=embed, clip_outdim=24,
                        layer_lifting_map=['2->0->0'], fusion_type='cat_fusion',
                        layer_depth=depth_layers, num_temporal=2)
    if sd is not None:
        net.load_state_dict(sd, strict=True)
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    net.lss_view_transformer = build_neck(dict(
        type='LSSViewTransformerRaw', grid_config=grid, input_size=(64, 176),
        downsample=16, out_channels=embed, collapse_z=False, ds_feat=[2, 2, 2]))
    net.num_frame, net.num_camera = 1, 2
    return net.to(DEV).eval()


def test_temporal_decoder_matches_reference_vectors():
    """AlignNetOcc3D(num_temporal=2): forward_early of a past frame, then
    forward(..., occ_feat_prevs) against the reference's own decoder run on CPU
    (oracle/tools/gen_golden_align_net.py, second half).  embed_dim 32: the fp32
    module definitions run around the HIP lift."""
    from tests.conftest import load_golden
    t = {k: torch.from_numpy(v).to(DEV) for k, v in load_golden('align_net_tiny').items()}
    tt = {k: torch.from_numpy(v).to(DEV)
          for k, v in load_golden('align_net_temporal_tiny').items()}
    net = _decoder(32, 1, {k[3:]: v for k, v in tt.items() if k.startswith('sd/')})
    metas = [t['s2e'], t['e2g'], t['intr'], t['pr'], t['pt'], t['bda'][None]]
    sem_feat = torch.zeros(2, 8, 4, 11, device=DEV)
    with torch.no_grad():
        early = net.forward_early(sem_feat, {1: t['clip1'], 2: tt['clip2_prev']},
                                  [tt['supp_prev']], tt['metric_prev'], metas)
        torch.testing.assert_close(early, tt['early_prev'], rtol=1e-4, atol=1e-4)
        out = net(sem_feat, {1: t['clip1'], 2: t['clip2']}, [t['supp']], t['metric'],
                  metas, [early])
    for key in ('bin_occ', 'feat_occ'):
        rel = ((out[key] - tt[key]).norm() / tt[key].norm()).item()
        assert rel < 1e-3, (key, rel)


def test_temporal_decoder_fast_path_agrees_with_modules():
    """embed_dim 256: past frame lifted straight into a PaddedVolume, warped on the
    grid, fused and decoded on the MFMA path -- against the same decoder with the
    native paths switched off (fp32 modules around the HIP lift)."""
    from tests.conftest import load_golden
    from veon_amd import _lib
    t = {k: torch.from_numpy(v).to(DEV) for k, v in load_golden('align_net_tiny').items()}
    torch.manual_seed(5)
    net = _decoder(256, 1)
    gen = torch.Generator().manual_seed(5)
    _randomise(net.cpu(), gen)
    net = net.to(DEV)
    metas = [t['s2e'], t['e2g'], t['intr'], t['pr'], t['pt'], t['bda'][None]]
    clip = {1: t['clip1'], 2: t['clip2']}
    sem_feat = torch.zeros(2, 8, 4, 11, device=DEV)
    grid = net.lss_view_transformer.grid_config
    eye = torch.eye(4, device=DEV)[None, None]
    move = eye.clone()
    move[0, 0, :3, 3] = torch.tensor([1.3, -0.6, 0.2])
    with torch.no_grad():
        vol = conv3d_ops.PaddedVolume(1, 256, 2, 10, 10, DEV)
        before = dict(_lib.CALLS)
        early = net.forward_early(sem_feat, clip, [t['supp'] * 0.5], t['metric'], metas,
                                  out_volume=vol)
        assert early is vol
        prev = tfm.align_after_lss(early, [eye, move], grid, (2, 2, 2))
        fast = net(sem_feat, clip, [t['supp']], t['metric'], metas, [prev])
        ran = {k for k, v in _lib.CALLS.items() if v > before.get(k, 0)}
        assert {'veon_volume_warp_bf16', 'veon_deform_attention_bf16',
                'veon_conv3d_k3_bf16'} <= ran, ran
        # the fused pool + max-pool: row kernel from 64 channels on, slab kernel below
        assert {'veon_bev_pool_v2_fwd_maxpool_padded',
                'veon_bev_pool_v2_fwd_rows_maxpool_ordered'} & ran, ran
        net.use_hip = False
        for m in (net.occupancy_pred, net.feat_pred):
            m._hip_ok = lambda x: False
        early32 = net.forward_early(sem_feat, clip, [t['supp'] * 0.5], t['metric'], metas)
        prev32 = tfm.align_after_lss(early32, [eye, move], grid, (2, 2, 2))
        slow = net(sem_feat, clip, [t['supp']], t['metric'], metas, [prev32])
    for key in ('bin_occ', 'feat_occ'):
        rel = ((fast[key] - slow[key]).norm() / slow[key].norm()).item()
        assert rel < 4e-2, (key, rel)


def test_occupancy_path_temporal_loop_and_cached_depth(tmp_path):
    """VeonOccupancyPath(num_temporal=2): lift_frame -> align -> forward(prev_volumes)
    runs on the native kernels; a depth map stored in / loaded from the depth
    cache replaces the depth branch with the same result; an identity-aligned copy
    of a kept volume equals the kept volume."""
    from veon_amd import _lib, depth_cache, synthetic
    from veon_amd.models.veon_occ import VeonOccupancyPath
    torch.manual_seed(0)
    size, ncam = (64, 176), 2
    net = VeonOccupancyPath(
        input_size=size, num_cam=ncam, encoder='vitb', clip_width=64, clip_layers=4,
        clip_heads=1, clip_first_tail=2, clip_proj_dim=64, embed_dim=128,
        occ_size=(4, 20, 20), hsa_dim=64, hsa_fusion_map=('0->1->1', '1->2->2'),
        num_temporal=2,
        grid_config={'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0],
                     'z': [-1.0, 3.0, 1.0], 'depth': [1.0, 13.0, 1.0]}).to(DEV).eval()
    geom = [t.to(DEV) for t in synthetic.rig_inputs(synthetic.make_rig(1, ncam, size))]
    cur = torch.randn(1, ncam, 3, *size, device=DEV)
    past = torch.randn(1, ncam, 3, *size, device=DEV)
    eye = torch.eye(4, device=DEV)[None, None]
    move = eye.clone()
    move[0, 0, :3, 3] = torch.tensor([1.0, 0.5, 0.0])
    before = dict(_lib.CALLS)
    with torch.no_grad():
        kept = net.lift_frame(past, geom)
        assert isinstance(kept, conv3d_ops.PaddedVolume)
        same = net.align(kept, [eye, eye])
        assert torch.equal(same.rows, kept.rows)
        out = net(cur, geom, [net.align(kept, [eye, move])])
        single = net(cur, geom)
        # depth cache round trip in place of the depth branch
        depth = net.estimate_depth(cur.flatten(0, 1))
        toks = ['0123abcd-CAM_FRONT', '0123abcd-CAM_BACK']
        depth_cache.store(str(tmp_path), toks, depth)
        cached = depth_cache.load(str(tmp_path), toks, DEV)[None]
        again = net(cur, geom, [net.align(kept, [eye, move])], depth=cached)
    ran = {k for k, v in _lib.CALLS.items() if v > before.get(k, 0)}
    assert {'veon_volume_warp_bf16', 'veon_warp_affine', 'veon_deform_attention_bf16',
            'veon_volume_zero_halo_bf16'} <= ran, ran
    assert out['sem_occ'].shape == (1, 17, 4, 20, 20)
    assert torch.isfinite(out['sem_occ']).all() and torch.isfinite(out['bin_occ']).all()
    assert not torch.equal(out['sem_occ'], single['sem_occ'])
    assert torch.equal(again['sem_occ'], out['sem_occ'])


def test_native_caches_follow_a_state_dict_load():
    """folded / packed weights cached by the MFMA path are rebuilt after
    load_state_dict (a stale cache would silently keep the old weights)."""
    torch.manual_seed(2)
    C = 128
    a = tfm.TemporalFusionMultiFrame(C, seqs=1).to(DEV).eval()
    b = tfm.TemporalFusionMultiFrame(C, seqs=1).to(DEV).eval()
    cur = torch.randn(1, C, 2, 6, 7, device=DEV)
    prev = [torch.randn(1, C, 2, 6, 7, device=DEV)]
    with torch.no_grad():
        ya = a.forward_fast(cur, prev)
        yb = b.forward_fast(cur, prev)
        assert not torch.allclose(ya, yb)
        b.load_state_dict(a.state_dict())
        assert torch.equal(b.forward_fast(cur, prev), ya)

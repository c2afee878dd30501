"""GPU parity of the implicit-GEMM Conv3d body (csrc/conv3d.hip) against plain
PyTorch fp32 ``conv3d`` / ``BatchNorm3d`` on the same bf16-rounded operands.

Tolerance: operands are bf16 on both sides and accumulation is fp32, so the only
differences are the fp32 summation order and the final bf16 rounding of the
kernel's output: |got - want| <= 2^-7 * |want| + 2e-3 * rms(want) per element.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from veon_amd import _lib, conv3d_ops
from veon_amd.models.semantic_net import AlignBody3D, ResBlock3D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close(got, want):
    rms = want.pow(2).mean().sqrt().item() + 1e-12
    err = (got - want).abs()
    bound = want.abs() * 2.0 ** -7 + 2e-3 * rms
    assert bool((err <= bound).all()), (err.max().item(), rms)


def test_pack_unpack_round_trip_and_halo():
    g = torch.Generator().manual_seed(0)
    x = _bf(torch.randn(2, 72, 3, 5, 70, generator=g)).to(DEV)
    vol = conv3d_ops.pack(x)
    back = conv3d_ops.unpack(vol)
    assert torch.equal(back, x)
    grid = vol.rows.view(2, 5, 7, 72, 72).float()
    assert torch.equal(grid[:, 1:-1, 1:-1, 1:-1].permute(0, 4, 1, 2, 3), x)
    halo = grid.clone()
    halo[:, 1:-1, 1:-1, 1:-1] = 0
    assert float(halo.abs().sum()) == 0.0
    assert float(vol.storage[:vol.guard].abs().sum()) == 0.0
    assert float(vol.storage[vol.guard + vol.M:].abs().sum()) == 0.0


@pytest.mark.parametrize('B,Cin,Cout,Z,Y,X', [(1, 64, 64, 3, 5, 7), (2, 128, 72, 2, 9, 33),
                                             (1, 64, 136, 1, 1, 1), (1, 192, 256, 4, 20, 21)])
@pytest.mark.parametrize('mode', ['plain', 'bn_relu', 'bn_resid_relu'])
def test_conv3d_matches_torch(B, Cin, Cout, Z, Y, X, mode):
    g = torch.Generator().manual_seed(Cin + Cout + X)
    x = _bf(torch.randn(B, Cin, Z, Y, X, generator=g)).to(DEV)
    w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv3d(x.double(), w.double(), padding=1).float()
    vol = conv3d_ops.pack(x)
    wp = conv3d_ops.pack_weight(w)
    if mode == 'plain':
        out = conv3d_ops.conv3d_k3(vol, wp)
    elif mode == 'bn_relu':
        want = F.relu(want * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1))
        out = conv3d_ops.conv3d_k3(vol, wp, scale, shift, relu=True)
    else:
        if Cin != Cout:
            pytest.skip('identity add needs Cin == Cout')
        want = F.relu(want * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1) + x)
        out = conv3d_ops.conv3d_k3(vol, wp, scale, shift, resid=vol, relu=True)
    got = conv3d_ops.unpack(out)
    _close(got, want)
    # halo rows written as zeros, guard rows never written
    grid = out.rows.view(B, Z + 2, Y + 2, X + 2, Cout).float().clone()
    grid[:, 1:-1, 1:-1, 1:-1] = 0
    assert float(grid.abs().sum()) == 0.0
    assert float(out.storage[:out.guard].abs().sum()) == 0.0
    assert float(out.storage[out.guard + out.M:].abs().sum()) == 0.0


def test_veon_body_shape_against_torch_on_device():
    """The actual VEON conv (256 -> 256 on 8 x 100 x 100, 283 GFLOP)."""
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(1, 256, 8, 100, 100, generator=g)).to(DEV)
    w = _bf(torch.randn(256, 256, 3, 3, 3, generator=g) * (27 * 256) ** -0.5).to(DEV)
    want = F.conv3d(x, w, padding=1)
    got = conv3d_ops.unpack(conv3d_ops.conv3d_k3(conv3d_ops.pack(x),
                                                 conv3d_ops.pack_weight(w)))
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 4e-3, rel  # bf16 output rounding (2^-9 rms) + MIOpen's own fp32 order


def test_align_body_blocks_match_module_definition():
    """Two ResBlock3D through the HIP stack vs the PyTorch definition with the
    same (bf16-rounded) weights and non-trivial BN statistics."""
    torch.manual_seed(3)
    body = AlignBody3D(embed_dim=64, layer_depth=2)
    for m in body.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Conv3d):
            m.weight.data = _bf(m.weight.data)
    body = body.to(DEV).eval()
    x = _bf(torch.randn(2, 64, 4, 10, 12)).to(DEV)
    before = dict(_lib.CALLS)
    with torch.no_grad():
        got = body(x)
        body.use_hip = False
        want = body(x)
    assert _lib.CALLS['veon_conv3d_k3_bf16'] - before.get('veon_conv3d_k3_bf16', 0) == 4
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 1.5e-2, rel  # three bf16 roundings of intermediate volumes
    # partial ranges compose (the reference applies one block per fusion step)
    body.use_hip = True
    with torch.no_grad():
        two = body(body(x, 0, 1), 1, 2)
    assert ((two - got).norm() / got.norm()).item() < 1e-2


def test_training_and_cpu_take_the_torch_definition():
    blk = ResBlock3D(64, 64)
    y = blk(torch.randn(1, 64, 2, 3, 3))
    assert y.shape == (1, 64, 2, 3, 3) and y.requires_grad


def test_prediction_heads_match_module_definition():
    """PredHead3DOcc / PredHead3DSem: the 1x1x1 ConvModule chains as GEMMs with
    the BN fold and ReLU in the epilogue, fed straight from the body's padded
    volume, vs the PyTorch definition on the body's fp32 output."""
    from veon_amd.models.semantic_net import PredHead3DOcc, PredHead3DSem
    torch.manual_seed(11)
    body = AlignBody3D(embed_dim=256, layer_depth=1)
    occ, sem = PredHead3DOcc(256, 2), PredHead3DSem(256, 96)
    for mod in (body, occ, sem):
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
            if isinstance(m, torch.nn.Conv3d):
                m.weight.data = _bf(m.weight.data)
        mod.to(DEV).eval()
    x = _bf(torch.randn(2, 256, 3, 9, 10)).to(DEV)
    before = dict(_lib.CALLS)
    with torch.no_grad():
        vol = body(x, return_volume=True)
        assert isinstance(vol, conv3d_ops.PaddedVolume)
        got_occ, got_sem = occ(vol), sem(vol)
        feat = conv3d_ops.unpack(vol)       # what the heads saw, as fp32
        occ_cpu, sem_cpu = occ.cpu(), sem.cpu()
        want_occ, want_sem = occ_cpu(feat.cpu()), sem_cpu(feat.cpu())
    assert _lib.CALLS['veon_vit_gemm'] - before.get('veon_vit_gemm', 0) == 5
    assert got_occ.shape == (2, 2, 3, 9, 10) and got_sem.shape == (2, 96, 3, 9, 10)
    for got, want in ((got_occ, want_occ), (got_sem, want_sem)):
        rel = ((got.cpu() - want).norm() / want.norm()).item()
        assert rel < 1.5e-2, rel   # bf16 intermediates between the convs
    assert float(got_sem.abs().max()) <= 0.5


def test_lift_writes_the_body_input_directly():
    """LSSViewTransformerRaw.forward(out_volume=...) + AlignBody3D on that
    volume == pack(lift output) + body, bit for bit, and the lift's volume is
    left intact by the body."""
    from veon_amd import synthetic
    from veon_amd.models import build_neck
    size, C = (128, 352), 64
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=synthetic.GRID_VEON,
                         input_size=size, out_channels=C, collapse_z=False,
                         ds_feat=[2, 2, 2])).to(DEV).eval()
    geom = [t.to(DEV) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    depth, feat = synthetic.make_depth_feat(1, 6, vt.D, C, size[0] // 16, size[1] // 16, 0)
    depth, feat = depth.to(DEV), feat.to(DEV)
    for sync_free in (False, True):
        vt.sync_free = sync_free
        with torch.no_grad():
            ref = vt([feat] + geom, depth)                     # (B,C,Zo,Yo,Xo) fp32
            assert ref.shape == (1, C, 8, 100, 100) and float(ref.abs().sum()) > 0
            vol = conv3d_ops.PaddedVolume(*ref.shape, DEV)
            got = vt([feat] + geom, depth, out_volume=vol)
            assert got is vol
            assert torch.equal(conv3d_ops.unpack(vol), _bf(ref))
    with torch.no_grad():
        body = AlignBody3D(embed_dim=C, layer_depth=2).to(DEV).eval()
        keep = vol.rows.clone()
        a = body(vol)
        b = body(ref)
        assert torch.equal(a, b)
        assert torch.equal(vol.rows, keep)
        # chaining partial ranges through volumes
        c = body(body(vol, 0, 1, return_volume=True), 1, 2)
        assert torch.equal(a, c)


def test_hip_body_and_heads_against_reference_vectors():
    """The MFMA path on the reference-generated fixture (bf16 tolerance)."""
    from tests.conftest import load_golden
    from veon_amd.models.semantic_net import PredHead3DOcc, PredHead3DSem
    g = load_golden('align_body_tiny')

    def sd(tag):
        return {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g
                if k.startswith(tag + '/')}
    body = AlignBody3D(embed_dim=64, layer_depth=1)
    body.layers_3d_body[0].load_state_dict(sd('block'), strict=True)
    sem = PredHead3DSem(64, 24)
    sem.load_state_dict(sd('sem'), strict=True)
    occ = PredHead3DOcc(64, 2)
    occ.load_state_dict(sd('occ'), strict=True)
    body, sem, occ = body.to(DEV).eval(), sem.to(DEV).eval(), occ.to(DEV).eval()
    x = torch.from_numpy(g['x']).to(DEV)
    before = dict(_lib.CALLS)
    with torch.no_grad():
        vol = body(x, return_volume=True)
        y = conv3d_ops.unpack(vol)
        s = sem(vol)
        o = occ(vol)      # 64 -> 16 -> 2: K = 16 is not MFMA-shaped, torch path
    assert _lib.CALLS['veon_conv3d_k3_bf16'] - before.get('veon_conv3d_k3_bf16', 0) == 2
    for got, key, tol in ((y, 'block_out', 1e-2), (s, 'sem_out', 2e-2), (o, 'occ_out', 2e-2)):
        want = torch.from_numpy(g[key])
        rel = ((got.cpu() - want).norm() / want.norm()).item()
        assert rel < tol, (key, rel)


@pytest.mark.parametrize('B,Cin,Cout,Y,X', [(1, 64, 64, 5, 7), (2, 128, 128, 9, 33),
                                           (1, 64, 32, 20, 41), (6, 128, 64, 36, 50)])
@pytest.mark.parametrize('mode', ['bias', 'bias_relu', 'bias_resid'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv2d_matches_torch(B, Cin, Cout, Y, X, mode, dtype):
    """The 2-D (9-tap) mode of the conv kernel, incl. the 64-wide tiles and the
    bf16 planar pack / unpack used by the DPT head."""
    if mode == 'bias_resid' and Cin != Cout:
        pytest.skip('identity add needs Cin == Cout')
    g = torch.Generator().manual_seed(Cin * 7 + Cout + X)
    x = _bf(torch.randn(B, Cin, Y, X, generator=g)).to(DEV)
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    want = F.conv2d(x.double(), w.double(), bias.double(), padding=1).float()
    img = conv3d_ops.pack_image(x.to(dtype))
    wp = conv3d_ops.pack_weight2d(w)
    shift = torch.zeros(wp.shape[0], device=DEV)
    shift[:Cout] = bias
    if mode == 'bias':
        out = conv3d_ops.conv2d_k3(img, wp, None, shift)
    elif mode == 'bias_relu':
        want = F.relu(want)
        out = conv3d_ops.conv2d_k3(img, wp, None, shift, relu=True)
    else:
        want = want + x
        out = conv3d_ops.conv2d_k3(img, wp, None, shift, resid=img)
    got = conv3d_ops.unpack_image(out, dtype, channels=Cout).float()
    if dtype == torch.bfloat16:
        assert torch.equal(got, conv3d_ops.unpack_image(out, torch.float32, Cout))
    _close(got, want)
    grid = out.rows.view(B, Y + 2, X + 2, -1).float().clone()
    grid[:, 1:-1, 1:-1] = 0
    assert float(grid.abs().sum()) == 0.0


@pytest.mark.parametrize('B,C,Y,X', [(1, 64, 5, 7), (6, 256, 36, 50), (2, 128, 9, 33)])
def test_conv2d_second_residual_and_relu_output(B, C, Y, X):
    """veon_conv2d_k3_bf16_ex: out = conv + bias + resid + resid2 and a second image
    holding relu(out) -- FeatureFusionBlock's x0 + RCU1(x1) and the ReLU in front of the
    next ResidualConvUnit (util/blocks.py:49-148) out of one epilogue.  Against torch in
    fp64 on the bf16 operands; out_relu is the ReLU of the ROUNDED result (what a separate
    pass over `out` would give); the halo of both outputs stays zero; each extra alone."""
    g = torch.Generator().manual_seed(C + X)
    xs = [_bf(torch.randn(B, C, Y, X, generator=g)).to(DEV) for _ in range(3)]
    x, r1, r2 = (conv3d_ops.pack_image(t.to(torch.bfloat16)) for t in xs)
    w = _bf(torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    wp = conv3d_ops.pack_weight2d(w)
    shift = torch.zeros(wp.shape[0], device=DEV)
    shift[:C] = bias
    conv = F.conv2d(xs[0].double(), w.double(), bias.double(), padding=1)
    out = conv3d_ops.PaddedImage(B, C, Y, X, DEV)
    rel = conv3d_ops.PaddedImage(B, C, Y, X, DEV)
    conv3d_ops.conv2d_k3(x, wp, None, shift, resid=r1, resid2=r2, out=out, out_relu=rel)
    got = conv3d_ops.unpack_image(out, torch.float32, C)
    _close(got, (conv + xs[1].double() + xs[2].double()).float())
    assert torch.equal(conv3d_ops.unpack_image(rel, torch.float32, C), got.clamp_min(0))
    for t in (out, rel):
        grid = t.rows.view(B, Y + 2, X + 2, -1).float().clone()
        grid[:, 1:-1, 1:-1] = 0
        assert float(grid.abs().sum()) == 0.0
    # each extra alone
    one = conv3d_ops.unpack_image(conv3d_ops.conv2d_k3(x, wp, None, shift, resid=r1),
                                  torch.float32, C)
    only_relu = conv3d_ops.PaddedImage(B, C, Y, X, DEV)
    o2 = conv3d_ops.conv2d_k3(x, wp, None, shift, resid=r1, out_relu=only_relu)
    assert torch.equal(conv3d_ops.unpack_image(o2, torch.float32, C), one)
    assert torch.equal(conv3d_ops.unpack_image(only_relu, torch.float32, C), one.clamp_min(0))
    o3 = conv3d_ops.conv2d_k3(x, wp, None, shift, resid2=r2)
    _close(conv3d_ops.unpack_image(o3, torch.float32, C), (conv + xs[2].double()).float())


@pytest.mark.parametrize('size_in,size_out', [((4, 6), (8, 12)), ((9, 25), (18, 50)),
                                              ((144, 400), (252, 700)), ((5, 5), (1, 1))])
def test_resize_bilinear_matches_interpolate(size_in, size_out):
    g = torch.Generator().manual_seed(size_in[0])
    x = _bf(torch.randn(2, 64, *size_in, generator=g)).to(DEV)
    want = F.interpolate(x, size_out, mode='bilinear', align_corners=True)
    img = conv3d_ops.pack_image(x)
    out = conv3d_ops.resize_bilinear(img, size_out)
    got = conv3d_ops.unpack_image(out)
    # fp32 blends of the same four bf16 taps, one bf16 rounding at the end
    assert float((got - _bf(want)).abs().max()) <= 2.0 ** -6 * float(want.abs().max())
    grid = out.rows.view(2, size_out[0] + 2, size_out[1] + 2, 64).float().clone()
    grid[:, 1:-1, 1:-1] = 0
    assert float(grid.abs().sum()) == 0.0


def test_fused_semantic_inference_on_padded_volume():
    """PredHead3DSem(return_volume) -> classifier GEMM on the padded rows -> upsample
    of the class logits, vs the reference order on the head's fp32 output."""
    from veon_amd.models.semantic_net import (PredHead3DSem, semantic_inference_3d,
                                              semantic_inference_3d_fused)
    torch.manual_seed(2)
    sem = PredHead3DSem(64, 128).to(DEV).eval()
    for m in sem.modules():
        if isinstance(m, torch.nn.Conv3d):
            m.weight.data = _bf(m.weight.data)
    W = (torch.randn(17, 128, device=DEV) * 2)
    x = _bf(torch.randn(1, 64, 4, 9, 11)).to(DEV)
    with torch.no_grad():
        vol = sem(conv3d_ops.pack(x), return_volume=True)
        assert isinstance(vol, conv3d_ops.PaddedVolume)
        got = semantic_inference_3d_fused(W, vol, (8, 18, 22))
        feat = sem(x)                                  # (1,128,4,9,11) fp32
        want = semantic_inference_3d(W, feat, (8, 18, 22))
    assert got.shape == want.shape == (1, 17, 8, 18, 22)
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 1.5e-2, rel


@pytest.mark.parametrize('seed', range(10))
def test_conv_random_shapes(seed):
    """Random small volumes / images and channel counts (all tile families, ragged
    last tiles, batch > 1, single-voxel planes) against torch in fp64."""
    rng = np.random.default_rng(4000 + seed)
    three_d = bool(seed % 2)
    B = int(rng.integers(1, 4))
    Cin = int(rng.choice([64, 128, 192]))
    Cout = int(rng.choice([8, 32, 64, 72, 128, 256, 264]))
    Z = int(rng.integers(1, 5)) if three_d else 1
    Y, X = int(rng.integers(1, 24)), int(rng.integers(1, 40))
    g = torch.Generator().manual_seed(seed)
    relu = bool(rng.integers(0, 2))
    use_res = Cin == Cout and bool(rng.integers(0, 2))
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    if three_d:
        x = _bf(torch.randn(B, Cin, Z, Y, X, generator=g)).to(DEV)
        w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).to(DEV)
        want = F.conv3d(x.double(), w.double(), padding=1).float()
        bc = (1, -1, 1, 1, 1)
        vol = conv3d_ops.pack(x)
        out = conv3d_ops.conv3d_k3(vol, conv3d_ops.pack_weight(w), scale, shift,
                                   resid=vol if use_res else None, relu=relu)
        got = conv3d_ops.unpack(out)
    else:
        x = _bf(torch.randn(B, Cin, Y, X, generator=g)).to(DEV)
        w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(DEV)
        want = F.conv2d(x.double(), w.double(), padding=1).float()
        bc = (1, -1, 1, 1)
        img = conv3d_ops.pack_image(x)
        out = conv3d_ops.conv2d_k3(img, conv3d_ops.pack_weight2d(w), scale, shift,
                                   resid=img if use_res else None, relu=relu)
        got = conv3d_ops.unpack_image(out, channels=Cout)
    want = want * scale.view(bc) + shift.view(bc)
    if use_res:
        want = want + x
    if relu:
        want = F.relu(want)
    _close(got, want)


def test_align_net_decoder_against_reference_vectors():
    """The whole occupancy decoder (depth prep -> CatFusionLift -> lift + max-pool
    -> ResBlock3D x2 -> heads) vs the reference's own AlignNetOcc3D run on CPU
    (oracle/tools/gen_golden_align_net.py).  Lifted volume: fp32, differs from
    the reference's index_add_ order only; decoder outputs: fp32 module path
    tight, MFMA fast path within bf16 tolerance."""
    from tests.conftest import load_golden
    from veon_amd.models import build_neck
    from veon_amd.models.semantic_net import AlignNetOcc3D
    g = load_golden('align_net_tiny')
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    net = AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=64, clip_outdim=24,
                        layer_lifting_map=['2->0->0'], fusion_type='cat_fusion',
                        layer_depth=2)
    net.load_state_dict({k[3:]: v for k, v in t.items() if k.startswith('sd/')})
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=grid,
                         input_size=(64, 176), downsample=16, out_channels=64,
                         collapse_z=False, ds_feat=[2, 2, 2]))
    net.lss_view_transformer = vt
    net.num_frame, net.num_camera = 1, 2
    net = net.to(DEV).eval()
    metas = [t['s2e'], t['e2g'], t['intr'], t['pr'], t['pt'], t['bda'][None]]
    clip = {1: t['clip1'], 2: t['clip2']}
    sem_feat = torch.zeros(2, 8, 4, 11, device=DEV)
    with torch.no_grad():
        lifted = net.forward_early(sem_feat, clip, [t['supp']], t['metric'], metas)
        torch.testing.assert_close(lifted, t['lifted'], rtol=1e-4, atol=1e-4)
        before = dict(_lib.CALLS)
        fast = net(sem_feat, clip, [t['supp']], t['metric'], metas)
        ran = {k for k, v in _lib.CALLS.items() if v > before.get(k, 0)}
        assert {'veon_conv3d_k3_bf16', 'veon_vit_gemm'} <= ran, ran
        assert {'veon_bev_pool_v2_fwd_maxpool_padded',
                'veon_bev_pool_v2_fwd_rows_maxpool_ordered'} & ran, ran
        net.use_hip = False
        for m in (net.occupancy_pred, net.feat_pred):
            m._hip_ok = lambda x: False
        slow = net(sem_feat, clip, [t['supp']], t['metric'], metas)
    for key in ('bin_occ', 'feat_occ'):
        want = t[key]
        rel_slow = ((slow[key] - want).norm() / want.norm()).item()
        rel_fast = ((fast[key] - want).norm() / want.norm()).item()
        assert rel_slow < 1e-3, (key, rel_slow)
        assert rel_fast < 2.5e-2, (key, rel_fast)


def test_hsa_network_convblocks_on_mfma():
    """HighresSideAdaptorNetwork with conv_dtype = bf16: the ConvBlocks (bias +
    GELU fused) run on the 2-D MFMA conv kernel; outputs within bf16 tolerance of
    the reference vectors (the attention biases are Gram matrices of the head
    output, so their error is about twice the feature error)."""
    from tests.conftest import load_golden
    from tests.test_host_logic import _hsa_from_golden
    net, t = _hsa_from_golden(load_golden('hsa_tiny'), DEV)
    net.set_conv_dtype(torch.bfloat16)
    before = _lib.CALLS.get('veon_conv2d_k3_bf16', 0)
    with torch.no_grad():
        cb = net.hsa_net_body[0].ff(t['tokens'], (4, 6))
        _, attns, supp = net(t['image'], {1: t['clip1'], 2: t['clip2']})
    assert _lib.CALLS.get("veon_conv2d_k3_bf16", 0) - before == 2 + 6
    for got, key, tol in ((cb, 'convblock_out', 2e-2), (supp, 'supp', 3e-2),
                          (attns, 'attns', 6e-2)):
        rel = ((got - t[key]).norm() / t[key].norm()).item()
        assert rel < tol, (key, rel)


def test_veon_occupancy_path_harness_runs_and_is_stream_invariant():
    """VeonOccupancyPath (depth model + CLIP trunk / HSA / tail + decoder +
    classifier): shapes, finiteness, native kernels in use, and the two-stream
    schedule gives the same result as the serial one."""
    from veon_amd import synthetic
    from veon_amd.models.veon_occ import VeonOccupancyPath
    torch.manual_seed(0)
    size, ncam = (64, 176), 2
    net = VeonOccupancyPath(
        input_size=size, num_cam=ncam, encoder='vitb', clip_width=64, clip_layers=4,
        clip_heads=1, clip_first_tail=2, clip_proj_dim=64, embed_dim=64,
        occ_size=(4, 20, 20), hsa_dim=64, hsa_fusion_map=('0->1->1', '1->2->2'),
        grid_config={'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0],
                     'z': [-1.0, 3.0, 1.0], 'depth': [1.0, 13.0, 1.0]}).to(DEV).eval()
    geom = [t.to(DEV) for t in synthetic.rig_inputs(synthetic.make_rig(1, ncam, size))]
    images = torch.randn(1, ncam, 3, *size, device=DEV)
    before = dict(_lib.CALLS)
    with torch.no_grad():
        a = net(images, geom)
        net.two_streams = False
        b = net(images, geom)
    ran = {k for k, v in _lib.CALLS.items() if v > before.get(k, 0)}
    assert {'veon_vit_block', 'veon_conv2d_k3_bf16', 'veon_conv3d_k3_bf16',
            'veon_two_hot_window'} <= ran, ran
    assert {'veon_bev_pool_v2_fwd_maxpool_padded',
            'veon_bev_pool_v2_fwd_rows_maxpool_ordered'} & ran, ran
    assert a['sem_occ'].shape == (1, 17, 4, 20, 20) and a['bin_occ'].shape == (1, 2, 4, 20, 20)
    assert a['occ_pred_cls'].shape == (1, 20, 20, 4)
    for k in ('sem_occ', 'bin_occ'):
        assert torch.isfinite(a[k]).all()
        assert torch.equal(a[k], b[k]), k


def test_cat_fusion_lift_mfma_path_feeds_the_lift_without_a_copy():
    """CatFusionLift.hip_dtype = bf16: LN + 1x1 conv + ReLU as LayerNorm rows ->
    GEMM, result channels-last bf16 handed over as an NCHW view -- exactly the
    (B,N,H,W,C) half-precision feature rows the pool kernels gather."""
    from veon_amd.models.semantic_net import CatFusionLift
    torch.manual_seed(4)
    fl = CatFusionLift(64, 128, 256).to(DEV).eval()
    with torch.no_grad():
        for ln in (fl.input_proj_1[0], fl.input_proj_2[0]):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.2)
    x1 = torch.randn(4, 64, 9, 13, device=DEV)
    x2 = torch.randn(4, 128, 5, 7, device=DEV)
    with torch.no_grad():
        want = fl(x1, x2, (8, 11))
        fl.hip_dtype = torch.bfloat16
        got = fl(x1, x2, (8, 11))
    assert got.dtype == torch.bfloat16 and got.shape == want.shape == (4, 256, 8, 11)
    assert got.permute(0, 2, 3, 1).is_contiguous()
    rel = ((got.float() - want).norm() / want.norm()).item()
    assert rel < 1.5e-2, rel


@pytest.mark.parametrize('C,Y,X', [(384, 5, 7), (64, 3, 4), (1024, 2, 3), (520, 4, 4)])
def test_image_layernorm_matches_torch(C, Y, X):
    """LayerNorm over the channels of a padded image: padded bf16 result (zero
    halo) and compact fp32 tokens, against F.layer_norm on the same bf16 rows."""
    g = torch.Generator().manual_seed(C)
    B = 2
    x = _bf(torch.randn(B, C, Y, X, generator=g) * 2 + 0.5).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    img = conv3d_ops.pack_image(x)
    want = F.layer_norm(x.permute(0, 2, 3, 1), (C,), gamma, beta, 1e-5)     # (B,Y,X,C)
    tok = conv3d_ops.image_layernorm(img, gamma, beta, 1e-5, tokens=True)
    torch.testing.assert_close(tok, want.reshape(B, Y * X, C), rtol=1e-5, atol=1e-5)
    res = torch.randn(B, Y * X, C, generator=g).to(DEV)
    tok2 = conv3d_ops.image_layernorm(img, gamma, beta, 1e-5, tokens=True, residual=res)
    torch.testing.assert_close(tok2, want.reshape(B, Y * X, C) + res, rtol=1e-5, atol=1e-5)
    out = conv3d_ops.image_layernorm(img, gamma, beta, 1e-5)
    grid = out.rows.view(B, Y + 2, X + 2, C).float()
    _close(grid[:, 1:-1, 1:-1], want)
    halo = grid.clone()
    halo[:, 1:-1, 1:-1] = 0
    assert float(halo.abs().sum()) == 0.0


def test_layernorm_tokens_to_padded_image_and_convblock_fusions():
    """veon_layernorm_f32_to_padded == LayerNorm + staging into the padded bf16
    image; and a dim-384 ConvBlock with pre_ln / residual folded into its first /
    last kernel agrees with the same block given them separately."""
    from veon_amd.models.semantic_net.hsa_network import ConvBlock
    g = torch.Generator().manual_seed(9)
    B, Y, X, C = 2, 5, 7, 384
    x = (torch.randn(B, Y * X, C, generator=g) * 2 + 0.3).to(DEV)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.3)
        img = conv3d_ops.PaddedImage(B, C, Y, X, DEV)
        conv3d_ops.layernorm_tokens_to_image(x, ln.weight, ln.bias, ln.eps, img)
        want = ln(x).view(B, Y, X, C)
        grid = img.rows.view(B, Y + 2, X + 2, C).float()
        _close(grid[:, 1:-1, 1:-1], want)
        halo = grid.clone()
        halo[:, 1:-1, 1:-1] = 0
        assert float(halo.abs().sum()) == 0.0
        blk = ConvBlock(C, C).to(DEV).eval()
        blk.conv_dtype = torch.bfloat16
        before = _lib.CALLS.get('veon_layernorm_f32_to_padded', 0)
        fused = blk(x, (Y, X), residual=x, pre_ln=ln)
        assert _lib.CALLS.get('veon_layernorm_f32_to_padded', 0) == before + 1
        apart = blk(ln(x), (Y, X)) + x
    rel = ((fused - apart).norm() / apart.norm()).item()
    assert rel < 5e-3, rel


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 64, 64, 9, 13), (1, 128, 192, 18, 50), (2, 64, 8, 5, 4)])
def test_conv2d_stride2_matches_torch(shape):
    """veon_conv2d_k3s2_bf16 (Conv2d k=3, s=2, p=1; DPTHead.resize_layers[3]) against
    torch's conv2d on the same bf16-rounded operands; odd and even sizes; also equal
    to the stride-1 kernel sampled at every second pixel."""
    import torch
    import torch.nn.functional as F
    from veon_amd import conv3d_ops
    B, Cin, Cout, Y, X = shape
    torch.manual_seed(0)
    x = torch.randn(B, Cin, Y, X, device='cuda:0').bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, device='cuda:0') * (9 * Cin) ** -0.5)
    bias = torch.randn(Cout, device='cuda:0')
    wp = conv3d_ops.pack_weight2d(w)
    shift = torch.zeros(wp.shape[0], device='cuda:0')
    shift[:Cout] = bias
    img = conv3d_ops.pack_image(x)
    out = conv3d_ops.conv2d_k3s2(img, wp, None, shift)
    got = conv3d_ops.unpack_image(out, torch.float32, Cout)
    ref = F.conv2d(x.float(), w.bfloat16().float(), bias, stride=2, padding=1)
    assert got.shape == ref.shape
    err = (got - ref).abs()
    assert bool((err <= 2 ** -7 * ref.abs() + 2e-3 * ref.pow(2).mean().sqrt()).all()), err.max()
    full = conv3d_ops.unpack_image(conv3d_ops.conv2d_k3(img, wp, None, shift), torch.float32,
                                   Cout)
    # (bit-equal only when both kernels sum the taps in the same order: Cin = 64)
    torch.testing.assert_close(got, full[:, :, ::2, ::2], rtol=2 ** -7, atol=2e-2)
    if Cin == 64:
        assert torch.equal(got, full[:, :, ::2, ::2])
    # the halo of the result is zero: it can feed the next conv directly
    rows = out.rows.view(B, out.shape[2] + 2, out.shape[3] + 2, -1)
    assert not rows[:, 0].any() and not rows[:, -1].any()
    assert not rows[:, :, 0].any() and not rows[:, :, -1].any()


def test_body_conv_is_deterministic_under_load():
    """The 3x3x3 body conv at the VEON shape (249 workgroups of 12 waves, one per CU),
    30 launches on the same operands: bit-identical."""
    g = torch.Generator().manual_seed(7)
    x = _bf(torch.randn(1, 256, 8, 100, 100, generator=g)).to(DEV)
    w = _bf(torch.randn(256, 256, 3, 3, 3, generator=g) * (27 * 256) ** -0.5).to(DEV)
    vol = conv3d_ops.pack(x.to(torch.bfloat16))
    wp = conv3d_ops.pack_weight(w)
    scale = torch.rand(256, generator=g).to(DEV) + 0.5
    shift = torch.randn(256, generator=g).to(DEV)
    first = conv3d_ops.conv3d_k3(vol, wp, scale, shift, relu=True).rows.clone()
    for _ in range(30):
        assert torch.equal(conv3d_ops.conv3d_k3(vol, wp, scale, shift, relu=True).rows, first)

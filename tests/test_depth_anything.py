"""DepthAnythingV2 / DINOv2 mirror against golden vectors produced by the
reference's own Python (oracle/tools/gen_golden_vit.py).

* CPU (fp32 torch path): must reproduce the reference to fp32 rounding.
* GPU (MFMA bf16 path): stated bf16 tolerance.
"""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden
from veon_amd.models.depth_anything import dinov2, dpt


def _build(g):
    d, depth, heads, lora_r = (int(v) for v in g['cfg'])
    enc = dinov2.DinoVisionTransformer(
        img_size=70, patch_size=14, embed_dim=d, depth=depth, num_heads=heads,
        mlp_ratio=4, init_values=1.0, lora_r=lora_r)
    head = dpt.DPTHead(d, features=8, use_bn=False, out_channels=[4, 8, 16, 16])
    enc_sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('enc.')}
    head_sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('head.')}
    missing, unexpected = enc.load_state_dict(enc_sd, strict=True), None
    head.load_state_dict(head_sd, strict=True)
    return enc.eval(), head.eval()


def test_state_dict_names_match_reference():
    g = load_golden('dinov2_tiny')
    enc, head = _build(g)          # strict load = identical key set
    assert any(k.endswith('attn.qkv.lora_A') for k in enc.state_dict())
    assert enc.blocks[0].attn.qkv.merged          # eval() merged the LoRA update


def test_cpu_path_reproduces_reference():
    g = load_golden('dinov2_tiny')
    enc, head = _build(g)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        feats = enc.get_intermediate_layers(x, [int(t) for t in g['taps']],
                                            return_class_token=True)
        final = enc.forward_features(x)
        depth = head(feats, 2, 3) * 80.0
    for i, (p, c) in enumerate(feats):
        np.testing.assert_allclose(p.numpy(), g['tap_patch'][i], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c.numpy(), g['tap_cls'][i], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(final['x_prenorm'].numpy(), g['x_prenorm'],
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(depth.squeeze(1).numpy(), g['depth'], rtol=1e-4,
                               atol=1e-4)


def test_train_mode_unmerges_lora():
    g = load_golden('dinov2_tiny')
    enc, _ = _build(g)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        a = enc.forward_features(x)['x_prenorm']
        enc.train()
        assert not enc.blocks[0].attn.qkv.merged
        b = enc.forward_features(x)['x_prenorm']
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_gpu_mfma_path_within_bf16_tolerance():
    g = load_golden('dinov2_tiny')
    enc, head = _build(g)
    enc, head = enc.to('cuda:0'), head.to('cuda:0')
    x = torch.from_numpy(g['x']).to('cuda:0')
    with torch.no_grad():
        assert enc._use_hip(x)
        feats = enc.get_intermediate_layers(x, [int(t) for t in g['taps']],
                                            return_class_token=True)
        final = enc.forward_features(x)
        depth = head(feats, 2, 3) * 80.0
        enc.use_hip = False
        ref_final = enc.forward_features(x)['x_prenorm']
    # bf16 operands (8-bit mantissa), fp32 accumulation and residual stream:
    # relative L2 error of the features <= 1e-2 after 4 blocks, depth map (after
    # the fp32 DPT head, sigmoid * 80 m) within 0.25 m
    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(final['x_prenorm'].cpu().numpy(), g['x_prenorm']) < 1e-2
    assert rel(final['x_prenorm'].cpu().numpy(), ref_final.cpu().numpy()) < 1e-2
    for i, (p, c) in enumerate(feats):
        assert rel(p.cpu().numpy(), g['tap_patch'][i]) < 1e-2
    assert np.abs(depth.squeeze(1).cpu().numpy() - g['depth']).max() < 0.25


@pytest.mark.gpu
def test_gpu_vitb_shape_mfma_vs_torch_fp32():
    """ViT-B geometry at the real token count (252x700 -> 18x50+1 = 901 tokens,
    2 images, 3 blocks): MFMA path vs the fp32 torch path on the same weights."""
    torch.manual_seed(0)
    enc = dinov2.DinoVisionTransformer(
        img_size=518, patch_size=14, embed_dim=768, depth=3, num_heads=12,
        mlp_ratio=4, init_values=1.0, lora_r=16).to('cuda:0').eval()
    x = torch.randn(2, 3, 252, 700, device='cuda:0')
    with torch.no_grad():
        a = enc.forward_features(x)['x_prenorm']
        enc.use_hip = False
        b = enc.forward_features(x)['x_prenorm']
    rel = (a - b).norm() / b.norm()
    assert a.shape == (2, 901, 768) and rel < 1e-2, rel.item()


@pytest.mark.gpu
def test_encoder_is_graph_capturable():
    from veon_amd.graphs import GraphedCallable
    g = load_golden('dinov2_tiny')
    enc, head = _build(g)
    enc = enc.to('cuda:0')
    x = torch.from_numpy(g['x']).to('cuda:0')
    with torch.no_grad():
        want = enc.forward_features(x)['x_prenorm'].clone()
    graphed = GraphedCallable(lambda im: enc.forward_features(im)['x_prenorm'], (x,))
    got = graphed(x).clone()
    assert torch.equal(got, want)
    got2 = graphed(x * 0.5).clone()
    with torch.no_grad():
        want2 = enc.forward_features(x * 0.5)['x_prenorm']
    assert torch.equal(got2, want2)


@pytest.mark.gpu
def test_dpt_head_bf16_mfma_convs_match_autocast():
    """head_dtype = bf16: the ResidualConvUnits and output convs run on the 2-D
    implicit-GEMM kernel; same result as PyTorch's bf16 autocast of the same
    head within bf16 noise, and the native conv really ran."""
    import torch
    from veon_amd import _lib
    from veon_amd.models import build_neck
    from veon_amd.models.depth_anything import dpt
    torch.manual_seed(0)
    m = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0, use_lora=True,
                        lora_r=4, encoder='vits', features=128,
                        out_channels=[48, 96, 192, 384])).to('cuda:0').eval()
    x = torch.randn(2, 3, 56, 84, device='cuda:0')
    m.head_dtype = torch.bfloat16
    with torch.no_grad():
        feats = m.encode(x)
        before = _lib.CALLS.get('veon_conv2d_k3_bf16', 0)
        got = m.decode(feats, 4, 6)
        ran = _lib.CALLS.get('veon_conv2d_k3_bf16', 0) - before
        ok = dpt._hip_convs_ok
        dpt._hip_convs_ok = lambda *a, **k: False          # PyTorch autocast path
        try:
            want = m.decode(feats, 4, 6)
        finally:
            dpt._hip_convs_ok = ok
        m.head_dtype = None
        ref32 = m.decode(feats, 4, 6)
    assert ran == 2 + 3 * 4 + 2, ran   # refinenet4: 2 convs; 3 blocks x 4; 2 output convs
    assert got.shape == want.shape == (2, 1, 56, 84)
    err_hip = ((got - ref32).norm() / ref32.norm()).item()
    err_amp = ((want - ref32).norm() / ref32.norm()).item()
    assert err_hip < max(2.0 * err_amp, 2e-2), (err_hip, err_amp)


@pytest.mark.gpu
def test_native_depth_forward_tokens_to_depth():
    """forward() with head_dtype = bf16: LayerNormed bf16 token rows of the four taps
    -> projection / transposed-conv GEMMs -> pixel-shuffle pack -> 3x3 convs (one of
    them stride 2) -> fusion blocks, no PyTorch op in the head.  Against
    the fp32 head on the same encoder output, and against PyTorch's bf16 autocast of
    the head (the error budget)."""
    import torch
    from veon_amd import _lib
    from veon_amd.models import build_neck
    from veon_amd.models.depth_anything import dpt
    torch.manual_seed(0)
    m = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0, use_lora=True,
                        lora_r=4, encoder='vits', features=128,
                        out_channels=[48, 96, 192, 384])).to('cuda:0').eval()
    with torch.no_grad():   # default-initialised biases are zero: make every term count
        for p_ in m.depth_head.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.05)
    x = torch.randn(2, 3, 70, 98, device='cuda:0')     # 5 x 7 patches (odd: s2 conv edge)
    with torch.no_grad():
        m.head_dtype = torch.bfloat16
        before = dict(_lib.CALLS)
        got = m(x)['metric_depth']
        ran = {k: _lib.CALLS.get(k, 0) - before.get(k, 0)
               for k in ('veon_tokens_to_image', 'veon_conv2d_k3s2_bf16')}
        ok = dpt.DPTHead.hip_front_ok
        dpt.DPTHead.hip_front_ok = lambda self, rows: False
        okc = dpt._hip_convs_ok
        dpt._hip_convs_ok = lambda *a, **k: False          # PyTorch autocast path
        try:
            amp = m(x)['metric_depth']
        finally:
            dpt.DPTHead.hip_front_ok, dpt._hip_convs_ok = ok, okc
        m.head_dtype = None
        ref32 = m(x)['metric_depth']
    assert ran == {'veon_tokens_to_image': 4, 'veon_conv2d_k3s2_bf16': 1}, ran
    assert got.shape == ref32.shape == (2, 70, 98)
    err_hip = ((got - ref32).norm() / ref32.norm()).item()
    err_amp = ((amp - ref32).norm() / ref32.norm()).item()
    assert err_hip < max(2.0 * err_amp, 2e-2), (err_hip, err_amp)


@pytest.mark.gpu
def test_intermediate_rows_match_reference_taps():
    """The native hand-over of the encoder (patch GEMM + bf16 LayerNormed tap rows)
    against the reference's own tap outputs (golden: get_intermediate_layers with
    norm=True): relative L2 error <= 1e-2 (bf16 operands incl. the image patches)."""
    g = load_golden('dinov2_tiny')
    enc, _ = _build(g)
    enc = enc.to('cuda:0')
    x = torch.from_numpy(g['x']).to('cuda:0')
    taps = [int(t) for t in g['taps']]
    with torch.no_grad():
        rows = enc.intermediate_rows(x, taps)
        tok = enc._native_tokens(x).view(x.shape[0], -1, enc.embed_dim)
        ref_tok = enc.prepare_tokens_with_masks(x)
    assert rows is not None and len(rows) == len(taps)
    assert ((tok - ref_tok).norm() / ref_tok.norm()).item() < 5e-3
    B = x.shape[0]
    for i, r in enumerate(rows):
        assert r.dtype == torch.bfloat16
        r = r.float().view(B, -1, enc.embed_dim).cpu().numpy()
        patch, cls = g['tap_patch'][i], g['tap_cls'][i]
        assert np.linalg.norm(r[:, 1:] - patch) / np.linalg.norm(patch) < 1e-2
        assert np.linalg.norm(r[:, 0] - cls) / np.linalg.norm(cls) < 1e-2


@pytest.mark.gpu
def test_single_image_calls_do_not_corrupt_the_position_cache():
    """with B == 1 the stream used to be a VIEW of the cached
    cls + position rows, so the in-place blocks overwrote the cache and every later
    single-image call started from garbage.  Two B == 1 calls must each equal the
    matching half of the B == 2 call and the reference taps."""
    g = load_golden('dinov2_tiny')
    enc, _ = _build(g)
    enc = enc.to('cuda:0')
    x = torch.from_numpy(g['x']).to('cuda:0')
    taps = [int(t) for t in g['taps']]
    d = enc.embed_dim
    with torch.no_grad():
        both = [r.float().view(2, -1, d) for r in enc.intermediate_rows(x, taps)]
        base_before = enc._pos_cache[('tok', x.shape[2], x.shape[3], x.device)][2].clone()
        first = [r.float().view(1, -1, d) for r in enc.intermediate_rows(x[:1], taps)]
        base_after = enc._pos_cache[('tok', x.shape[2], x.shape[3], x.device)][2]
        assert torch.equal(base_before, base_after), 'cached position rows were overwritten'
        second = [r.float().view(1, -1, d) for r in enc.intermediate_rows(x[1:], taps)]
        again = [r.float().view(1, -1, d) for r in enc.intermediate_rows(x[:1], taps)]
    for i in range(len(taps)):
        assert torch.equal(first[i], again[i])
        # the GEMM tiles see M = T instead of 2T rows, the arithmetic per row is the same
        torch.testing.assert_close(first[i][0], both[i][0], rtol=0, atol=0)
        torch.testing.assert_close(second[i][0], both[i][1], rtol=0, atol=0)
        patch = g['tap_patch'][i]
        got = torch.cat([first[i], second[i]])[:, 1:].cpu().numpy()
        assert np.linalg.norm(got - patch) / np.linalg.norm(patch) < 1e-2

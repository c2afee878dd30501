"""SAN's 2-D mask branch: ``RegionwiseSideAdapterNetwork`` +
``MLPMaskDecoder`` + the hand-over to the CLIP recognition head against a vector made
by the reference's own classes (oracle/tools/gen_golden_side_adapter.py).

Pinned: query / position tokens, bicubic position resize, AddFusion points, mask
decoder, attention-bias hand-over, 2-D semantic inference.  UNPINNED: timm's ViT
block and open_clip's residual block (both absent from the build image) -- the same
restatement runs on both sides of this comparison."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import load_golden
from veon_amd.models.semantic_net import (ClipRecHead, ClipVisualTrunk,
                                          RegionwiseSideAdapterNetwork, semantic_branch_2d)

CFG = dict(clip_width=64, clip_layers=4, clip_heads=2, clip_first_tail=3, clip_proj_dim=24,
           width=48, depth=4, heads=3, queries=5, vit_image=64)


def _sub(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def _build(g):
    W, K = CFG['clip_width'], CFG['clip_first_tail']
    trunk = ClipVisualTrunk(image_size=32, patch_size=16, width=W, layers=CFG['clip_layers'],
                            heads=CFG['clip_heads'])
    trunk.load_state_dict(_sub(g, 'trunk/'), strict=True)
    ln_post = torch.nn.LayerNorm(W)
    ln_post.load_state_dict(_sub(g, 'ln_post/'), strict=True)
    proj = torch.nn.Parameter(torch.from_numpy(g['clip_proj']))
    head = ClipRecHead(trunk.resblocks, ln_post, proj, first_layer_idx=K,
                       sos_token_num=CFG['queries'])
    net = RegionwiseSideAdapterNetwork.build(
        clip_dim=W, image_size=CFG['vit_image'], width=CFG['width'], depth=CFG['depth'],
        num_heads=CFG['heads'], num_queries=CFG['queries'],
        fusion_map=('0->0', '1->1', '2->2', '3->3'), deep_supervision_idxs=(4,),
        attn_heads=CFG['clip_heads'], embed_channels=16, mlp_channels=24, mlp_num_layers=3)
    net.load_state_dict(_sub(g, 'net/'), strict=True)      # reference parameter names
    return trunk.eval(), head.eval(), net.eval()


def _clip_feats(trunk, images, K):
    x = F.interpolate(images, scale_factor=0.5, mode='bilinear', align_corners=False)
    outs, hw = trunk(x, last_layer_idx=K)
    feats = {}
    for i, t in enumerate(outs):
        ClipRecHead._save(feats, i, t, hw)
    return feats


def _close(a, ref, tol=1e-4):
    err = np.abs(a.detach().cpu().float().numpy() - ref).max()
    assert err <= tol * max(1.0, np.abs(ref).max()), err


def test_side_adapter_matches_reference_classes():
    g = load_golden('side_adapter_tiny')
    trunk, head, net = _build(g)
    images = torch.from_numpy(g['images'])
    with torch.no_grad():
        feats = _clip_feats(trunk, images, CFG['clip_first_tail'])
        for i in range(CFG['clip_first_tail'] + 1):
            _close(feats[i], g['clip_feat_%d' % i])
        mask_preds, attn_biases, san_feats = net(images, feats)
    assert len(mask_preds) == 1 and len(attn_biases) == 1 and len(attn_biases[0]) == 1
    _close(mask_preds[0], g['mask_preds'])
    _close(attn_biases[0][0], g['attn_bias'])
    assert len(san_feats) == CFG['depth']
    for i, f in enumerate(san_feats):
        _close(f, g['san_feat_%d' % i])


def test_2d_branch_matches_reference_chain():
    g = load_golden('side_adapter_tiny')
    trunk, head, net = _build(g)
    images = torch.from_numpy(g['images'])
    ov = torch.from_numpy(g['ov_classifier_weight'])
    with torch.no_grad():
        out = semantic_branch_2d(net, head, ov, images,
                                 _clip_feats(trunk, images, CFG['clip_first_tail']))
    for k in ('mask_embs', 'mask_logits', 'sem_seg_ds', 'sem_embed_ds', 'sem_seg'):
        _close(out[k], g[k])


def test_default_build_is_the_san_configuration():
    """configs/san_config.py:58-75 / timm_wrapper.py:67-74: width 240, 8 blocks,
    6 heads, patch 16, 100 queries, CLIP maps 0/3/6/9 fused before blocks 0..3."""
    net = RegionwiseSideAdapterNetwork.build()
    assert net.num_features == 240 and len(net.vit_model.blocks) == 8
    assert net.vit_model.blocks[0].attn.num_heads == 6
    assert net.query_embed.shape == (1, 100, 240)
    assert net.fusion_map == {0: 0, 1: 3, 2: 6, 3: 9}
    assert net.vit_model.pos_embed.shape == (1, 40 * 40, 240)
    assert net.mask_decoder.total_heads == 12


def test_path_exposes_the_2d_branch():
    from veon_amd.models.veon_occ import VeonOccupancyPath
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    kw = dict(input_size=(64, 96), num_cam=2, clip_width=64, clip_layers=4, clip_heads=2,
              clip_first_tail=3, clip_proj_dim=24, embed_dim=64, n_classes=5,
              occ_size=(4, 20, 20), hsa_dim=64, hsa_fusion_map=('0->1->1', '1->2->2'),
              grid_config=grid, bf16_heads=False, two_streams=False, clip_image=32)
    with pytest.raises(RuntimeError):
        VeonOccupancyPath(**kw).forward_2d(torch.zeros(1, 2, 3, 64, 96))
    net = VeonOccupancyPath(side_adapter=dict(
        image_size=64, width=48, depth=4, num_heads=3, num_queries=5,
        fusion_map=('0->0', '1->1', '2->2', '3->3'), deep_supervision_idxs=(4,),
        embed_channels=16, mlp_channels=24), **kw).eval()
    with torch.no_grad():
        out = net.forward_2d(torch.randn(1, 2, 3, 64, 96))
    assert out['mask_preds'].shape == (2, 5, 4, 6)
    assert out['sem_embed_ds'].shape == (2, 24, 4, 6)
    assert out['sem_seg'].shape == (2, 4, 64, 96)
    assert torch.isfinite(out['sem_seg']).all()


@pytest.mark.gpu
def test_side_adapter_blocks_on_the_mfma_kernels():
    """the side-adapter ViT on the HIP kernels (width 48 -> 64, head_dim
    16 -> 64 by zero padding, LayerNorm over the real columns) against the reference
    vector of tests/golden/side_adapter_tiny.npz (the reference's own classes around a
    restated timm block: UNPINNED block, pinned wiring) and against the fp32 PyTorch
    path on the same weights.  Tolerance: bf16 operands with fp32 accumulation through
    4 residual blocks -- 2e-2 of the output range, the CLIP trunk's tolerance."""
    from veon_amd import _lib
    g = load_golden('side_adapter_tiny')
    trunk, head, net = _build(g)
    dev = torch.device('cuda:0')
    net = net.to(dev)
    images = torch.from_numpy(g['images']).to(dev)
    feats = {i: torch.from_numpy(g['clip_feat_%d' % i]).to(dev)
             for i in range(CFG['clip_first_tail'] + 1)}      # the reference's CLIP maps
    with torch.no_grad():
        before = dict(_lib.CALLS)
        mask_preds, attn_biases, san_feats = net(images, feats)
        ran = {k: _lib.CALLS.get(k, 0) - before.get(k, 0)
               for k in ('veon_vit_layernorm_padded', 'veon_vit_attention', 'veon_vit_gemm')}
        net.vit_model.use_hip = False
        ref_preds, ref_biases, ref_feats = net(images, feats)
    depth = CFG['depth']
    assert ran == {'veon_vit_layernorm_padded': 2 * depth, 'veon_vit_attention': depth,
                   'veon_vit_gemm': 4 * depth}, ran
    _close(mask_preds[0], g['mask_preds'], tol=2e-2)
    _close(attn_biases[0][0], g['attn_bias'], tol=2e-2)
    for i, f in enumerate(san_feats):
        _close(f, g['san_feat_%d' % i], tol=2e-2)
        _close(f, ref_feats[i].cpu().numpy(), tol=2e-2)
    _close(mask_preds[0], ref_preds[0].cpu().numpy(), tol=2e-2)


@pytest.mark.gpu
def test_side_adapter_san_width_on_the_mfma_kernels():
    """The SAN configuration itself (width 240, 6 heads of 40, 8 blocks, 100 queries) at
    the bench resolution (256x704 -> 16x44 patches): native blocks against the fp32
    PyTorch blocks on the same random weights, relative L2 <= 2e-2 per tapped map."""
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    net = RegionwiseSideAdapterNetwork.build(clip_dim=768).to(dev).eval()
    images = torch.randn(2, 3, 256, 704, device=dev)
    feats = {i: torch.randn(2, 768, 8, 22, device=dev) for i in (0, 3, 6, 9)}
    with torch.no_grad():
        a_preds, a_bias, a_feats = net(images, feats)
        net.vit_model.use_hip = False
        b_preds, b_bias, b_feats = net(images, feats)
    for x, y in zip(a_feats + a_preds, b_feats + b_preds):
        rel = ((x - y).norm() / y.norm()).item()
        assert rel <= 2e-2, rel
    assert a_bias[0][0].shape == (2, 12, 100, 16, 44)


@pytest.mark.gpu
def test_2d_branch_on_the_gpu():
    """Same chain on cuda:0: the CLIP trunk and the side adapter's blocks run on the
    MFMA kernels there.  Tolerance 3e-2 of the output range (bf16 operands,
    fp32 accumulation, 4 residual blocks)."""
    g = load_golden('side_adapter_tiny')
    trunk, head, net = _build(g)
    dev = torch.device('cuda:0')
    trunk, head, net = trunk.to(dev), head.to(dev), net.to(dev)
    images = torch.from_numpy(g['images']).to(dev)
    ov = torch.from_numpy(g['ov_classifier_weight']).to(dev)
    with torch.no_grad():
        out = semantic_branch_2d(net, head, ov, images,
                                 _clip_feats(trunk, images, CFG['clip_first_tail']))
    for k in ('mask_preds', 'sem_seg_ds', 'sem_embed_ds', 'sem_seg'):
        _close(out[k], g[k], tol=3e-2)


def test_mask_decoder_inference_forms_equal_the_reference_einsum_and_linear():
    """MLPMaskDecoder at inference writes the attention-bias contraction as a batched
    matmul and the Linear(1, 1) bias scaling as a multiply-add (side_adapter.py); both
    must give what the reference's einsum / nn.Linear give (the grad-enabled path)."""
    from veon_amd.models.semantic_net.side_adapter import MLPMaskDecoder
    torch.manual_seed(5)
    dec = MLPMaskDecoder(in_channels=24, total_heads=3, total_layers=2, embed_channels=16,
                         mlp_channels=16, mlp_num_layers=2, rescale_attn_bias=True).eval()
    q, x = torch.randn(2, 7, 24), torch.randn(2, 24, 5, 6)
    m_ref, a_ref = dec(q, x)
    with torch.no_grad():
        m_inf, a_inf = dec(q, x)
    torch.testing.assert_close(m_inf, m_ref.detach(), rtol=1e-6, atol=1e-6)
    assert len(a_inf) == len(a_ref) == 2
    for a, b in zip(a_inf, a_ref):
        assert a.shape == b.shape == (2, 3, 7, 5, 6)
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)

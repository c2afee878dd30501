"""CLIP residual-attention block restatement.  PARITY UNPINNED at the reference
level: the arithmetic lives in third-party open_clip (absent, unpinned version,
SURVEY 8c) and the reference holds no vectors for it.  What is checked:
(CPU) the block equals a hand-written nn.MultiheadAttention-semantics formula;
(GPU) the MFMA path matches the torch fp32 path within bf16 tolerance, with and
without the dense additive attn_mask of update_remaining_clip_feats."""
import pytest
import torch

from veon_amd.models.semantic_net import clip_blocks


def _manual_block(blk, x, mask=None):
    L, N, D = x.shape
    H = blk.n_head
    h = torch.nn.functional.layer_norm(x, (D,), blk.ln_1.weight, blk.ln_1.bias, blk.ln_1.eps)
    qkv = h @ blk.attn.in_proj_weight.t() + blk.attn.in_proj_bias
    q, k, v = qkv.view(L, N, 3, H, D // H).permute(2, 1, 3, 0, 4)   # N,H,L,hd
    s = (q * (D // H) ** -0.5) @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask.view(N, H, L, L)
    o = (s.softmax(-1) @ v).permute(2, 0, 1, 3).reshape(L, N, D)
    x = x + o @ blk.attn.out_proj.weight.t() + blk.attn.out_proj.bias
    h = torch.nn.functional.layer_norm(x, (D,), blk.ln_2.weight, blk.ln_2.bias, blk.ln_2.eps)
    u = h @ blk.mlp.c_fc.weight.t() + blk.mlp.c_fc.bias
    u = u * torch.sigmoid(1.702 * u) if blk.quick_gelu else torch.nn.functional.gelu(u)
    return x + u @ blk.mlp.c_proj.weight.t() + blk.mlp.c_proj.bias


@pytest.mark.parametrize('quick', [True, False])
def test_block_matches_mha_semantics_cpu(quick):
    torch.manual_seed(0)
    blk = clip_blocks.ResidualAttentionBlock(128, 2, quick_gelu=quick).eval()
    x = torch.randn(9, 3, 128)
    mask = torch.randn(3 * 2, 9, 9)
    with torch.no_grad():
        torch.testing.assert_close(blk(x), _manual_block(blk, x), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(blk(x, attn_mask=mask), _manual_block(blk, x, mask),
                                   rtol=1e-4, atol=1e-5)


def test_trunk_names_and_shapes_cpu():
    m = clip_blocks.ClipVisualTrunk(image_size=64, patch_size=16, width=64, layers=2,
                                    heads=1).eval()
    keys = set(m.state_dict())
    for k in ('conv1.weight', 'class_embedding', 'positional_embedding',
              'ln_pre.weight', 'resblocks.0.attn.in_proj_weight',
              'resblocks.0.attn.out_proj.bias', 'resblocks.1.mlp.c_fc.weight',
              'resblocks.1.mlp.c_proj.bias', 'resblocks.0.ln_2.bias'):
        assert k in keys, k
    with torch.no_grad():
        outs, hw = m(torch.randn(2, 3, 32, 48))
    assert hw == (2, 3) and len(outs) == 3 and outs[-1].shape == (7, 2, 64)


@pytest.mark.gpu
def test_clip_b16_blocks_mfma_vs_torch():
    """ViT-B/16 geometry at 128x352 (the x0.5 image of a 256x704 crop):
    8x22+1 = 177 tokens, 6 images, 3 blocks; then a tail pass with a dense
    additive mask."""
    torch.manual_seed(0)
    dev = 'cuda:0'
    m = clip_blocks.ClipVisualTrunk(224, 16, 768, 3, 12).to(dev).eval()
    img = torch.randn(6, 3, 128, 352, device=dev)
    with torch.no_grad():
        outs, hw = m(img)
        t0 = outs[0]
        ref = t0
        for blk in m.resblocks:
            ref = blk(ref)
        rel = (outs[-1] - ref).norm() / ref.norm()
        assert hw == (8, 22) and outs[-1].shape == (177, 6, 768)
        assert rel < 1e-2, rel.item()
        L = t0.shape[0]
        mask = torch.randn(6, 12, L, L, device=dev)
        masks = [mask.reshape(-1, L, L)] * 3
        got = clip_blocks.run_blocks(list(m.resblocks), t0, masks)[-1]
        ref = t0
        for blk in m.resblocks:
            ref = blk(ref, attn_mask=masks[0])
        rel = (got - ref).norm() / ref.norm()
        assert rel < 1e-2, rel.item()


def test_cross_attn_with_self_bias_equals_augmented_softmax():
    """Independent formulation: append every query to its own key/value list
    (one extra key per query) and run an ordinary masked softmax attention."""
    torch.manual_seed(1)
    D, H, K, L, N = 128, 2, 5, 11, 3
    blk = clip_blocks.ResidualAttentionBlock(D, H).eval()
    x, mem = torch.randn(K, N, D), torch.randn(L, N, D)
    bias = torch.randn(N * H, K, L)
    with torch.no_grad():
        got = clip_blocks.cross_attn_layer(blk, x, mem, bias)
        a = blk.attn
        hd = D // H
        qx, mx = blk.ln_1(x), blk.ln_1(mem)
        w, b = a.in_proj_weight, a.in_proj_bias
        ref_rows = []
        for n in range(N):
            per_q = []
            for kq in range(K):
                keys = torch.cat([mx[:, n], qx[kq:kq + 1, n]], 0)      # L+1 tokens
                q = (qx[kq, n] @ w[:D].t() + b[:D]).view(H, hd) * hd ** -0.5
                kk = (keys @ w[D:2 * D].t() + b[D:2 * D]).view(L + 1, H, hd)
                vv = (keys @ w[2 * D:].t() + b[2 * D:]).view(L + 1, H, hd)
                logit = torch.einsum('hd,lhd->hl', q, kk)
                logit[:, :L] += bias.view(N, H, K, L)[n, :, kq]
                o = torch.einsum('hl,lhd->hd', logit.softmax(-1), vv).reshape(D)
                per_q.append(o @ a.out_proj.weight.t() + a.out_proj.bias)
            ref_rows.append(torch.stack(per_q))
        attn_out = torch.stack(ref_rows, 1)                             # K,N,D
        ref = x + attn_out
        ref = ref + blk.mlp(blk.ln_2(ref))
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


def test_cross_attention_with_self_logit_matches_reference_vectors():
    """cross_attn_with_self_bias vs the reference's own attn_helper.py run on a
    seeded nn.MultiheadAttention (oracle/tools/gen_golden_clip_attn.py): float
    bias, no mask and boolean mask."""
    from tests.conftest import load_golden
    from veon_amd.models.semantic_net.clip_blocks import cross_attn_with_self_bias
    g = load_golden('clip_cross_attn')
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    mha = torch.nn.MultiheadAttention(64, 4).eval()
    mha.load_state_dict({k[4:]: v for k, v in t.items() if k.startswith('mha/')})
    with torch.no_grad():
        for mask, key in ((t['bias'], 'out_bias'), (None, 'out_none'),
                          (t['bool_mask'], 'out_bool')):
            got = cross_attn_with_self_bias(mha, t['q'], t['mem'], t['mem'], attn_mask=mask)
            assert torch.allclose(got, t[key], atol=2e-6), key


def test_trunk_wiring_matches_reference_feature_extractor():
    """ClipVisualTrunk vs the reference's own FeatureExtractor.forward
    (clip_utils/visual.py:57-91, run by oracle/tools/gen_golden_clip_trunk.py
    around the same sub-modules): patchify, class token, position-embedding
    resize (4x4 -> 2x3), ln_pre, LND layout and the per-block outputs, in
    training-free eval mode for both the conv and the GEMM patchify."""
    from tests.conftest import load_golden
    from veon_amd.models.semantic_net import ClipVisualTrunk
    g = load_golden('clip_trunk_tiny')
    trunk = ClipVisualTrunk(image_size=64, patch_size=16, width=64, layers=2, heads=1).eval()
    trunk.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                           if k.startswith('sd/')})
    x = torch.from_numpy(g['x'])
    for ctx in (torch.no_grad, torch.enable_grad):   # GEMM patchify / conv1 patchify
        with ctx():
            outs, hw = trunk(x)
        assert tuple(hw) == tuple(int(v) for v in g['hw'])
        for i in range(3):
            t = outs[i].detach()                      # (L, N, D)
            n, c = t.shape[1], t.shape[2]
            feat = t[1:].permute(1, 2, 0).reshape(n, c, *hw)
            assert torch.allclose(feat, torch.from_numpy(g['feat_%d' % i]), atol=2e-5), i
            assert torch.allclose(t[0:1], torch.from_numpy(g['cls_%d' % i]), atol=2e-5), i


def _head_from_golden(g, device='cpu'):
    from veon_amd.models.semantic_net import ClipRecHead
    from veon_amd.models.semantic_net.clip_blocks import ResidualAttentionBlock
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    blocks = torch.nn.ModuleList([ResidualAttentionBlock(64, 1) for _ in range(5)])
    blocks.load_state_dict({k[7:]: v for k, v in t.items() if k.startswith('blocks/')})
    ln_post = torch.nn.LayerNorm(64)
    ln_post.load_state_dict({k[8:]: v for k, v in t.items() if k.startswith('ln_post/')})
    head = ClipRecHead(blocks, ln_post, torch.nn.Parameter(t['proj']),
                       first_layer_idx=int(t['first']), sos_token_num=3)
    return head.to(device).eval(), {k: v.to(device) for k, v in t.items()}


def test_rec_head_matches_reference_vectors():
    """ClipRecHead vs the reference's own RecWithAttnbiasHead
    (clip_utils/visual.py:112-292, oracle/tools/gen_golden_clip_head.py): the
    SOS-token forward with cross attention, and update_remaining_clip_feats with
    offsets + dense attention biases."""
    from tests.conftest import load_golden
    g = load_golden('clip_head_tiny')
    head, t = _head_from_golden(g)
    first = int(t['first'])
    feats = {first: t['feat'], '%d_cls_token' % first: t['cls']}
    with torch.no_grad():
        sos = head(feats, [t['attn_bias']], normalize=True)
        assert torch.allclose(sos, t['sos'], atol=2e-5)
        outs = {first: t['feat'].clone(), '%d_cls_token' % first: t['cls'].clone()}
        attns = [t['attn_%d' % i] for i in range(3)]
        head.update_remaining_clip_feats(outs, t['offsets'], attns)
    for i in range(first + 1, 6):
        assert torch.allclose(outs[i], t['out_%d' % i], atol=5e-5), i
        assert torch.allclose(outs['%d_cls_token' % i], t['out_cls_%d' % i], atol=5e-5), i
    assert torch.allclose(outs['clip_feat_proj'], t['clip_feat_proj'], atol=5e-5)


@pytest.mark.gpu
def test_rec_head_tail_blocks_on_mfma():
    """update_remaining_clip_feats on a ROCm device: the tail blocks (with their
    dense masks) run through veon_vit_block; result within bf16 tolerance of the
    reference vectors."""
    from tests.conftest import load_golden
    from veon_amd import _lib
    g = load_golden('clip_head_tiny')
    head, t = _head_from_golden(g, 'cuda:0')
    first = int(t['first'])
    outs = {first: t['feat'].clone(), '%d_cls_token' % first: t['cls'].clone()}
    attns = [t['attn_%d' % i] for i in range(3)]
    before = _lib.CALLS.get('veon_vit_block', 0)
    with torch.no_grad():
        head.update_remaining_clip_feats(outs, t['offsets'], attns)
    assert _lib.CALLS.get('veon_vit_block', 0) - before == 3
    want = t['clip_feat_proj']
    rel = ((outs['clip_feat_proj'] - want).norm() / want.norm()).item()
    assert rel < 2e-2, rel


@pytest.mark.gpu
def test_trunk_native_token_stream_equals_torch_tokens():
    """ClipVisualTrunk._native_stream (patchify kernel + one MFMA GEMM onto the position
    rows + fp32 LayerNorm kernel) against ``tokens`` (conv1 as an fp32 matmul, cat, add,
    nn.LayerNorm): the same stream up to the half-precision rounding of the patch operands;
    entry 0 of the trunk's outputs is that stream, and the blocks after it agree with the
    torch blocks started from it.  Also with a resized position embedding and an image
    whose size is not a multiple of the patch."""
    torch.manual_seed(1)
    dev = 'cuda:0'
    m = clip_blocks.ClipVisualTrunk(224, 16, 768, 2, 12).to(dev).eval()
    for shape in ((6, 3, 128, 352), (2, 3, 100, 70)):
        img = torch.randn(*shape, device=dev)
        with torch.no_grad():
            want, hw = m.tokens(img)                       # (L, N, D)
            s, hw2 = m._native_stream(img)                 # (N, L, D)
            assert hw == hw2
            got = s.permute(1, 0, 2)
            rel = (got - want).norm() / want.norm()
            assert rel < 4e-3, rel.item()
            outs, _ = m(img, taps={0, 2})
            assert torch.equal(outs[0], got)
            assert outs[1] is None and outs[2] is not None
            ref = want
            for blk in m.resblocks:
                ref = blk(ref)
            assert (outs[2] - ref).norm() / ref.norm() < 1e-2


def test_padded_attention_biases_equal_the_bordered_gram_matrices():
    """AttnManipulateBlock.pad_class_token: the Gram matrices of the head embeddings with a
    zero row in front ARE what ClipRecHead.build_attn_bias makes of the unpadded ones
    (clip_utils/visual.py:287-292), and update_remaining_clip_feats passes them through."""
    from veon_amd.models.semantic_net.hsa_network import AttnManipulateBlock
    torch.manual_seed(3)
    blk = AttnManipulateBlock(dim=64, mlp_dim=64, clip_dim=128, heads=2, dim_head=8,
                              attn_layers=3, add_layers=1, supp_dim=32).eval()
    x = torch.randn(2, 6 * 10, 64)
    with torch.no_grad():
        _, plain, supp = blk(x, (6, 10), (3, 5))
        blk.pad_class_token = True
        _, padded, supp2 = blk(x, (6, 10), (3, 5))
    assert plain.shape == (3, 2, 2, 15, 15) and padded.shape == (3, 2, 2, 16, 16)
    assert torch.equal(supp, supp2)
    for t in range(3):
        want = clip_blocks.ClipRecHead.build_attn_bias(plain[t])       # (B*H, L+1, L+1)
        torch.testing.assert_close(padded[t].reshape(-1, 16, 16), want, rtol=1e-6, atol=1e-6)
        assert float(padded[t][..., 0, :].abs().sum()) == 0.0
        assert float(padded[t][..., :, 0].abs().sum()) == 0.0

"""ViT block kernels (bf16 MFMA) against plain PyTorch fp32 references of the
same ops on the same (bf16-rounded) operands.  Tolerances are stated per test:
the products are exact in fp32 accumulation, so the error budget is the bf16
rounding of the OUTPUT (2^-9 relative) plus accumulation-order noise."""
import pytest
import torch

from veon_amd import vit_ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_cast_bf16_round_to_nearest_even():
    x = _rand(1000, 37, seed=1)
    got = vit_ops.to_bf16(x)
    assert torch.equal(got, x.to(torch.bfloat16))


@pytest.mark.parametrize('T,d', [(901, 768), (17, 1024), (5, 384), (3, 100)])
def test_layernorm(T, d):
    x = _rand(T, d, seed=2, scale=3.0) + 0.5
    w = _rand(d, seed=3) * 0.1 + 1.0
    b = _rand(d, seed=4) * 0.1
    got = vit_ops.layernorm(x, w, b, eps=1e-6).float()
    ref = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-6)
    # output is bf16: half-ulp relative error 2^-9, plus fp32 reduction noise
    torch.testing.assert_close(got, ref, rtol=2 ** -8, atol=2e-3)


@pytest.mark.parametrize('M,N,K', [(901, 768, 768), (5406, 2304, 768),
                                   (130, 3072, 768), (64, 128, 64), (1, 4, 64),
                                   (300, 1024, 4096)])
def test_gemm_bias_bf16(M, N, K):
    a = _rand(M, K, seed=5).to(torch.bfloat16)
    w = (_rand(N, K, seed=6) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=7)
    got = vit_ops.linear(a, w, bias).float()
    ref = a.float() @ w.float().t() + bias
    torch.testing.assert_close(got, ref, rtol=2 ** -8, atol=2e-3)
    got_nb = vit_ops.linear(a, w, None).float()
    torch.testing.assert_close(got_nb, a.float() @ w.float().t(),
                               rtol=2 ** -8, atol=2e-3)


@pytest.mark.parametrize('cfg', [1, 2, 5, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize('M,N,K', [(5406, 2304, 768), (300, 1024, 4096), (901, 768, 192),
                                   (257, 520, 64), (1, 4, 64)])
def test_gemm_every_big_tile_configuration(cfg, M, N, K):
    """The DMA-ring kernels (1..7: whole K = 64 stages, eight waves; 8..10: the same with
    sixteen waves; 11..14: ring of K = 32 granules,
    four slots, three in flight) forced through veon_gemm_ring_set, on ragged shapes
    (M, N not multiples of the tile, K of one to 64 stages): all three epilogue
    families against fp32 references."""
    from veon_amd import _lib
    a = _rand(M, K, seed=21).to(torch.bfloat16)
    w = (_rand(N, K, seed=22) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=23)
    gamma = _rand(N, seed=24) * 0.1
    x = _rand(M, N, seed=25)
    pre = a.float() @ w.float().t() + bias
    L = _lib.lib()
    L.veon_gemm_ring_set(cfg)
    try:
        got = vit_ops.linear(a, w, bias).float()
        gelu = vit_ops.linear(a, w, bias, vit_ops.EPI_GELU).float()
        res = vit_ops.linear_residual_(x.clone(), a, w, bias, gamma)
    finally:
        L.veon_gemm_ring_set(-1)
    torch.testing.assert_close(got, pre, rtol=2 ** -8, atol=2e-3)
    torch.testing.assert_close(gelu, torch.nn.functional.gelu(pre), rtol=2 ** -8, atol=2e-3)
    torch.testing.assert_close(res, x + gamma * pre, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('M,N,K', [(5406, 768, 3072), (5406, 1024, 4096), (4200, 768, 2048)])
def test_gemm_splitk_residual(M, N, K):
    """fc2-shaped residual GEMM with K split over two workgroups per tile (slab + ticket
    + agent-scope release / acquire): against the fp32 reference, against the unsplit
    kernel within accumulation-order noise, bit-identical across repeated launches (the
    result must not depend on which half arrives first) and with the sync words left
    zero."""
    from veon_amd import _lib
    L = _lib.lib()
    a = _rand(M, K, seed=31).to(torch.bfloat16)
    w = (_rand(N, K, seed=32) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=33)
    gamma = _rand(N, seed=34) * 0.1
    x = _rand(M, N, seed=35)
    need = L.veon_vit_gemm_splitk_plan(M, N, K, None)
    assert need > 0
    slab = torch.empty(need, dtype=torch.uint8, device=DEV)
    sync = torch.zeros(1024, dtype=torch.int32, device=DEV)
    ref = x + gamma * (a.float() @ w.float().t() + bias)
    outs = []
    for _ in range(4):
        slab.fill_(255)     # stale slab contents must not matter
        outs.append(vit_ops.linear_residual_splitk_(x.clone(), a, w, bias, gamma,
                                                    (slab, sync)))
        assert int(sync.abs().sum()) == 0
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=1e-4)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    plain = vit_ops.linear_residual_(x.clone(), a, w, bias, gamma)
    torch.testing.assert_close(outs[0], plain, rtol=1e-5, atol=2e-5)
    assert L.veon_vit_gemm_splitk_plan(901, 768, 768, None) == 0      # short K: not split


def test_gemm_gelu_and_quickgelu():
    M, N, K = 901, 3072, 768
    a = _rand(M, K, seed=8).to(torch.bfloat16)
    w = (_rand(N, K, seed=9) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=10)
    pre = a.float() @ w.float().t() + bias
    got = vit_ops.linear(a, w, bias, vit_ops.EPI_GELU).float()
    torch.testing.assert_close(got, torch.nn.functional.gelu(pre),
                               rtol=2 ** -8, atol=2e-3)
    got = vit_ops.linear(a, w, bias, vit_ops.EPI_QUICKGELU).float()
    torch.testing.assert_close(got, pre * torch.sigmoid(1.702 * pre),
                               rtol=2 ** -8, atol=2e-3)


def test_gemm_layerscale_residual_inplace():
    M, N, K = 901, 768, 3072
    a = _rand(M, K, seed=11).to(torch.bfloat16)
    w = (_rand(N, K, seed=12) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=13)
    gamma = _rand(N, seed=14) * 0.1
    x = _rand(M, N, seed=15)
    ref = x + gamma * (a.float() @ w.float().t() + bias)
    got = vit_ops.linear_residual_(x.clone(), a, w, bias, gamma)
    # fp32 output: only accumulation-order noise
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    got2 = vit_ops.linear_residual_(x.clone(), a, w, None, None)
    torch.testing.assert_close(got2, x + a.float() @ w.float().t(),
                               rtol=1e-4, atol=1e-4)


def _ref_attention(qkv, H, bias=None, q_log2=False, dtype=torch.float32):
    B, T, _ = qkv.shape
    q, k, v = qkv.to(dtype).view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    if q_log2:
        q = q / vit_ops.LOG2E
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias.to(dtype)
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)


def _qkv(B, T, H, seed, scale=0.5, q_log2=False):
    """Packed qkv as the projection GEMM leaves it; ``q_log2``: q also carries log2(e)
    (the packers fold it into the weights), the reference takes it out again."""
    x = _rand(B, T, 3 * H * 64, seed=seed) * scale
    if q_log2:
        x.view(B, T, 3, H * 64)[:, :, 0] *= vit_ops.LOG2E
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('q_log2', [False, True])
@pytest.mark.parametrize('B,T,H', [(2, 901, 12), (1, 64, 1), (1, 65, 2),
                                   (3, 17, 3), (1, 705, 12), (1, 300, 16)])
def test_attention(B, T, H, q_log2):
    qkv = _qkv(B, T, H, 16, q_log2=q_log2)
    got = vit_ops.attention(qkv, H, q_log2=q_log2).float()
    ref = _ref_attention(qkv, H, q_log2=q_log2)
    # P is rounded to bf16 before P.V (2^-9 relative on each weight), output bf16
    torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=4e-3)


@pytest.mark.parametrize('q_log2', [False, True])
def test_attention_with_bias_and_masking(q_log2):
    B, T, H = 2, 130, 4
    qkv = _qkv(B, T, H, 17, q_log2=q_log2)
    bias = _rand(B, H, T, T, seed=18)
    bias[:, :, :, 100:] = float('-inf')     # masked keys (attn_mask style)
    got = vit_ops.attention(qkv, H, bias, q_log2=q_log2).float()
    ref = _ref_attention(qkv, H, bias, q_log2)
    torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=4e-3)
    # head-broadcast bias
    b1 = bias[:1, :1].contiguous()
    got = vit_ops.attention(qkv, H, b1, q_log2=q_log2).float()
    torch.testing.assert_close(got, _ref_attention(qkv, H, b1, q_log2), rtol=2 ** -7,
                               atol=4e-3)
    # a whole 64-key tile masked in front of live keys (its row maximum is -inf)
    b2 = _rand(B, H, T, T, seed=19)
    b2[:, :, :, :64] = float('-inf')
    got = vit_ops.attention(qkv, H, b2, q_log2=q_log2).float()
    torch.testing.assert_close(got, _ref_attention(qkv, H, b2, q_log2), rtol=2 ** -7,
                               atol=4e-3)


@pytest.mark.parametrize('q_log2', [False, True])
@pytest.mark.parametrize('with_bias', [False, True])
def test_attention_reference_maximum_moves(q_log2, with_bias):
    """The kernel forms its scores relative to a per-query reference maximum that is set
    by tile 0 and then moves only when a tile stands more than 8 (log2 units) above it:
    a rare, data-dependent branch that bounded random data never takes.  Force it
    (cdna_hip_programming.md rule 26): spike chosen keys against chosen queries so that
    the maximum jumps at tiles 1, 3, the last full tile and the ragged tail tile, by
    amounts on both sides of the threshold; make tile 0 itself sit far below zero for
    some queries (the reference has to move DOWN there) and far above for others.
    Full-tensor fp64 reference."""
    B, T, H = 2, 333, 3            # six 64-key tiles, the last with 13 keys
    x = _rand(B, T, 3 * H * 64, seed=41) * 0.4
    v = x.view(B, T, 3, H, 64)
    q, k = v[:, :, 0], v[:, :, 1]
    unit = torch.zeros(64, device=DEV)
    unit[5] = 1.0
    # queries 0..95 get a large component along e5; keys pick it up with chosen gains
    q[:, :96, :, :] = q[:, :96, :, :] * 0.2 + 3.0 * unit
    gains = {70: 2.0, 200: 4.5, 300: 7.0, 328: 9.5}     # key -> extra score / 3.0
    for key, g in gains.items():
        k[:, key, :, :] = k[:, key, :, :] * 0.2 + g * unit
    # queries 100..131: every score of tile 0 far below zero, later tiles near zero
    k[:, :64, 1, :] -= 0.0
    q[:, 100:132, 1, :] = 2.0 * unit
    k[:, :64, 1, 5] = -40.0
    # queries 140..171: tile 0 far above zero
    q[:, 140:172, 2, :] = 2.0 * unit
    k[:, :64, 2, 5] = 30.0
    if q_log2:
        q *= vit_ops.LOG2E
    qkv = x.to(torch.bfloat16)
    bias = _rand(B, H, T, T, seed=42) * 3 if with_bias else None
    got = vit_ops.attention(qkv, H, bias, q_log2=q_log2).double()
    ref = _ref_attention(qkv, H, bias, q_log2, dtype=torch.float64)
    # the branch really is exercised: the running maximum of the spiked queries rises by
    # more than the threshold at several tiles
    qq, kk = qkv.double().view(B, T, 3, H, 64)[:, :, 0], qkv.double().view(B, T, 3, H, 64)[:, :, 1]
    sc = torch.einsum('bqhd,bkhd->bhqk', qq, kk) * (1.0 if q_log2 else vit_ops.LOG2E)
    if bias is not None:
        sc = sc + bias.double() * vit_ops.LOG2E
    tmax = torch.stack([sc[..., i:i + 64].amax(-1) for i in range(0, T, 64)], -1)
    rises = ((tmax[..., 1:] - tmax.cummax(-1).values[..., :-1]) > 8).sum().item()
    assert rises > 500, rises
    assert (tmax[..., 0] < -60).any() and (tmax[..., 0] > 60).any()
    torch.testing.assert_close(got, ref, rtol=2 ** -6, atol=6e-3)


def test_attention_is_deterministic_under_load():
    """Regression for a fault seen while the exp2-domain kernel was built: with the move
    of the reference written as scalar code, hipcc packed it into v_pk_add_f32 with
    operand crossing (op_sel:[0,1]), and on a full chip (B6 T901 H12, 576 workgroups)
    the low halves in lanes 48-63 now and then kept the unshifted score -- a wrong
    softmax weight for one key of one query, in EVERY launch of this input, different
    queries each time.  The kernel is deterministic by construction, so: same input,
    many launches, bit-identical outputs, and correct against fp32."""
    B, T, H = 6, 901, 12
    qkv = _qkv(B, T, H, 43, scale=0.9, q_log2=True)      # reference moves for ~20 % of queries
    outs = torch.empty(B, T, H * 64, dtype=qkv.dtype, device=DEV)
    first = vit_ops.attention(qkv, H, q_log2=True).clone()
    for _ in range(60):
        vit_ops.attention(qkv, H, out=outs, q_log2=True)
        assert torch.equal(outs, first)
    ref = _ref_attention(qkv, H, q_log2=True)
    err = (first.float() - ref).norm(dim=-1) / ref.norm(dim=-1)
    assert err.max().item() < 0.02, err.max().item()


@pytest.mark.parametrize('act,with_gamma,with_bias,q_log2',
                         [(vit_ops.EPI_GELU, True, False, True),
                          (vit_ops.EPI_QUICKGELU, False, True, False),
                          (vit_ops.EPI_QUICKGELU, False, True, True)])
def test_whole_block_call_equals_the_seven_ops(act, with_gamma, with_bias, q_log2):
    """veon_vit_block (one native call per block) is exactly the sequence
    LN -> qkv -> attention -> proj(+res) -> LN -> fc1(act) -> fc2(+res)."""
    torch.manual_seed(5)
    B, T, H = 2, 77, 2
    d, mlp = H * 64, 4 * H * 64
    dev = DEV

    def vec(n, s=1.0):
        return (torch.randn(n, device=dev) * s).contiguous()

    def mat(n, k):
        return vit_ops.to_bf16(torch.randn(n, k, device=dev) * k ** -0.5)
    n1 = (vec(d, 0.1) + 1, vec(d, 0.1), 1e-6)
    n2 = (vec(d, 0.1) + 1, vec(d, 0.1), 1e-5)
    w_qkv, b_qkv, w_proj, b_proj = mat(3 * d, d), vec(3 * d, 0.1), mat(d, d), vec(d, 0.1)
    w_fc1, b_fc1, w_fc2, b_fc2 = mat(mlp, d), vec(mlp, 0.1), mat(d, mlp), vec(d, 0.1)
    g1 = vec(d, 0.5) if with_gamma else None
    g2 = vec(d, 0.5) if with_gamma else None
    bias = torch.randn(B, H, T, T, device=dev) if with_bias else None
    x0 = torch.randn(B * T, d, device=dev)
    # reference: the individual ops
    x = x0.clone()
    h = vit_ops.layernorm(x, *n1)
    qkv = vit_ops.linear(h, w_qkv, b_qkv)
    o = vit_ops.attention(qkv.view(B, T, -1), H, bias, q_log2=q_log2)
    vit_ops.linear_residual_(x, o.view(B * T, -1), w_proj, b_proj, g1)
    h = vit_ops.layernorm(x, *n2)
    u = vit_ops.linear(h, w_fc1, b_fc1, act)
    vit_ops.linear_residual_(x, u, w_fc2, b_fc2, g2)
    # one call
    w = vit_ops.BlockWeights(H, n1, w_qkv, b_qkv, w_proj, b_proj, g1, n2, w_fc1,
                             b_fc1, w_fc2, b_fc2, g2, act, q_log2=q_log2)
    ws = vit_ops.block_workspace(B, T, d, mlp, dev)
    y = vit_ops.block_forward_(x0.clone(), w, B, T, ws, bias)
    assert torch.equal(x, y)
    # too-small workspace is refused, not overrun
    with pytest.raises(Exception):
        vit_ops.block_forward_(x0.clone(), w, B, T, ws[:ws.numel() // 2], bias)


@pytest.mark.parametrize('B,L,d,YX,hw', [(2, 64 * 176, 384, (64, 176), (32, 88)),
                                         (1, 5 * 7 + 3, 128, (5, 7), (3, 3)),
                                         (3, 6 * 9, 256, (6, 9), (7, 11))])
def test_layernorm_f32_add_nearest_is_the_torch_sequence(B, L, d, YX, hw):
    """The fused tail of HighresSideAdaptorBlock (nearest resize of the projected CLIP map,
    add onto the last Y*X tokens, ln_4) against the reference's own sequence
    (highres_side_adaptor.py:123-135): interpolate -> reshape / permute -> cat -> LayerNorm;
    bit-identical to the plain LayerNorm kernel on the torch-composed input (the fused
    kernel adds in fp32 before the same statistics), up- and down-sampling, with
    leading tokens that get no offset."""
    import torch.nn.functional as F
    (Y, X), (h, w) = YX, hw
    g = torch.Generator().manual_seed(L + d)
    x = (torch.randn(B, L, d, generator=g) * 2).to(DEV)
    add = torch.randn(B, h * w, d, generator=g).to(DEV)
    wt = (torch.rand(d, generator=g) + 0.5).to(DEV)
    bs = torch.randn(d, generator=g).to(DEV)
    off = F.interpolate(add.permute(0, 2, 1).reshape(B, d, h, w), size=(Y, X))
    off = off.reshape(B, d, -1).permute(0, 2, 1)
    composed = torch.cat([x[:, :-off.shape[1]], x[:, -off.shape[1]:] + off], 1)
    got = vit_ops.layernorm_f32_add_nearest(x, add, (Y, X), (h, w), wt, bs, 1e-5)
    assert torch.equal(got, vit_ops.layernorm_f32(composed.contiguous(), wt, bs, 1e-5))
    torch.testing.assert_close(got, F.layer_norm(composed, (d,), wt, bs, 1e-5),
                               rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('T,d', [(1000, 384), (7, 128), (67584, 384), (33, 1024)])
def test_layernorm_f32_matches_torch(T, d):
    """veon_layernorm_f32 (fp32 in / out, half a wave per row) vs F.layer_norm."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(d + T)
    x = (torch.randn(T, d, generator=g) * 3 + 1).to('cuda:0')
    w = (torch.rand(d, generator=g) + 0.5).to('cuda:0')
    b = torch.randn(d, generator=g).to('cuda:0')
    got = vit_ops.layernorm_f32(x, w, b, 1e-5)
    want = F.layer_norm(x, (d,), w, b, 1e-5)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('M,N,K,epi', [(5406, 2304, 768, vit_ops.EPI_BF16),      # 16-wave ring tile
                                       (5406, 3072, 768, vit_ops.EPI_GELU),      # small-tile kernel
                                       (5406, 1024, 4096, 'resid')])             # 16-wave, residual
def test_gemm_is_deterministic_under_load(M, N, K, epi):
    """Same operands, 40 launches on a full chip, bit-identical outputs: the GEMM kernels
    accumulate in a fixed order, so any difference would be a hardware / scheduling fault
    of the kind DESIGN 4b describes for the attention kernel."""
    a = _rand(M, K, seed=51).to(torch.bfloat16)
    w = (_rand(N, K, seed=52) * K ** -0.5).to(torch.bfloat16)
    bias = _rand(N, seed=53)
    x0 = _rand(M, N, seed=54)

    def run():
        if epi == 'resid':
            return vit_ops.linear_residual_(x0.clone(), a, w, bias, None)
        return vit_ops.linear(a, w, bias, epi)
    first = run().clone()
    for _ in range(40):
        assert torch.equal(run(), first)

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
                                            (1, 256, 4, 1, 1, 1), (1, 64, 2, 8, 13, 10)])
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

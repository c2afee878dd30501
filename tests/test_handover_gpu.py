"""Direct checks of the small hand-over kernels between the stages:
patch embedding as GEMM rows, token rows -> padded image with the pixel shuffle of a
stride = kernel transposed convolution, strided sampling of a padded image."""
import pytest
import torch
import torch.nn.functional as F

from veon_amd import conv3d_ops, vit_ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('B,C,H,W,p,skip', [(2, 3, 28, 42, 14, 1), (1, 3, 64, 176, 8, 0),
                                            (2, 3, 30, 45, 14, 1)])   # remainder not read
def test_patchify_rows_are_the_conv_operand(B, C, H, W, p, skip):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=DEV)
    rows = vit_ops.patchify(x, p, skip)
    h, w = H // p, W // p
    k = C * p * p
    kpad = (k + 63) // 64 * 64
    assert rows.shape == (B * (skip + h * w), kpad) and rows.dtype == torch.bfloat16
    r = rows.view(B, skip + h * w, kpad)
    assert not r[:, :skip].any() and not r[..., k:].any()
    want = x[:, :, :h * p, :w * p].reshape(B, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5) \
        .reshape(B, h * w, k).bfloat16()
    assert torch.equal(r[:, skip:, :k], want)
    # as a GEMM operand it reproduces the convolution (bf16 operands, fp32 accumulation)
    conv = torch.nn.Conv2d(C, 64, p, p).to(DEV)
    wp = torch.zeros(64, kpad, device=DEV)
    wp[:, :k] = conv.weight.detach().view(64, k)
    out = torch.zeros(B * (skip + h * w), 64, device=DEV)
    vit_ops.linear_residual_(out, rows, vit_ops.to_bf16(wp), conv.bias.detach().float())
    ref = F.conv2d(x.bfloat16().float(), conv.weight.bfloat16().float(), conv.bias, stride=p)
    got = out.view(B, skip + h * w, 64)[:, skip:].permute(0, 2, 1).reshape(B, 64, h, w)
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('s', [1, 2, 4])
def test_tokens_to_image_is_the_transposed_conv_pixel_shuffle(s):
    torch.manual_seed(1)
    B, h, w, cin, c = 2, 3, 5, 64, 16
    T = 1 + h * w
    tok = torch.randn(B * T, cin, device=DEV).bfloat16()
    ct = torch.nn.ConvTranspose2d(cin, c, s, s).to(DEV)
    # GEMM weight: row (i*s + j)*c + co, column ci  (dpt.DPTHead._front_weights)
    w2 = ct.weight.detach().float().permute(2, 3, 1, 0).reshape(s * s * c, cin)
    b2 = ct.bias.detach().float().repeat(s * s)
    rows = vit_ops.linear(tok, vit_ops.to_bf16(w2), b2, vit_ops.EPI_BF16)
    img = conv3d_ops.PaddedImage(B, c, s * h, s * w, DEV)
    conv3d_ops.tokens_to_image(rows, T, 1, h, w, s, c, img)
    got = conv3d_ops.unpack_image(img, torch.float32)
    x = tok.float().view(B, T, cin)[:, 1:].permute(0, 2, 1).reshape(B, cin, h, w)
    ref = F.conv_transpose2d(x, ct.weight.bfloat16().float(), ct.bias, stride=s)
    torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=2e-2)
    # the halo stays zero
    r = img.rows.view(B, s * h + 2, s * w + 2, c)
    assert not r[:, 0].any() and not r[:, -1].any() and not r[:, :, 0].any() \
        and not r[:, :, -1].any()


@pytest.mark.parametrize('Y,X,step', [(18, 50, 2), (7, 5, 2), (9, 9, 3)])
def test_image_subsample(Y, X, step):
    torch.manual_seed(2)
    x = torch.randn(2, 16, Y, X, device=DEV).bfloat16()
    out = conv3d_ops.image_subsample(conv3d_ops.pack_image(x), step)
    got = conv3d_ops.unpack_image(out, torch.bfloat16)
    assert torch.equal(got, x[:, :, ::step, ::step])


def test_volume_maxpool2_equals_torch_amax():
    """veon_volume_maxpool2_f32 (the camera-sharded path's ds_feat step after the
    cross-rank sum) against view + amax (view_transformer_raw.py:549-553), bit for bit,
    NaN included; through VeonOccupancyPath._max_pool's dispatch."""
    import ctypes
    from veon_amd import _lib
    g = torch.Generator().manual_seed(5)
    for B, C, Z, Y, X in ((1, 7, 4, 6, 10), (2, 16, 16, 20, 200), (1, 64, 16, 200, 200)):
        vol = torch.randn(B, C, Z, Y, X, generator=g).to(DEV)
        vol[0, 0, 0, 0, 1] = float('nan')
        vol[0, -1, -1, -1, -1] = float('-inf')
        out = torch.empty(B, C, Z // 2, Y // 2, X // 2, device=DEV)
        st = _lib.lib().veon_volume_maxpool2_f32(_lib.ptr(vol), _lib.ptr(out), B * C, Z, Y, X,
                                                 _lib.stream_ptr(vol.device))
        _lib.check(st, 'veon_volume_maxpool2_f32')
        want = vol.view(B, C, Z // 2, 2, Y // 2, 2, X // 2, 2).amax(dim=(3, 5, 7))
        assert torch.equal(torch.nan_to_num(out, nan=12345.0), torch.nan_to_num(want, nan=12345.0))
        assert torch.isnan(out[0, 0, 0, 0, 0])
    # odd extents are refused
    assert _lib.lib().veon_volume_maxpool2_f32(_lib.ptr(vol), _lib.ptr(out), 1, 3, 4, 4,
                                               ctypes.c_void_p(0)) != 0


def test_sensor2keyego_kernel_equals_the_reference_algebra():
    """AlignNetOcc3D.prepare_meta's 4x4 algebra (align_net_occ3d.py:328-352) in one
    launch: inverse(ego2global[:, 0]) @ ego2global @ sensor2ego in double precision,
    against torch's own double-precision inverse / matmul of the same inputs."""
    import torch
    from veon_amd import lss_prepare_hip, synthetic
    rig = synthetic.make_rig(2, 6, (256, 704))
    s2e = rig['sensor2ego'].to('cuda:0')
    g = torch.Generator().manual_seed(3)
    # general rigid ego poses (yaw + translation per camera time stamp), not identity
    e2g = torch.eye(4).repeat(2, 6, 1, 1)
    ang = torch.rand(2, 6, generator=g) * 6.28
    e2g[..., 0, 0], e2g[..., 0, 1] = ang.cos(), -ang.sin()
    e2g[..., 1, 0], e2g[..., 1, 1] = ang.sin(), ang.cos()
    e2g[..., :3, 3] = torch.randn(2, 6, 3, generator=g) * 50
    e2g = e2g.to('cuda:0')
    got = lss_prepare_hip.sensor2keyego(s2e, e2g)
    key = e2g[:, :1].double()
    want = (torch.linalg.inv(key) @ e2g.double() @ s2e.double()).float()
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-5)
    # and through the decoder's prepare_meta (one frame, inference)
    from veon_amd.models.semantic_net import AlignNetOcc3D
    dec = AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=64, clip_outdim=24,
                        layer_lifting_map=['2->0->0'], fusion_type='cat_fusion', layer_depth=2)
    dec.num_frame, dec.num_camera = 1, 6
    metas = [s2e, e2g, rig['intrins'].to('cuda:0'), rig['post_rots'].to('cuda:0'),
             rig['post_trans'].to('cuda:0'), rig['bda'].to('cuda:0')[None]]
    with torch.no_grad():
        fast = dec.prepare_meta(metas)
    with torch.enable_grad():          # the torch definition
        slow = dec.prepare_meta(metas)
    for a, b in zip(fast, slow):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-5)

"""Torch-level wrappers of the ViT block kernels (csrc/vit_block.hip).

bf16 tensors are ordinary ``torch.bfloat16`` tensors; the residual stream and
all vectors (bias, LayerNorm / LayerScale parameters) are fp32.  Everything
launches on the current stream; there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from . import half as _half

EPI_BF16, EPI_GELU, EPI_QUICKGELU, EPI_RESID = 0, 1, 2, 3
EPI_AFFINE, EPI_AFFINE_RELU = 4, 5
EPI_AFFINE_SIGM = 6     # sigmoid(gamma * x + bias) - 0.5


def _dev(*ts):
    return _lib.require_device(*ts)


def to_bf16(x):
    """fp32 -> bf16 (round to nearest even) on the device."""
    dev = _dev(x)
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=_half.dtype(), device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_cast_bf16(_lib.ptr(x), _lib.ptr(out), x.numel(),
                                           _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_cast_bf16')
    return out


def patchify(img, patch, skip=0, kpad=None):
    """img fp32 (B,C,H,W) -> bf16 [B*(skip + h*w), kpad]: the conv-weight-ordered
    p x p patches as GEMM rows, ``skip`` zero rows in front of every image."""
    dev = _dev(img)
    img = img.contiguous().float()
    B, C, H, W = img.shape
    k = C * patch * patch
    kpad = kpad or (k + 63) // 64 * 64
    out = torch.empty((B * (skip + (H // patch) * (W // patch)), kpad), dtype=_half.dtype(),
                      device=dev)
    with _lib.on_device(dev):
        st = _lib.lib().veon_vit_patchify(_lib.ptr(img), _lib.ptr(out), B, C, H, W, patch,
                                          skip, kpad, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_patchify')
    return out


def layernorm(x, weight, bias, eps=1e-6, out=None):
    """x fp32 [..., d] -> bf16 [..., d] (nn.LayerNorm over the last dim)."""
    dev = _dev(x, weight, bias)
    d = x.shape[-1]
    T = x.numel() // d
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=_half.dtype(), device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_layernorm(
            _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), T, d,
            float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_layernorm')
    return out


def layernorm_padded(x, weight, bias, d, eps=1e-6):
    """x fp32 [T, ld] whose first ``d`` columns are the token -> bf16 [T, ld]:
    nn.LayerNorm over those d columns, zeros in the padding."""
    dev = _dev(x, weight, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    T, ld = x.shape
    assert weight.numel() >= d and bias.numel() >= d and d <= ld
    out = torch.empty((T, ld), dtype=_half.dtype(), device=dev)
    with _lib.on_device(dev):
        st = _lib.lib().veon_vit_layernorm_padded(
            _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), T, int(d), ld,
            float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_layernorm_padded')
    return out


def layernorm_f32(x, weight, bias, eps=1e-5):
    """x fp32 [..., d] -> fp32 [..., d] (nn.LayerNorm over the last dim);
    d % 128 == 0, d <= 1024."""
    dev = _dev(x, weight, bias)
    d = x.shape[-1]
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    with _lib.on_device(dev):
        st = _lib.lib().veon_layernorm_f32(
            _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), x.numel() // d, d,
            float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_layernorm_f32')
    return out


def layernorm_f32_add_nearest(x, add, map_shape, add_shape, weight, bias, eps=1e-5):
    """LayerNorm(x + offset): x fp32 [B, L, d]; ``add`` fp32 [B, h*w, d] is resized
    (nearest, F.interpolate's default) from ``add_shape`` = (h, w) to ``map_shape`` =
    (Y, X) and added to the last Y*X tokens of every sample."""
    dev = _dev(x, add, weight, bias)
    B, L, d = x.shape
    (Y, X), (h, w) = map_shape, add_shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    assert add.dtype == torch.float32 and add.is_contiguous() and add.shape == (B, h * w, d)
    out = torch.empty_like(x)
    with _lib.on_device(dev):
        st = _lib.lib().veon_layernorm_f32_add_nearest(
            _lib.ptr(x), _lib.ptr(add), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out),
            B, L, d, Y, X, h, w, float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_layernorm_f32_add_nearest')
    return out


def linear(a, w, bias=None, epilogue=EPI_BF16, out=None, gamma=None):
    """a bf16 [M,K], w bf16 [N,K] (nn.Linear layout) -> bf16 [M,N].  ``gamma``
    (fp32 [N]) is the per-feature scale of the EPI_AFFINE* epilogues."""
    dev = _dev(a, w)
    _lib.require_half(a, w)
    assert a.is_contiguous() and w.is_contiguous()
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=_half.dtype(), device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_gemm(
            _lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(gamma),
            _lib.ptr(None), _lib.ptr(out), M, N, K, epilogue,
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_gemm')
    return out


def linear_residual_(resid, a, w, bias=None, gamma=None):
    """resid fp32 [M,N] += gamma * (a @ w^T + bias), in place."""
    dev = _dev(resid, a, w)
    assert resid.dtype == torch.float32 and resid.is_contiguous()
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    assert resid.numel() == M * N
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_gemm(
            _lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(gamma),
            _lib.ptr(resid), _lib.ptr(None), M, N, K, EPI_RESID,
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_gemm')
    return resid


LOG2E = 1.4426950408889634   # folded into q by the packers that set ``q_log2``


def attention(qkv, num_heads, bias=None, out=None, q_log2=False):
    """qkv bf16 [B,T,3*H*64] (q pre-scaled by head_dim^-0.5; ``q_log2``: also by
    ``LOG2E``, the exp2-domain form of the kernel) -> bf16 [B,T,H*64].
    bias: optional fp32 additive logits, broadcastable [B|1, H|1, T, T]."""
    dev = _dev(qkv)
    B, T, three_d = qkv.shape
    H = num_heads
    hd = three_d // (3 * H)
    _lib.require_half(qkv)
    assert qkv.is_contiguous()
    if out is None:
        out = torch.empty((B, T, H * hd), dtype=_half.dtype(), device=dev)
    sb = sh = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.dim() == 4
        assert bias.shape[-2:] == (T, T) and bias.stride(-1) == 1 \
            and bias.stride(-2) == T
        sb = bias.stride(0) if bias.shape[0] > 1 else 0
        sh = bias.stride(1) if bias.shape[1] > 1 else 0
    with torch.cuda.device(dev):
        L = _lib.lib()
        fn = L.veon_vit_attention_log2 if q_log2 else L.veon_vit_attention
        st = fn(_lib.ptr(qkv), _lib.ptr(bias), sb, sh, _lib.ptr(out), B, T, H, hd,
                _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_attention')
    return out


class BlockWeights:
    """Device weights of one transformer block packed for ``veon_vit_block``
    (include/veon_hip.h): keeps the tensors alive and the C struct ready."""

    def __init__(self, heads, n1, w_qkv, b_qkv, w_proj, b_proj, g1, n2, w_fc1,
                 b_fc1, w_fc2, b_fc2, g2, act, q_log2=False):
        self.heads = heads
        self.q_log2 = bool(q_log2)
        self.keep = (n1[0], n1[1], w_qkv, b_qkv, w_proj, b_proj, g1, n2[0], n2[1],
                     w_fc1, b_fc1, w_fc2, b_fc2, g2)
        for t in self.keep:
            assert t is None or (t.is_cuda and t.is_contiguous())
        self.d = w_qkv.shape[1]
        self.mlp_dim = w_fc1.shape[0]
        p = _lib.ptr
        self.c = _lib.VitBlockWeights(
            p(n1[0]).value, p(n1[1]).value, p(w_qkv).value, p(b_qkv).value,
            p(w_proj).value, p(b_proj).value, p(g1).value, p(n2[0]).value,
            p(n2[1]).value, p(w_fc1).value, p(b_fc1).value, p(w_fc2).value,
            p(b_fc2).value, p(g2).value, float(n1[2]), float(n2[2]),
            int(self.mlp_dim), int(act), int(self.q_log2))


def block_workspace(B, T, d, mlp_dim, device):
    """Workspace of ``veon_vit_block``.  It ends in the sync words of the split-K fc2
    (4 KiB), which every call expects zero and leaves zero: only those are cleared."""
    n = _lib.lib().veon_vit_block_workspace_bytes(B, T, d, mlp_dim)
    ws = torch.empty(n, dtype=torch.uint8, device=device)
    ws[n - 4096:].zero_()
    return ws


def linear_residual_splitk_(resid, a, w, bias=None, gamma=None, workspace=None):
    """``linear_residual_`` by the split-K kernel (fc2 shapes); raises when the shape is
    not one ``veon_vit_gemm_splitk_plan`` splits.  ``workspace``: (slab uint8 tensor,
    zeroed int32 sync tensor) to reuse; allocated when None."""
    dev = _dev(resid, a, w)
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    L = _lib.lib()
    need = L.veon_vit_gemm_splitk_plan(M, N, K, None)
    if need == 0:
        raise _lib.VeonHipError('no split-K plan for %d x %d x %d' % (M, N, K))
    if workspace is None:
        workspace = (torch.empty(need, dtype=torch.uint8, device=dev),
                     torch.zeros(1024, dtype=torch.int32, device=dev))
    slab, sync = workspace
    with _lib.on_device(dev):
        st = L.veon_vit_gemm_splitk(_lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(gamma),
                                    _lib.ptr(resid), M, N, K, _lib.ptr(slab), slab.numel(),
                                    _lib.ptr(sync), sync.numel(), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_gemm_splitk')
    return resid


def block_forward_(x, w, B, T, ws, bias=None):
    """One pre-norm block on the fp32 residual stream x [B*T, d], in place, as a
    single native call (seven launches)."""
    dev = _dev(x, ws)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape == (B * T, w.d)
    sb = sh = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.dim() == 4
        assert bias.shape[-2:] == (T, T) and bias.stride(-1) == 1 \
            and bias.stride(-2) == T
        sb = bias.stride(0) if bias.shape[0] > 1 else 0
        sh = bias.stride(1) if bias.shape[1] > 1 else 0
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_block(
            _lib.ptr(x), ctypes.byref(w.c), _lib.ptr(bias), sb, sh, _lib.ptr(ws),
            ws.numel(), B, T, w.d, w.heads, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_block')
    return x

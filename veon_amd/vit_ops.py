"""Torch-level wrappers of the ViT block kernels (csrc/vit_block.hip).

bf16 tensors are ordinary ``torch.bfloat16`` tensors; the residual stream and
all vectors (bias, LayerNorm / LayerScale parameters) are fp32.  Everything
launches on the current stream; there is no CPU path.
"""
import torch

from . import _lib

EPI_BF16, EPI_GELU, EPI_QUICKGELU, EPI_RESID = 0, 1, 2, 3


def _dev(*ts):
    return _lib.require_device(*ts)


def to_bf16(x):
    """fp32 -> bf16 (round to nearest even) on the device."""
    dev = _dev(x)
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_cast_bf16(_lib.ptr(x), _lib.ptr(out), x.numel(),
                                           _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_cast_bf16')
    return out


def layernorm(x, weight, bias, eps=1e-6, out=None):
    """x fp32 [..., d] -> bf16 [..., d] (nn.LayerNorm over the last dim)."""
    dev = _dev(x, weight, bias)
    d = x.shape[-1]
    T = x.numel() // d
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_layernorm(
            _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), T, d,
            float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_layernorm')
    return out


def linear(a, w, bias=None, epilogue=EPI_BF16, out=None):
    """a bf16 [M,K], w bf16 [N,K] (nn.Linear layout) -> bf16 [M,N]."""
    dev = _dev(a, w)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert a.is_contiguous() and w.is_contiguous()
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_gemm(
            _lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(None),
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


def attention(qkv, num_heads, bias=None, out=None):
    """qkv bf16 [B,T,3*H*64] (q pre-scaled) -> bf16 [B,T,H*64].
    bias: optional fp32 additive logits, broadcastable [B|1, H|1, T, T]."""
    dev = _dev(qkv)
    B, T, three_d = qkv.shape
    H = num_heads
    hd = three_d // (3 * H)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    if out is None:
        out = torch.empty((B, T, H * hd), dtype=torch.bfloat16, device=dev)
    sb = sh = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.dim() == 4
        assert bias.shape[-2:] == (T, T) and bias.stride(-1) == 1 \
            and bias.stride(-2) == T
        sb = bias.stride(0) if bias.shape[0] > 1 else 0
        sh = bias.stride(1) if bias.shape[1] > 1 else 0
    with torch.cuda.device(dev):
        st = _lib.lib().veon_vit_attention(
            _lib.ptr(qkv), _lib.ptr(bias), sb, sh, _lib.ptr(out), B, T, H, hd,
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_vit_attention')
    return out

"""Torch-level wrappers of the Conv3d body kernels (csrc/conv3d.hip).

``PaddedVolume`` owns the zero-padded channels-last bf16 grid
[B][Z+2][Y+2][X+2][C] (plus the guard rows the kernel may read) that the
implicit-GEMM conv consumes and produces.  Everything launches on the current
stream; there is no CPU path.
"""
import torch

from . import _lib
from . import half as _half


class PaddedVolume:
    """Channels-last bf16 volume with a zero halo.  ``rows`` is the
    [M, C] view of the padded grid (M = B*(Z+2)*(Y+2)*(X+2)); the storage has
    ``veon_conv3d_guard_rows`` zero rows before and after it."""

    def __init__(self, B, C, Z, Y, X, device):
        self.shape = (int(B), int(C), int(Z), int(Y), int(X))
        B, C, Z, Y, X = self.shape
        self.guard = int(_lib.lib().veon_conv3d_guard_rows(Y, X))
        self.M = B * (Z + 2) * (Y + 2) * (X + 2)
        self.storage = torch.zeros((self.M + 2 * self.guard, C),
                                   dtype=_half.dtype(), device=device)
        self.rows = self.storage[self.guard:self.guard + self.M]

    @property
    def device(self):
        return self.storage.device

    def like(self, C=None):
        B, C0, Z, Y, X = self.shape
        return PaddedVolume(B, C0 if C is None else C, Z, Y, X, self.device)

    def interior(self):
        """(B,Z,Y,X,C) bf16 view of the un-padded voxels."""
        B, C, Z, Y, X = self.shape
        return self.rows.view(B, Z + 2, Y + 2, X + 2, C)[:, 1:-1, 1:-1, 1:-1]


def pack(x, out=None):
    """(B,C,Z,Y,X) fp32 -> PaddedVolume (bf16, round to nearest even)."""
    dev = _lib.require_device(x)
    x = x.contiguous().float()
    B, C, Z, Y, X = x.shape
    if out is None:
        out = PaddedVolume(B, C, Z, Y, X, dev)
    assert out.shape == tuple(x.shape)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_volume_pack_bf16(_lib.ptr(x), _lib.ptr(out.rows), B, C,
                                              Z, Y, X, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_volume_pack_bf16')
    return out


def unpack(vol, out=None):
    """PaddedVolume -> (B,C,Z,Y,X) fp32."""
    dev = _lib.require_device(vol.storage)
    B, C, Z, Y, X = vol.shape
    if out is None:
        out = torch.empty(vol.shape, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_volume_unpack_f32(_lib.ptr(vol.rows), _lib.ptr(out), B,
                                               C, Z, Y, X, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_volume_unpack_f32')
    return out


def pack_weight(w):
    """nn.Conv3d weight (Cout,Cin,3,3,3) -> bf16 [Cout][3][3][3][Cin]."""
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
    return w.detach().permute(0, 2, 3, 4, 1).contiguous().to(_half.dtype())


def conv3d_k3(vol, w_packed, scale=None, shift=None, resid=None, relu=False,
              out=None, act=None):
    """3x3x3 stride-1 pad-1 convolution on a PaddedVolume with the fused
    epilogue ``act(conv*scale + shift + resid?)`` -> PaddedVolume; ``act`` in
    none / relu / gelu (``relu=True`` is shorthand for act='relu')."""
    dev = _lib.require_device(vol.storage, w_packed)
    B, Cin, Z, Y, X = vol.shape
    Cout = w_packed.shape[0]
    _lib.require_half(w_packed, vol.rows)
    assert w_packed.is_contiguous()
    assert w_packed.numel() == Cout * 27 * Cin
    if out is None:
        out = vol.like(Cout)
    assert out.shape == (B, Cout, Z, Y, X) and out is not vol
    if resid is not None:
        assert resid.shape == out.shape
    with torch.cuda.device(dev):
        st = _lib.lib().veon_conv3d_k3_bf16(
            _lib.ptr(vol.rows), _lib.ptr(w_packed), _lib.ptr(scale),
            _lib.ptr(shift), _lib.ptr(None if resid is None else resid.rows),
            _lib.ptr(out.rows), B, Z, Y, X, Cin, Cout, 1 if relu else _ACT[act],
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_conv3d_k3_bf16')
    return out


def deform_attention(kv, q, off, heads, samples=8, out=None):
    """Sampling + attention core of TemporalDeformable on PaddedVolumes: ``kv``
    (2C channels, per head [key | value]), ``q`` (C), ``off`` (>= heads*samples*3
    raw offsets; tanh is applied here) -> PaddedVolume (C), halo zero."""
    dev = _lib.require_device(kv.storage, q.storage, off.storage)
    B, C, Z, Y, X = q.shape
    assert kv.shape == (B, 2 * C, Z, Y, X) and off.shape[0] == B
    assert tuple(off.shape[2:]) == (Z, Y, X)
    if out is None:
        out = q.like()
    assert out.shape == q.shape and out is not q
    with torch.cuda.device(dev):
        st = _lib.lib().veon_deform_attention_bf16(
            _lib.ptr(kv.rows), _lib.ptr(q.rows), _lib.ptr(off.rows), _lib.ptr(out.rows),
            B, Z, Y, X, C, heads, samples, off.shape[1], _lib.stream_ptr(dev))
    _lib.check(st, 'veon_deform_attention_bf16')
    return out


def warp_volume(vol, affine, out=None):
    """Trilinear resampling of a PaddedVolume at ``affine`` (B,3,4 fp32, voxel
    index units) applied to each output voxel index; zeros outside."""
    dev = _lib.require_device(vol.storage)
    B, C, Z, Y, X = vol.shape
    affine = affine.to(device=dev, dtype=torch.float32).contiguous()
    assert tuple(affine.shape) == (B, 3, 4)
    if out is None:
        out = vol.like()
    assert out.shape == vol.shape and out is not vol
    with torch.cuda.device(dev):
        st = _lib.lib().veon_volume_warp_bf16(
            _lib.ptr(vol.rows), _lib.ptr(out.rows), _lib.ptr(affine), B, C, Z, Y, X,
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_volume_warp_bf16')
    return out


def warp_affine(cur2glob, prev2glob, first_xyz, step_xyz):
    """(B,1,4,4) or (B,4,4) ROCm fp32 ego->global transforms of the current and a
    past frame -> (B,3,4) voxel-index map for ``warp_volume`` (no host sync)."""
    import ctypes
    dev = _lib.require_device(cur2glob, prev2glob)
    mats = []
    for m in (cur2glob, prev2glob):
        m = m.float()
        if m.dim() == 4:
            m = m[:, 0]
        mats.append(m.contiguous())
    B = mats[0].shape[0]
    assert tuple(mats[0].shape) == tuple(mats[1].shape) == (B, 4, 4)
    out = torch.empty((B, 3, 4), dtype=torch.float32, device=dev)
    first = (ctypes.c_double * 3)(*[float(v) for v in first_xyz])
    step = (ctypes.c_double * 3)(*[float(v) for v in step_xyz])
    with torch.cuda.device(dev):
        st = _lib.lib().veon_warp_affine(_lib.ptr(mats[0]), _lib.ptr(mats[1]), 16,
                                         first, step, _lib.ptr(out), B,
                                         _lib.stream_ptr(dev))
    _lib.check(st, 'veon_warp_affine')
    return out


def zero_halo(vol):
    """Reset the halo rows of a PaddedVolume to zero, in place."""
    dev = _lib.require_device(vol.storage)
    B, C, Z, Y, X = vol.shape
    with torch.cuda.device(dev):
        st = _lib.lib().veon_volume_zero_halo_bf16(_lib.ptr(vol.rows), B, C, Z, Y, X,
                                                   _lib.stream_ptr(dev))
    _lib.check(st, 'veon_volume_zero_halo_bf16')
    return vol


# ----------------------------------------------------------------- 2-D images
class PaddedImage:
    """(B,C,Y,X) images as a zero-haloed channels-last bf16 grid
    [B][Y+2][X+2][C] (+ guard rows), the 2-D twin of ``PaddedVolume``."""

    def __init__(self, B, C, Y, X, device):
        self.shape = (int(B), int(C), int(Y), int(X))
        B, C, Y, X = self.shape
        self.guard = int(_lib.lib().veon_conv3d_guard_rows(Y, X))
        self.M = B * (Y + 2) * (X + 2)
        self.storage = torch.zeros((self.M + 2 * self.guard, C),
                                   dtype=_half.dtype(), device=device)
        self.rows = self.storage[self.guard:self.guard + self.M]

    @property
    def device(self):
        return self.storage.device


def pack_image(x, out=None):
    """(B,C,H,W) fp32 or bf16 -> PaddedImage."""
    dev = _lib.require_device(x)
    if x.dtype not in (torch.float32, _half.dtype()):
        x = x.float()
    B, C, Y, X = x.shape
    if out is None:
        out = PaddedImage(B, C, Y, X, dev)
    assert out.shape == tuple(x.shape)
    if not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
        # already channels-last (e.g. a MIOpen NHWC result): one strided copy
        # into the interior, no transpose
        out.rows.view(B, Y + 2, X + 2, C)[:, 1:-1, 1:-1].copy_(x.permute(0, 2, 3, 1))
        return out
    x = x.contiguous()
    with torch.cuda.device(dev):
        st = _lib.lib().veon_image_pack_bf16(
            _lib.ptr(x), 1 if x.dtype == _half.dtype() else 0, _lib.ptr(out.rows),
            B, C, Y, X, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_pack_bf16')
    return out


def unpack_image(img, dtype=torch.float32, channels=None):
    """PaddedImage -> (B,C,H,W) fp32 or bf16 (first ``channels`` channels)."""
    dev = _lib.require_device(img.storage)
    B, C, Y, X = img.shape
    out = torch.empty(img.shape, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_image_unpack(
            _lib.ptr(img.rows), _lib.ptr(out), 1 if dtype == _half.dtype() else 0,
            B, C, Y, X, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_unpack')
    return out if channels is None else out[:, :channels]


def pack_weight2d(w, pad_out_to=8):
    """nn.Conv2d weight (Cout,Cin,3,3) -> bf16 [Cout'][3][3][Cin], Cout padded
    with zero filters to a multiple of ``pad_out_to``."""
    assert w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
    cout = w.shape[0]
    npad = (cout + pad_out_to - 1) // pad_out_to * pad_out_to
    wp = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
    wp[:cout] = w.detach().float()
    return wp.permute(0, 2, 3, 1).contiguous().to(_half.dtype())


_ACT = {None: 0, 'none': 0, 'relu': 1, 'gelu': 2}


def conv2d_k3(img, w_packed, scale=None, shift=None, resid=None, relu=False,
              out=None, act=None, resid2=None, out_relu=None):
    """3x3 stride-1 pad-1 convolution on a PaddedImage with the fused epilogue
    ``act(conv*scale + shift + resid? + resid2?)`` -> PaddedImage; ``act`` in none /
    relu / gelu (``relu=True`` is shorthand for act='relu').  ``out_relu``: a second
    PaddedImage that receives relu(result)."""
    dev = _lib.require_device(img.storage, w_packed)
    B, Cin, Y, X = img.shape
    Cout = w_packed.shape[0]
    _lib.require_half(w_packed)
    assert w_packed.is_contiguous()
    assert w_packed.numel() == Cout * 9 * Cin
    if out is None:
        out = PaddedImage(B, Cout, Y, X, dev)
    assert out.shape == (B, Cout, Y, X) and out is not img
    for extra in (resid, resid2, out_relu):
        assert extra is None or extra.shape == out.shape
    assert out_relu is None or (out_relu is not out and out_relu is not img)

    def rows(p):
        return _lib.ptr(None if p is None else p.rows)
    with torch.cuda.device(dev):
        if resid2 is None and out_relu is None:
            st = _lib.lib().veon_conv2d_k3_bf16(
                _lib.ptr(img.rows), _lib.ptr(w_packed), _lib.ptr(scale), _lib.ptr(shift),
                rows(resid), _lib.ptr(out.rows),
                B, Y, X, Cin, Cout, 1 if relu else _ACT[act], _lib.stream_ptr(dev))
        else:
            st = _lib.lib().veon_conv2d_k3_bf16_ex(
                _lib.ptr(img.rows), _lib.ptr(w_packed), _lib.ptr(scale), _lib.ptr(shift),
                rows(resid), rows(resid2), _lib.ptr(out.rows), rows(out_relu),
                B, Y, X, Cin, Cout, 1 if relu else _ACT[act], _lib.stream_ptr(dev))
    _lib.check(st, 'veon_conv2d_k3_bf16')
    return out


def conv2d_k3s2(img, w_packed, scale=None, shift=None, out=None, act=None):
    """3x3 stride-2 pad-1 convolution on a PaddedImage -> PaddedImage
    (B, Cout, ceil(Y/2), ceil(X/2)), epilogue as ``conv2d_k3``."""
    dev = _lib.require_device(img.storage, w_packed)
    B, Cin, Y, X = img.shape
    Cout = w_packed.shape[0]
    _lib.require_half(w_packed)
    assert w_packed.is_contiguous()
    assert w_packed.numel() == Cout * 9 * Cin
    Yo, Xo = (Y + 1) // 2, (X + 1) // 2
    if out is None:
        out = PaddedImage(B, Cout, Yo, Xo, dev)
    assert out.shape == (B, Cout, Yo, Xo) and out is not img
    with _lib.on_device(dev):
        st = _lib.lib().veon_conv2d_k3s2_bf16(
            _lib.ptr(img.rows), _lib.ptr(w_packed), _lib.ptr(scale), _lib.ptr(shift),
            _lib.ptr(None), _lib.ptr(out.rows), B, Y, X, Cin, Cout, _ACT[act],
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_conv2d_k3s2_bf16')
    return out


def image_layernorm(img, gamma, beta, eps, out=None, tokens=False, residual=None):
    """LayerNorm over the channels of every pixel of a PaddedImage.  ``tokens``
    False: -> PaddedImage (zero halo); True: -> fp32 tokens (B, Y*X, C), plus
    ``residual`` (fp32 tokens of that shape) when given."""
    dev = _lib.require_device(img.storage, gamma, beta)
    B, C, Y, X = img.shape
    assert gamma.dtype == beta.dtype == torch.float32 and gamma.numel() == beta.numel() == C
    if tokens:
        if out is None:
            out = torch.empty((B, Y * X, C), dtype=torch.float32, device=dev)
        assert out.is_contiguous() and tuple(out.shape) == (B, Y * X, C)
        if residual is not None:
            assert (residual.dtype == torch.float32 and residual.is_contiguous()
                    and tuple(residual.shape) == (B, Y * X, C))
        dst = out
    else:
        if out is None:
            out = PaddedImage(B, C, Y, X, dev)
        assert out.shape == img.shape and out is not img and residual is None
        dst = out.rows
    with torch.cuda.device(dev):
        st = _lib.lib().veon_image_layernorm_bf16(
            _lib.ptr(img.rows), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(dst),
            1 if tokens else 0, B, C, Y, X, float(eps), _lib.ptr(residual),
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_layernorm_bf16')
    return out


def layernorm_tokens_to_image(x, gamma, beta, eps, out):
    """nn.LayerNorm of fp32 tokens (B, Y*X, C) written as bf16 into the interior of
    the PaddedImage ``out`` (its halo is left as it is: zero)."""
    dev = _lib.require_device(x, gamma, beta, out.storage)
    B, C, Y, X = out.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == B * Y * X * C
    with _lib.on_device(dev):
        st = _lib.lib().veon_layernorm_f32_to_padded(
            _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out.rows), B, Y, X, C,
            float(eps), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_layernorm_f32_to_padded')
    return out


def resize_bilinear(img, size, out=None):
    """F.interpolate(mode='bilinear', align_corners=True) on a PaddedImage."""
    dev = _lib.require_device(img.storage)
    B, C, Yi, Xi = img.shape
    Yo, Xo = int(size[0]), int(size[1])
    if out is None:
        out = PaddedImage(B, C, Yo, Xo, dev)
    assert out.shape == (B, C, Yo, Xo)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_image_resize_bilinear(
            _lib.ptr(img.rows), _lib.ptr(out.rows), B, C, Yi, Xi, Yo, Xo,
            _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_resize_bilinear')
    return out


def tokens_to_image(rows, tokens_per_image, skip, h, w, s, C, out):
    """bf16 token rows [B*tokens_per_image, >= s*s*C] (the GEMM output of a
    ConvTranspose2d(k = s, stride = s) applied to the tokens; s = 1: plain tokens)
    -> interior of the PaddedImage ``out`` (B, C, s*h, s*w); the first ``skip`` rows
    of every image (class token) are passed over."""
    dev = _lib.require_device(rows, out.storage)
    B = out.shape[0]
    _lib.require_half(rows)
    assert rows.is_contiguous() and rows.dim() == 2
    assert out.shape == (B, C, s * h, s * w) and rows.shape[0] == B * tokens_per_image
    with _lib.on_device(dev):
        st = _lib.lib().veon_tokens_to_image(
            _lib.ptr(rows), rows.shape[1], tokens_per_image, skip, h, w, s, C,
            _lib.ptr(out.rows), B, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_tokens_to_image')
    return out


def image_subsample(img, step, out=None):
    """out(y,x) = img(step*y, step*x) -> PaddedImage (B, C, ceil(Y/step), ceil(X/step))."""
    dev = _lib.require_device(img.storage)
    B, C, Y, X = img.shape
    Yo, Xo = (Y + step - 1) // step, (X + step - 1) // step
    if out is None:
        out = PaddedImage(B, C, Yo, Xo, dev)
    assert out.shape == (B, C, Yo, Xo)
    with _lib.on_device(dev):
        st = _lib.lib().veon_image_subsample(_lib.ptr(img.rows), _lib.ptr(out.rows), B, C,
                                             Y, X, step, _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_subsample')
    return out


def image_dot(img, w, bias, act='none'):
    """1x1 conv C -> 1 (+activation) of a PaddedImage -> (B,1,H,W) fp32.
    w fp32 [C], act in none / relu / sigmoid."""
    dev = _lib.require_device(img.storage, w)
    B, C, Y, X = img.shape
    assert w.dtype == torch.float32 and w.numel() >= C and w.is_contiguous()
    out = torch.empty((B, 1, Y, X), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_image_dot(
            _lib.ptr(img.rows), _lib.ptr(w), float(bias), _lib.ptr(out), B, C, Y, X,
            {'none': 0, 'relu': 1, 'sigmoid': 2}[act], _lib.stream_ptr(dev))
    _lib.check(st, 'veon_image_dot')
    return out


def occ_classify(sem_low, bin_low, occ_size):
    """Trilinear upsampling (align_corners=False) of the class logits ``sem_low``
    (B,Q,z,y,x) and occupancy logits ``bin_low`` (B,2,z,y,x) -- fp32, ANY strides --
    to ``occ_size``, plus the label volume of VEONTemporal.simple_test, in one kernel.
    -> (sem_occ (B,Q,Z,Y,X), bin_occ (B,2,Z,Y,X), occ_pred_cls (B,X,Y,Z) int64)."""
    import ctypes
    dev = _lib.require_device(sem_low, bin_low)
    assert sem_low.dtype == bin_low.dtype == torch.float32
    B, Q, zi, yi, xi = sem_low.shape
    assert tuple(bin_low.shape) == (B, 2, zi, yi, xi)
    Zo, Yo, Xo = (int(v) for v in occ_size)
    sem = torch.empty((B, Q, Zo, Yo, Xo), dtype=torch.float32, device=dev)
    binv = torch.empty((B, 2, Zo, Yo, Xo), dtype=torch.float32, device=dev)
    cls = torch.empty((B, Xo, Yo, Zo), dtype=torch.int64, device=dev)
    s5 = ctypes.c_int64 * 5
    ss, bs = s5(*sem_low.stride()), s5(*bin_low.stride())
    with _lib.on_device(dev):
        st = _lib.lib().veon_occ_classify(
            _lib.ptr(sem_low), ctypes.cast(ss, ctypes.c_void_p), Q, _lib.ptr(bin_low),
            ctypes.cast(bs, ctypes.c_void_p), B, zi, yi, xi, Zo, Yo, Xo, _lib.ptr(sem),
            _lib.ptr(binv), _lib.ptr(cls), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_occ_classify')
    return sem, binv, cls

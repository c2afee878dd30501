"""Conv3d body of VEON's 3D alignment network -- mirror of ``ResBlock3D`` and the
``layers_3d_body`` stack of ``AlignNetOcc3D``
(mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:224-228, 363-399).

The reference builds each conv as an mmcv ``ConvModule`` (conv -> norm -> act,
bias=False, BN3d); parameter names follow that layout (``conv1.conv.weight``,
``conv1.bn.weight`` ...) so a VEON checkpoint's ``layers_3d_body.*`` keys load
unchanged.  mmcv is absent here, so this is a restatement of ConvModule's
documented order; numerics are pinned against torch's own Conv3d /
BatchNorm3d (tests/test_conv3d_gpu.py), not against reference-generated vectors.

In eval mode on a ROCm device the whole stack runs on the implicit-GEMM MFMA
kernel (csrc/conv3d.hip): the lifted (B,C,Z,Y,X) fp32 volume is packed once into
the zero-padded channels-last bf16 grid, every conv writes the next conv's
padded input, BatchNorm (eval) + ReLU + the identity add are the conv's
epilogue, and the result is unpacked once at the end.  Training and CPU tensors
take the plain PyTorch formulation (the definition of the module).
"""
import torch
import torch.nn as nn

from ... import conv3d_ops, vit_ops
from .._native_cache import NativeCacheMixin
from ... import half as _half


class ConvModule3d(NativeCacheMixin, nn.Module):
    """Conv3d -> BN3d -> ReLU with mmcv ConvModule's attribute names."""

    _native_cache = ('_hip', '_hip3', '_hip_out')

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1, bias=False,
                 norm=True, act=True):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride, padding, bias=bias)
        self.bn = nn.BatchNorm3d(cout) if norm else None
        self.activate = nn.ReLU(inplace=True) if act else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.activate is not None:
            x = self.activate(x)
        return x

    def folded(self, pack=True):
        """(packed bf16 weight, fp32 scale, fp32 shift) of conv + eval-mode BN."""
        w = conv3d_ops.pack_weight(self.conv.weight) if pack else None
        cout = self.conv.out_channels
        dev = self.conv.weight.device
        scale = torch.ones(cout, device=dev)
        shift = torch.zeros(cout, device=dev)
        if self.bn is not None:
            scale = (self.bn.weight.detach().float() /
                     torch.sqrt(self.bn.running_var.float() + self.bn.eps))
            shift = self.bn.bias.detach().float() - \
                self.bn.running_mean.float() * scale
        if self.conv.bias is not None:
            shift = shift + self.conv.bias.detach().float() * scale
        return w, scale.contiguous(), shift.contiguous()


def _pointwise(vol, cm, out_channels=None, epilogue=None):
    """1x1x1 ConvModule3d on a PaddedVolume as one GEMM over its rows (the halo
    rows get the bias/shift -- they are never unpacked)."""
    if '_hip' not in cm.__dict__ or cm.__dict__['_hip'] is None:
        conv = cm.conv
        cout = conv.out_channels
        npad = (cout + 7) // 8 * 8       # the GEMM wants N % 4 == 0, rows 16-B
        w = torch.zeros(npad, conv.in_channels, device=conv.weight.device)
        w[:cout] = conv.weight.detach().float().view(cout, -1)
        _, scale, shift = cm.folded(pack=False)
        sc = torch.ones(npad, device=w.device)
        sh = torch.zeros(npad, device=w.device)
        sc[:cout], sh[:cout] = scale, shift
        cm.__dict__['_hip'] = (w.to(_half.dtype()).contiguous(), sc, sh, npad)
    w, sc, sh, npad = cm.__dict__['_hip']
    B, C, Z, Y, X = vol.shape
    key = (B, npad, Z, Y, X, str(vol.device))
    bufs = cm.__dict__.setdefault('_hip_out', {})
    if key not in bufs:  # reused across calls: the result is consumed at once
        bufs[key] = conv3d_ops.PaddedVolume(B, npad, Z, Y, X, vol.device)
    out = bufs[key]
    epi = vit_ops.EPI_AFFINE_RELU if cm.activate is not None else vit_ops.EPI_AFFINE
    if epilogue is not None:
        assert cm.activate is None
        epi = epilogue
    vit_ops.linear(vol.rows, w, sh, epi, out=out.rows, gamma=sc)
    return out


class _PredHead3D(nn.Module):
    """Shared machinery of the two prediction heads: a chain of 1x1x1
    ConvModules.  On a PaddedVolume (or a ROCm fp32 volume at inference) the
    chain runs as GEMMs on the channels-last rows."""

    _names = ()

    def _chain(self):
        return [getattr(self, n) for n in self._names]

    def train(self, mode=True):
        for cm in self._chain():
            cm.__dict__['_hip'] = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        for cm in self._chain():
            cm.__dict__['_hip'] = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _hip_ok(self, x):
        if not all(cm.conv.in_channels % 64 == 0 for cm in self._chain()):
            return False
        if isinstance(x, conv3d_ops.PaddedVolume):
            return True
        return x.is_cuda and not self.training and not torch.is_grad_enabled()

    def _run(self, x, return_volume=False, last_epilogue=None):
        """``last_epilogue``: GEMM epilogue of the LAST conv on the native path (the
        caller's output activation fused in); ignored by the PyTorch path."""
        if not self._hip_ok(x):
            if isinstance(x, conv3d_ops.PaddedVolume):
                x = conv3d_ops.unpack(x)
            for cm in self._chain():
                x = cm(x)
            return x
        vol = x if isinstance(x, conv3d_ops.PaddedVolume) else conv3d_ops.pack(x)
        chain = self._chain()
        for cm in chain:
            vol = _pointwise(vol, cm, epilogue=last_epilogue if cm is chain[-1] else None)
        if return_volume:
            return vol
        out = conv3d_ops.unpack(vol)
        return out[:, :self._chain()[-1].conv.out_channels]


class PredHead3DOcc(_PredHead3D):
    """Binary occupancy head (align_net_occ3d.py:431-470): 1x1x1 conv C -> C/4,
    BN, ReLU; 1x1x1 conv C/4 -> channels_out."""

    _names = ('occ_conv1', 'occ_conv2')

    def __init__(self, channels_in, channels_out, stride=1, use_checkpoint=False):
        super().__init__()
        mid = channels_in // 4
        self.occ_conv1 = ConvModule3d(channels_in, mid, 1, stride, 0, bias=False,
                                      norm=True, act=True)
        self.occ_conv2 = ConvModule3d(mid, channels_out, 1, stride, 0, bias=False,
                                      norm=False, act=False)
        self.use_checkpoint = use_checkpoint

    def forward(self, x):
        return self._run(x)


class PredHead3DSem(_PredHead3D):
    """Feature head (align_net_occ3d.py:473-534): three 1x1x1 convs (BN+ReLU
    after the first two, the first with a conv bias), then sigmoid - 0.5."""

    _names = ('occ_conv1', 'occ_conv2', 'occ_conv3')

    def __init__(self, channels_in, channels_out, stride=1, use_checkpoint=False):
        super().__init__()
        self.occ_conv1 = ConvModule3d(channels_in, channels_in, 1, stride, 0,
                                      bias=True, norm=True, act=True)
        self.occ_conv2 = ConvModule3d(channels_in, channels_in, 1, stride, 0,
                                      bias=False, norm=True, act=True)
        self.occ_conv3 = ConvModule3d(channels_in, channels_out, 1, stride, 0,
                                      bias=False, norm=False, act=False)
        self.use_checkpoint = use_checkpoint

    def forward(self, x, return_volume=False):
        """``return_volume`` (MFMA path only): keep the result -- sigmoid - 0.5
        applied in place on the bf16 rows -- as the PaddedVolume that
        ``semantic_inference_3d_fused`` consumes."""
        if return_volume and self._hip_ok(x):
            # sigmoid(x) - 0.5 = tanh(x/2)/2 in the last GEMM's epilogue (fp32, before
            # the bf16 rounding; halo rows are never read)
            return self._run(x, True, last_epilogue=vit_ops.EPI_AFFINE_SIGM)
        out = self._run(x, return_volume)
        if isinstance(out, conv3d_ops.PaddedVolume):
            out.rows.mul_(0.5).tanh_().mul_(0.5)
            return out
        return out.sigmoid() - 0.5


def semantic_inference_3d(ov_classifier_weight, feat_occ, occ_size):
    """The reference's order (san_in_veon_temporal.py:196-201, 257-259):
    trilinear-upsample the C-channel feature volume to ``occ_size``, then
    ``einsum('qc,bczhw->bqzhw')`` with the open-vocabulary classifier."""
    feat = nn.functional.interpolate(feat_occ, size=tuple(occ_size), mode='trilinear',
                                     align_corners=False)
    return torch.einsum('qc,bczhw->bqzhw', ov_classifier_weight, feat)


def classifier_logits_low(ov_classifier_weight, feat_occ):
    """Class logits at the head's resolution: (B,Q,z,y,x) fp32 -- for a PaddedVolume
    (``PredHead3DSem(..., return_volume=True)``) a strided view of the GEMM's
    channels-last rows (MFMA, fp32 logits), otherwise the einsum."""
    W = ov_classifier_weight
    Q, C = W.shape
    if isinstance(feat_occ, conv3d_ops.PaddedVolume):
        vol = feat_occ
        B, Cv, Z, Y, X = vol.shape
        assert Cv >= C
        qp = (Q + 7) // 8 * 8
        wp = torch.zeros(qp, Cv, device=W.device)
        wp[:Q, :C] = W.detach().float()
        if Cv % 64 == 0:
            logits = torch.zeros(vol.M, qp, dtype=torch.float32, device=W.device)
            vit_ops.linear_residual_(logits, vol.rows, wp.to(_half.dtype()).contiguous())
        else:   # a K the MFMA tile does not divide (toy widths): rocBLAS, same operands
            logits = vol.rows.float() @ wp.to(_half.dtype()).float().t()
        return logits.view(B, Z + 2, Y + 2, X + 2, qp)[:, 1:-1, 1:-1, 1:-1, :Q] \
            .permute(0, 4, 1, 2, 3)
    return torch.einsum('qc,bczhw->bqzhw', W, feat_occ)


def semantic_inference_3d_fused(ov_classifier_weight, feat_occ, occ_size):
    """Same logits with the two linear maps swapped: classify at the head's
    resolution (one GEMM over the voxels: C -> Q classes), then upsample Q
    channels instead of C (768 -> ~20: the 2 GB upsampled feature volume of the
    reference is never formed).  Interpolation weights are per-channel and sum
    to one, the classifier is per-voxel linear, so the results agree up to
    rounding.  ``feat_occ``: (B,C,Z,Y,X) tensor or the PaddedVolume of
    ``PredHead3DSem(..., return_volume=True)`` (GEMM on MFMA, fp32 logits)."""
    low = classifier_logits_low(ov_classifier_weight, feat_occ)
    return nn.functional.interpolate(low, size=tuple(occ_size), mode='trilinear',
                                     align_corners=False)


class ResBlock3D(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x) (align_net_occ3d.py:363-399;
    ``stride`` / ``downsample`` as the reference, unused by VEON)."""

    def __init__(self, channels_in, channels_out, stride=1, downsample=None,
                 use_checkpoint=False):
        super().__init__()
        self.conv1 = ConvModule3d(channels_in, channels_out, 3, stride, 1,
                                  bias=False, norm=True, act=True)
        self.conv2 = ConvModule3d(channels_out, channels_out, 3, 1, 1,
                                  bias=False, norm=True, act=False)
        self.downsample = downsample
        self.relu = nn.ReLU(inplace=True)
        self.use_checkpoint = use_checkpoint

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        x = self.conv2(self.conv1(x))
        return self.relu(x + identity)

    def hip_supported(self):
        c1, c2 = self.conv1.conv, self.conv2.conv
        return (self.downsample is None and c1.stride == (1, 1, 1)
                and c1.in_channels == c1.out_channels
                and c1.in_channels % 64 == 0 and c2.out_channels % 8 == 0)


class AlignBody3D(nn.Module):
    """``layers_3d_body``: ``layer_depth`` ResBlock3D on the lifted volume
    (align_net_occ3d.py:224-228; applied one block per fusion step in
    ``forward`` :252-264 -- ``forward`` here runs blocks ``[start, stop)``)."""

    def __init__(self, embed_dim=256, layer_depth=4):
        super().__init__()
        self.layers_3d_body = nn.ModuleList(
            [ResBlock3D(embed_dim, embed_dim) for _ in range(layer_depth)])
        self.use_hip = True
        self._hip = None     # folded weights per block
        self._bufs = {}      # (shape, device) -> three PaddedVolumes

    def invalidate_hip_cache(self):
        self._hip = None

    def train(self, mode=True):
        self._hip = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .half()
        self._hip = None
        return super()._apply(fn, *args, **kwargs)

    def _use_hip(self, x):
        return (self.use_hip and x.is_cuda and not self.training
                and not torch.is_grad_enabled()
                and all(b.hip_supported() for b in self.layers_3d_body))

    def _volumes(self, shape, device):
        key = (tuple(shape), str(device))
        if key not in self._bufs:
            B, C, Z, Y, X = shape
            self._bufs[key] = [conv3d_ops.PaddedVolume(B, C, Z, Y, X, device)
                               for _ in range(3)]
        return self._bufs[key]

    def forward(self, x, start=0, stop=None, return_volume=False):
        """``return_volume``: hand the result over as the PaddedVolume the
        prediction heads consume directly (no unpack / re-pack)."""
        blocks = list(self.layers_3d_body)[start:stop]
        from_volume = isinstance(x, conv3d_ops.PaddedVolume)
        if from_volume and not (self.use_hip and not self.training
                                and all(b.hip_supported() for b in blocks)):
            x, from_volume = conv3d_ops.unpack(x), False
        if not from_volume and not self._use_hip(x):
            for blk in blocks:
                x = blk(x)
            return x
        if self._hip is None:
            self._hip = [(b.conv1.folded(), b.conv2.folded())
                         for b in self.layers_3d_body]
        folded = self._hip[start:stop]
        internal = list(self._volumes(x.shape, x.device))
        external = None
        if from_volume:   # e.g. written by the lift's fused max-pool kernel
            src = x
            if not any(x is b for b in internal):
                external = x               # read only, never overwritten
        else:
            src = internal[0]
            conv3d_ops.pack(x, out=src)
        pool = [b for b in internal if b is not src]
        for (w1, s1, b1), (w2, s2, b2) in folded:
            tmp, dst = pool[0], pool[1]
            conv3d_ops.conv3d_k3(src, w1, s1, b1, relu=True, out=tmp)
            conv3d_ops.conv3d_k3(tmp, w2, s2, b2, resid=src, relu=True, out=dst)
            pool = [tmp] + pool[2:] + ([src] if src is not external else [])
            src = dst
        a = src
        return a if return_volume else conv3d_ops.unpack(a)

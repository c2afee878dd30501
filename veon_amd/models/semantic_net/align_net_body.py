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

from ... import conv3d_ops


class ConvModule3d(nn.Module):
    """Conv3d -> BN3d -> ReLU with mmcv ConvModule's attribute names."""

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

    def folded(self):
        """(packed bf16 weight, fp32 scale, fp32 shift) of conv + eval-mode BN."""
        w = conv3d_ops.pack_weight(self.conv.weight)
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

    def forward(self, x, start=0, stop=None):
        blocks = list(self.layers_3d_body)[start:stop]
        if not self._use_hip(x):
            for blk in blocks:
                x = blk(x)
            return x
        if self._hip is None:
            self._hip = [(b.conv1.folded(), b.conv2.folded())
                         for b in self.layers_3d_body]
        folded = self._hip[start:stop]
        a, t, o = self._volumes(x.shape, x.device)
        conv3d_ops.pack(x, out=a)
        for (w1, s1, b1), (w2, s2, b2) in folded:
            conv3d_ops.conv3d_k3(a, w1, s1, b1, relu=True, out=t)
            conv3d_ops.conv3d_k3(t, w2, s2, b2, resid=a, relu=True, out=o)
            a, o = o, a
        return conv3d_ops.unpack(a)

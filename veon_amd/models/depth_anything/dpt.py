"""DepthAnythingV2 depth estimator -- mirror of
mmdet3d/models/depth_anything/dpt.py (DPTHead :39-150, DepthAnythingV2Adaptor
:226-263) and util/blocks.py.  The encoder's dense contractions run on MFMA
(``dinov2.py``).  The DPT convolution head is plain PyTorch / MIOpen in fp32
(the reference's numerics); when the adaptor runs the head in bf16
(``head_dtype = torch.bfloat16``) its heavy 3x3 convolutions -- the
ResidualConvUnits of the fusion blocks and the two output convolutions -- go to
the implicit-GEMM MFMA kernel (csrc/conv3d.hip, 2-D mode) with bias / ReLU /
identity fused, the rest (1x1 convs, transposed convs, bilinear resizes) stays
PyTorch.  Same module / parameter names as the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import conv3d_ops, vit_ops
from ..builder import register_neck
from .._native_cache import NativeCacheMixin
from .dinov2 import DINOv2Adaptor
from ... import half as _half


def _hip_convs_ok(x, *convs):
    """bf16 inference on a ROCm device with MFMA-shaped channel counts."""
    return (x.is_cuda and x.dtype == _half.dtype() and not torch.is_grad_enabled()
            and all(c.in_channels % 64 == 0 and c.kernel_size == (3, 3)
                    and c.stride == (1, 1) and c.padding == (1, 1) for c in convs))


class _HipConv:
    """Packed bf16 weight + fp32 bias of one nn.Conv2d, and per-shape buffers."""

    def __init__(self, conv):
        self.w = conv3d_ops.pack_weight2d(conv.weight)
        self.cout = conv.out_channels
        self.shift = torch.zeros(self.w.shape[0], device=conv.weight.device)
        if conv.bias is not None:
            self.shift[:self.cout] = conv.bias.detach().float()
        self.bufs = {}

    def out_buf(self, img, tag=0):
        key = (img.shape[0], img.shape[2], img.shape[3], tag)
        if key not in self.bufs:
            self.bufs[key] = conv3d_ops.PaddedImage(img.shape[0], self.w.shape[0],
                                                    img.shape[2], img.shape[3], img.device)
        return self.bufs[key]

    def __call__(self, img, relu=False, resid=None, tag=0, resid2=None, out_relu=None):
        return conv3d_ops.conv2d_k3(img, self.w, None, self.shift, resid=resid,
                                    relu=relu, out=self.out_buf(img, tag), resid2=resid2,
                                    out_relu=out_relu)


def _hip_cache(mod, name, conv):
    cache = mod.__dict__.setdefault('_hip_convs', {})
    if name not in cache:
        cache[name] = _HipConv(conv)
    return cache[name]


class ResidualConvUnit(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip_convs',)

    def __init__(self, features, bn=False):
        super().__init__()
        self.bn = bn
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        if bn:
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)

    def forward(self, x):
        out = self.conv1(F.relu(x))
        if self.bn:
            out = self.bn1(out)
        out = self.conv2(F.relu(out))
        if self.bn:
            out = self.bn2(out)
        return out + x

    def hip_forward(self, img, scratch, plus=None, out_relu=None):
        """conv2(relu(conv1(relu(x)))) + x on a PaddedImage (``plus``: + another image;
        ``out_relu``: relu(result) written too): bias + ReLU and bias + identity (+ the
        extras) are the convs' epilogues.  ReLU of the input: the producer left it in
        ``img.relu_copy`` (a conv with ``out_relu``), else one elementwise pass here
        (the zero halo stays zero)."""
        relu_in = getattr(img, 'relu_copy', None)
        if relu_in is None:
            torch.clamp_min(img.rows, 0, out=scratch.rows)
            relu_in = scratch
        u = _hip_cache(self, 'conv1', self.conv1)(relu_in, relu=True, tag=1)
        r = _hip_cache(self, 'conv2', self.conv2)(u, resid=img, tag=2, resid2=plus,
                                                  out_relu=out_relu)
        r.relu_copy = out_relu
        return r


class FeatureFusionBlock(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip_bufs', '_hip_1x1')

    """util/blocks.py:86-148 (expand=False, align_corners=True)."""

    def __init__(self, features, bn=False, size=None):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features, bn)
        self.resConfUnit2 = ResidualConvUnit(features, bn)
        self.size = size

    def _buf(self, tag, B, C, H, W, device):
        bufs = self.__dict__.setdefault('_hip_bufs', {})
        key = (tag, B, C, H, W)
        if key not in bufs:
            bufs[key] = conv3d_ops.PaddedImage(B, C, H, W, device)
        return bufs[key]

    def hip_ok(self, x):
        u1, u2 = self.resConfUnit1, self.resConfUnit2
        return (not u1.bn and _hip_convs_ok(x, u1.conv1, u1.conv2, u2.conv1, u2.conv2)
                and self.out_conv.in_channels % 64 == 0
                and self.out_conv.out_channels % 8 == 0)

    def hip_block(self, x0, x1, size):
        """The whole block on PaddedImages: [x0 += RCU1(x1)], RCU2, 1x1 out_conv
        as a GEMM on the rows, bilinear resize.  The 1x1 conv is applied BEFORE
        the resize: a per-pixel affine map and a per-channel interpolation with
        weights summing to one commute, and it is 4x less work there.  (x0 is
        updated in place; its producer does not need it afterwards.)"""
        B, C, H, W = x0.shape
        scratch = self._buf('relu', B, C, H, W, x0.device)
        if x1 is not None:
            # x0 + RCU1(x1) and its ReLU (RCU2's input) out of RCU1.conv2's epilogue
            x0 = self.resConfUnit1.hip_forward(x1, scratch, plus=x0,
                                               out_relu=self._buf('relu2', B, C, H, W,
                                                                  x0.device))
        y = self.resConfUnit2.hip_forward(x0, scratch)
        if '_hip_1x1' not in self.__dict__:
            oc = self.out_conv
            self.__dict__['_hip_1x1'] = (
                oc.weight.detach().float().view(oc.out_channels, -1)
                .to(_half.dtype()).contiguous(),
                oc.bias.detach().float().contiguous())
        w, bias = self.__dict__['_hip_1x1']
        z = self._buf('1x1', B, w.shape[0], H, W, x0.device)
        vit_ops.linear(y.rows, w, bias, vit_ops.EPI_BF16, out=z.rows)
        if size is None:
            size = self.size if self.size is not None else (2 * H, 2 * W)
        out = self._buf('up', B, w.shape[0], int(size[0]), int(size[1]), x0.device)
        return conv3d_ops.resize_bilinear(z, size, out=out)

    def forward(self, *xs, size=None):
        out = xs[0]
        if len(xs) == 2:
            out = out + self.resConfUnit1(xs[1])
        out = self.resConfUnit2(out)
        if size is None and self.size is None:
            kw = dict(scale_factor=2)
        else:
            kw = dict(size=self.size if size is None else size)
        out = F.interpolate(out, **kw, mode='bilinear', align_corners=True)
        return self.out_conv(out)


def _make_scratch(in_shape, out_shape):
    scratch = nn.Module()
    for i, c in enumerate(in_shape, 1):
        setattr(scratch, 'layer%d_rn' % i,
                nn.Conv2d(c, out_shape, 3, 1, 1, bias=False))
    return scratch


class DPTHead(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip_convs', '_hip_in', '_hip_tail', '_hip_front_w')

    def __init__(self, in_channels, features=256, use_bn=False,
                 out_channels=[256, 512, 1024, 1024], use_clstoken=False):
        super().__init__()
        self.use_clstoken = use_clstoken
        self.projects = nn.ModuleList(
            [nn.Conv2d(in_channels, oc, 1, 1, 0) for oc in out_channels])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(out_channels[0], out_channels[0], 4, 4, 0),
            nn.ConvTranspose2d(out_channels[1], out_channels[1], 2, 2, 0),
            nn.Identity(),
            nn.Conv2d(out_channels[3], out_channels[3], 3, 2, 1)])
        if use_clstoken:
            self.readout_projects = nn.ModuleList([
                nn.Sequential(nn.Linear(2 * in_channels, in_channels), nn.GELU())
                for _ in out_channels])
        self.scratch = _make_scratch(out_channels, features)
        self.scratch.stem_transpose = None
        for i in (1, 2, 3, 4):
            setattr(self.scratch, 'refinenet%d' % i,
                    FeatureFusionBlock(features, use_bn))
        self.scratch.output_conv1 = nn.Conv2d(features, features // 2, 3, 1, 1)
        self.scratch.output_conv2 = nn.Sequential(
            nn.Conv2d(features // 2, 32, 3, 1, 1), nn.ReLU(True),
            nn.Conv2d(32, 1, 1, 1, 0), nn.Sigmoid())

    def forward(self, out_features, patch_h, patch_w):
        out = []
        for i, x in enumerate(out_features):
            if self.use_clstoken:
                x, cls_token = x[0], x[1]
                readout = cls_token.unsqueeze(1).expand_as(x)
                x = self.readout_projects[i](torch.cat((x, readout), -1))
            else:
                x = x[0]
            x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], patch_h, patch_w)
            out.append(self.resize_layers[i](self.projects[i](x)))
        l1, l2, l3, l4 = out
        s = self.scratch
        l1, l2, l3, l4 = s.layer1_rn(l1), s.layer2_rn(l2), s.layer3_rn(l3), s.layer4_rn(l4)
        oc2 = s.output_conv2
        if (all(getattr(s, 'refinenet%d' % i).hip_ok(l1) for i in (1, 2, 3, 4))
                and _hip_convs_ok(l1, s.output_conv1, oc2[0])
                and all(t.dtype == _half.dtype() for t in (l2, l3, l4))):
            return self._hip_refine(l1, l2, l3, l4, patch_h, patch_w)
        p4 = s.refinenet4(l4, size=l3.shape[2:])
        p3 = s.refinenet3(p4, l3, size=l2.shape[2:])
        p2 = s.refinenet2(p3, l2, size=l1.shape[2:])
        p1 = s.refinenet1(p2, l1)
        out = s.output_conv1(p1)
        out = F.interpolate(out, (int(patch_h * 14), int(patch_w * 14)),
                            mode='bilinear', align_corners=True)
        return s.output_conv2(out)

    # ---- native front: tokens -> the four pyramid levels, no PyTorch op ----------
    def hip_front_ok(self, rows):
        """bf16 inference from normalised token ROWS (dinov2 ``intermediate_rows``):
        the four 1x1 projections and the two transposed convolutions are GEMMs over
        the token rows, the stride-2 and ``layerN_rn`` 3x3 convolutions run on the
        MFMA conv kernel; see ``_hip_front``."""
        s = self.scratch
        rl = self.resize_layers
        return (not self.use_clstoken and rows.is_cuda and rows.dtype == _half.dtype()
                and not torch.is_grad_enabled() and rows.shape[-1] % 64 == 0
                and isinstance(rl[0], nn.ConvTranspose2d) and rl[0].kernel_size == (4, 4)
                and isinstance(rl[1], nn.ConvTranspose2d) and rl[1].kernel_size == (2, 2)
                and isinstance(rl[2], nn.Identity) and isinstance(rl[3], nn.Conv2d)
                and rl[3].kernel_size == (3, 3) and rl[3].stride == (2, 2)
                and s.layer1_rn.out_channels % 64 == 0
                and all(getattr(s, 'refinenet%d' % i).hip_ok(rows) for i in (1, 2, 3, 4))
                and _hip_convs_ok(rows, s.output_conv1, s.output_conv2[0]))

    def _front_weights(self, dev):
        if '_hip_front_w' in self.__dict__:
            return self.__dict__['_hip_front_w']
        s = self.scratch
        levels = []
        for i in range(4):
            pj = self.projects[i]
            oc, d = pj.out_channels, pj.in_channels
            ocp = (oc + 63) // 64 * 64
            w1 = torch.zeros(ocp, d, device=dev)
            b1 = torch.zeros(ocp, device=dev)
            w1[:oc] = pj.weight.detach().float().view(oc, d)
            b1[:oc] = pj.bias.detach().float()
            lv = {'ocp': ocp, 'w1': w1.to(_half.dtype()).contiguous(), 'b1': b1}
            rl = self.resize_layers[i]
            if isinstance(rl, nn.ConvTranspose2d):
                # out[(s*y+i, s*x+j), co] = b[co] + sum_c in[(y,x), c] W[c, co, i, j]
                k = rl.kernel_size[0]
                w2 = torch.zeros(k, k, ocp, ocp, device=dev)
                w2[:, :, :oc, :oc] = rl.weight.detach().float().permute(2, 3, 1, 0)
                b2 = torch.zeros(k, k, ocp, device=dev)
                b2[:, :, :oc] = rl.bias.detach().float()
                lv.update(s=k, w2=w2.view(k * k * ocp, ocp).to(_half.dtype()).contiguous(),
                          b2=b2.view(-1).contiguous())
            elif isinstance(rl, nn.Conv2d):
                wc = torch.zeros(ocp, ocp, 3, 3, device=dev)
                wc[:oc, :oc] = rl.weight.detach().float()
                bc = torch.zeros(ocp, device=dev)
                bc[:oc] = rl.bias.detach().float()
                lv.update(s=1, wc=conv3d_ops.pack_weight2d(wc), bc=bc)
            else:
                lv['s'] = 1
            rn = getattr(s, 'layer%d_rn' % (i + 1))
            wr = torch.zeros(rn.out_channels, ocp, 3, 3, device=dev)
            wr[:, :oc] = rn.weight.detach().float()
            lv['wr'] = conv3d_ops.pack_weight2d(wr)
            lv['br'] = (rn.bias.detach().float().contiguous() if rn.bias is not None
                        else None)
            levels.append(lv)
        self.__dict__['_hip_front_w'] = levels
        return levels

    def _hip_front(self, taps, B, T, patch_h, patch_w):
        """taps: four bf16 [B*T, d] matrices of normalised tokens (class token first
        in every image).  -> the four ``layerN_rn`` outputs as PaddedImages."""
        dev = taps[0].device
        ins = self.__dict__.setdefault('_hip_in', {})

        def image(tag, C, Y, X):
            key = (tag, B, C, Y, X)
            if key not in ins:
                ins[key] = conv3d_ops.PaddedImage(B, C, Y, X, dev)
            return ins[key]
        outs = []
        for i, (rows, lv) in enumerate(zip(taps, self._front_weights(dev))):
            ocp, sc = lv['ocp'], lv['s']
            x = vit_ops.linear(rows, lv['w1'], lv['b1'], vit_ops.EPI_BF16)
            if 'w2' in lv:
                x = vit_ops.linear(x, lv['w2'], lv['b2'], vit_ops.EPI_BF16)
            img = conv3d_ops.tokens_to_image(x, T, T - patch_h * patch_w, patch_h, patch_w,
                                             sc, ocp, image(('lvl', i), ocp, sc * patch_h,
                                                            sc * patch_w))
            if 'wc' in lv:   # 3x3 stride 2: only the needed pixels (gathered DMA rows)
                img = conv3d_ops.conv2d_k3s2(
                    img, lv['wc'], None, lv['bc'],
                    out=image(('s2', i), ocp, (patch_h + 1) // 2, (patch_w + 1) // 2))
            Y, X = img.shape[2:]
            rn = conv3d_ops.conv2d_k3(img, lv['wr'], None, lv['br'],
                                      out=image(('rn', i), lv['wr'].shape[0], Y, X),
                                      out_relu=image(('rnr', i), lv['wr'].shape[0], Y, X))
            rn.relu_copy = image(('rnr', i), lv['wr'].shape[0], Y, X)
            outs.append(rn)
        return outs

    def hip_forward(self, taps, B, T, patch_h, patch_w):
        i1, i2, i3, i4 = self._hip_front(taps, B, T, patch_h, patch_w)
        return self._hip_refine_images(i1, i2, i3, i4, patch_h, patch_w)

    def _hip_refine(self, l1, l2, l3, l4, patch_h, patch_w):
        """Fusion blocks + output convs entirely in the padded channels-last
        bf16 layout of the MFMA conv kernel: four packs in, one unpack out."""
        ins = self.__dict__.setdefault('_hip_in', {})

        def packed(tag, t):
            key = (tag,) + tuple(t.shape)
            if key not in ins:
                ins[key] = conv3d_ops.PaddedImage(*t.shape, t.device)
            return conv3d_ops.pack_image(t, out=ins[key])
        i1, i2, i3, i4 = (packed(k, t) for k, t in enumerate((l1, l2, l3, l4)))
        return self._hip_refine_images(i1, i2, i3, i4, patch_h, patch_w)

    def _hip_refine_images(self, i1, i2, i3, i4, patch_h, patch_w):
        s = self.scratch
        ins = self.__dict__.setdefault('_hip_in', {})
        p4 = s.refinenet4.hip_block(i4, None, i3.shape[2:])
        p3 = s.refinenet3.hip_block(p4, i3, i2.shape[2:])
        p2 = s.refinenet2.hip_block(p3, i2, i1.shape[2:])
        p1 = s.refinenet1.hip_block(p2, i1, None)
        o = _hip_cache(self, 'output_conv1', s.output_conv1)(p1)
        size = (int(patch_h * 14), int(patch_w * 14))
        key = ('final',) + (o.shape[0], o.shape[1]) + size
        if key not in ins:
            ins[key] = conv3d_ops.PaddedImage(o.shape[0], o.shape[1], size[0], size[1],
                                              o.device)
        o = conv3d_ops.resize_bilinear(o, size, out=ins[key])
        oc2 = s.output_conv2
        o = _hip_cache(self, 'output_conv2', oc2[0])(o, relu=True)  # conv + ReLU
        tail = list(oc2)[2:]
        last = tail[0] if tail else None
        if (isinstance(last, nn.Conv2d) and last.out_channels == 1
                and last.kernel_size == (1, 1) and o.shape[1] in (32, 64)
                and len(tail) <= 2
                and (len(tail) == 1 or isinstance(tail[1], (nn.Sigmoid, nn.ReLU)))):
            # 1x1 conv to one channel + its activation straight off the padded rows
            if '_hip_tail' not in self.__dict__:
                self.__dict__['_hip_tail'] = (
                    last.weight.detach().float().reshape(-1).contiguous(),
                    float(last.bias.detach().float().item()) if last.bias is not None
                    else 0.0)
            w, b = self.__dict__['_hip_tail']
            act = 'none' if len(tail) == 1 else \
                ('sigmoid' if isinstance(tail[1], nn.Sigmoid) else 'relu')
            return conv3d_ops.image_dot(o, w, b, act)
        out = conv3d_ops.unpack_image(o, _half.dtype(), channels=oc2[0].out_channels)
        for layer in tail:
            out = layer(out)
        return out


_TAPS = {'vits': [2, 5, 8, 11], 'vitb': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}


@register_neck()
class DepthAnythingV2Adaptor(nn.Module):
    """``forward(x (B,3,H,W)) -> {'metric_depth': (B,H,W)}`` (dpt.py:256-263);
    kwargs as configs/veon/*: encoder, features, out_channels, max_depth,
    use_lora, lora_r."""

    def __init__(self, encoder='vitl', features=256,
                 out_channels=[256, 512, 1024, 1024], use_bn=False,
                 use_clstoken=False, max_depth=20.0, use_lora=True, lora_r=8):
        super().__init__()
        self.intermediate_layer_idx = dict(_TAPS)
        self.max_depth = max_depth
        self.encoder = encoder
        self.pretrained = DINOv2Adaptor(encoder, lora_r=lora_r if use_lora else -1)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn,
                                  out_channels=out_channels,
                                  use_clstoken=use_clstoken)
        # veon_amd extension: run the DPT head's convolutions (PyTorch/MIOpen)
        # under autocast at inference, e.g. torch.bfloat16 for BASELINE config 3
        # ("bf16").  None = fp32, the reference's numerics.
        self.head_dtype = None

    def encode(self, x):
        """The four intermediate (patch tokens, class token) pairs the head
        consumes (dpt.py:258-259)."""
        return self.pretrained.get_intermediate_layers(
            x, self.intermediate_layer_idx[self.encoder], return_class_token=True)

    def decode(self, feats, patch_h, patch_w):
        if (self.head_dtype is not None and feats[0][0].is_cuda
                and not torch.is_grad_enabled()):
            with torch.autocast('cuda', dtype=self.head_dtype):
                depth = self.depth_head(feats, patch_h, patch_w)
            depth = depth.float()
        else:
            depth = self.depth_head(feats, patch_h, patch_w)
        return depth * self.max_depth

    def forward(self, x):
        patch_h, patch_w = x.shape[-2] // 14, x.shape[-1] // 14
        if (self.head_dtype == _half.dtype() and x.is_cuda
                and not torch.is_grad_enabled() and not self.training):
            # native: the encoder hands over LayerNormed bf16 token rows of the four
            # taps, the head consumes them without a PyTorch op in between
            taps = self.pretrained.intermediate_rows(
                x, self.intermediate_layer_idx[self.encoder])
            if taps is not None and self.depth_head.hip_front_ok(taps[0]):
                B = x.shape[0]
                depth = self.depth_head.hip_forward(taps, B, taps[0].shape[0] // B,
                                                    patch_h, patch_w)
                return {'metric_depth': (depth.float() * self.max_depth).squeeze(1)}
        depth = self.decode(self.encode(x), patch_h, patch_w)
        return {'metric_depth': depth.squeeze(1)}

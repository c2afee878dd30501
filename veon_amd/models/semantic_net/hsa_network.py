"""High-resolution side adaptor (HSA) network -- mirror of
mmdet3d/models/semantic_net/side_adapter/highres_side_adaptor.py (SURVEY 8 row
f3, the reference-owned half; the timm side-adapter ViT is third-party).

It runs on the full-resolution image: 8x8 patches (11 264 tokens per camera at
512x1408), three ``HighresSideAdaptorBlock`` s and the ``AttnManipulateBlock``
that emits the dense attention biases for CLIP's tail blocks and the "supp"
features for the lift's fusion layer.  Its cost is the eight 3x3 convolutions
of the ``ConvBlock`` s (dim 384 on 64x176 maps: 179 GFLOP each, 1.4 TFLOP per
6-camera sample -- more than the Conv3d body).

Same class / parameter names as the reference; constructor arguments replace
the detectron2 config (``HighresSideAdaptorNetwork.from_veon_config`` holds the
values of configs/san_config.py:78-93).  With ``conv_dtype = torch.bfloat16`` (an
opt-in, default ``None`` = the reference's fp32 PyTorch) the ConvBlocks run on
the implicit-GEMM MFMA kernel in its 2-D mode with bias and GELU fused.
Pinned by vectors from the reference file (oracle/tools/gen_golden_hsa.py).
"""
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import conv3d_ops, vit_ops
from .._native_cache import NativeCacheMixin
from ... import half as _half


from .resize import interpolate as _interp


def _ln(mod, x, native):
    """``mod(x)`` for the token LayerNorms of the blocks; with ``native`` (the
    block runs its ConvBlock on the MFMA path) on the fp32 LayerNorm kernel."""
    if (native and isinstance(mod, nn.LayerNorm) and x.is_cuda
            and not torch.is_grad_enabled() and x.dtype == torch.float32
            and x.shape[-1] % 128 == 0 and x.shape[-1] <= 1024
            and mod.weight is not None and mod.bias is not None):
        return vit_ops.layernorm_f32(x.contiguous(), mod.weight.detach(),
                                     mod.bias.detach(), mod.eps)
    return mod(x)


class FeedForward(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip',)

    def __init__(self, dim, hidden_dim, out_dim=-1):
        super().__init__()
        out_dim = dim if out_dim == -1 else out_dim
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(),
                                 nn.Linear(hidden_dim, out_dim))
        self.conv_dtype = None   # set with the ConvBlocks (set_conv_dtype)

    def _hip_ok(self, x):
        _, fc1, _, fc2 = self.net
        return (self.conv_dtype == _half.dtype() and x.is_cuda and not self.training
                and not torch.is_grad_enabled() and fc1.in_features % 64 == 0
                and fc1.out_features % 64 == 0 and fc2.out_features % 4 == 0
                and x.dtype == torch.float32)

    def _hip_hidden(self, x):
        """LN -> GEMM(+bias, GELU) on the MFMA kernels -> bf16 (tokens, hidden)."""
        ln, fc1, _, fc2 = self.net
        if '_hip' not in self.__dict__:
            self.__dict__['_hip'] = (
                vit_ops.to_bf16(fc1.weight.detach().float()),
                fc1.bias.detach().float().contiguous(),
                vit_ops.to_bf16(fc2.weight.detach().float()),
                fc2.bias.detach().float().contiguous())
        w1, b1, w2, b2 = self.__dict__['_hip']
        h = vit_ops.layernorm(x.contiguous().view(-1, x.shape[-1]), ln.weight.detach(),
                              ln.bias.detach(), ln.eps)
        return vit_ops.linear(h, w1, b1, vit_ops.EPI_GELU), w2, b2

    def forward(self, x):
        if self._hip_ok(x):
            h, w2, b2 = self._hip_hidden(x)
            return vit_ops.linear(h, w2, b2).float().view(*x.shape[:-1], -1)
        return self.net(x)

    def forward_resized(self, x, side_shape, new_shape):
        """``interpolate(self(x) as a (B, out, H, W) map, size=new_shape,
        bilinear)`` -> contiguous (B, out, h, w).  The last Linear and the bilinear
        resize are both linear and act on different axes, so on the MFMA path the
        resize is applied to the hidden activations and the wide output GEMM runs
        on the h*w tokens that survive it (16x fewer at VEON's shapes; the
        (B, H*W, out) tensor -- 623 MB there -- is never formed).  Bilinear weights
        sum to one, so the bias commutes too."""
        B = x.shape[0]
        H, W = side_shape
        h, w = new_shape
        if not self._hip_ok(x):
            y = self(x).permute(0, 2, 1).reshape(B, -1, H, W)
            return _interp(y, size=(h, w), mode='bilinear').contiguous()
        hid, w2, b2 = self._hip_hidden(x)                       # (B*H*W, hidden) bf16
        hid = hid.view(B, H, W, -1).permute(0, 3, 1, 2)          # NCHW view of NHWC rows
        hid = F.interpolate(hid, size=(h, w), mode='bilinear')   # stays channels-last
        rows = hid.permute(0, 2, 3, 1).reshape(B * h * w, -1)
        y = vit_ops.linear(rows, w2, b2)                         # (B*h*w, out) bf16
        return y.view(B, h * w, -1).permute(0, 2, 1).float().contiguous().view(B, -1, h, w)


class ConvBlock(NativeCacheMixin, nn.Module):
    _native_cache = ('_hip',)

    """tokens (B, L, dim) on an H x W map: conv3x3 -> GELU -> LN -> conv3x3 -> LN
    (:31-52)."""

    def __init__(self, dim, hidden_dim, out_dim=-1):
        super().__init__()
        out_dim = dim if out_dim == -1 else out_dim
        self.conv1 = nn.Conv2d(dim, hidden_dim, stride=1, padding=1, kernel_size=3)
        self.gelu = nn.GELU()
        self.ln1 = nn.LayerNorm(hidden_dim)
        self.conv2 = nn.Conv2d(hidden_dim, out_dim, stride=1, padding=1, kernel_size=3)
        self.ln2 = nn.LayerNorm(out_dim)
        self.dim, self.h_dim, self.out_dim = dim, hidden_dim, out_dim
        self.conv_dtype = None

    def _hip_ok(self, x):
        return (self.conv_dtype == _half.dtype() and x.is_cuda
                and not torch.is_grad_enabled() and not self.training
                and self.dim % 64 == 0 and self.h_dim % 64 == 0
                and self.h_dim % 8 == 0 and self.out_dim % 8 == 0)

    def _hip_forward(self, x, size, residual=None, pre_ln=None):
        B, L, _ = x.shape
        H, W = size
        st = self.__dict__.setdefault('_hip', {})
        if 'w' not in st:
            st['w'] = [(conv3d_ops.pack_weight2d(c.weight),
                        c.bias.detach().float().contiguous())
                       for c in (self.conv1, self.conv2)]
            st['ln'] = [(ln.weight.detach().float().contiguous(),
                         ln.bias.detach().float().contiguous()) for ln in (self.ln1, self.ln2)]
        key = (B, H, W)
        if key not in st:
            dev = x.device
            st[key] = [conv3d_ops.PaddedImage(B, c, H, W, dev)
                       for c in (self.dim, self.h_dim, self.h_dim, self.out_dim)]
        a, b, c, d = st[key]
        (w1, b1), (w2, b2) = st['w']
        (g1, e1), (g2, e2) = st['ln']

        def interior(img):
            return img.rows.view(B, H + 2, W + 2, -1)[:, 1:-1, 1:-1]
        if (pre_ln is not None and x.dtype == torch.float32 and self.dim % 128 == 0
                and self.dim <= 1024):
            # the block's LayerNorm and the staging of the conv input in one pass
            conv3d_ops.layernorm_tokens_to_image(
                x.contiguous(), pre_ln.weight.detach(), pre_ln.bias.detach(), pre_ln.eps, a)
        else:
            if pre_ln is not None:
                x = pre_ln(x)
            # tokens are channels-last already: one strided copy into the interior
            interior(a).copy_(x.view(B, H, W, self.dim))
        conv3d_ops.conv2d_k3(a, w1, None, b1, act='gelu', out=b)
        # LayerNorm over the channels of every pixel on the padded rows (fp32
        # statistics): the first writes the next conv's padded input, the second
        # the fp32 tokens
        conv3d_ops.image_layernorm(b, g1, e1, self.ln1.eps, out=c)
        conv3d_ops.conv2d_k3(c, w2, None, b2, out=d)
        if residual is not None:
            residual = residual.contiguous().view(B, L, self.out_dim)
        return conv3d_ops.image_layernorm(d, g2, e2, self.ln2.eps, tokens=True,
                                          residual=residual)

    def forward(self, x, size=(1, 1), residual=None, pre_ln=None):
        """``pre_ln`` (an nn.LayerNorm): applied to ``x`` first; ``residual``
        (tokens of the output shape): added to the result.  Both fold into the
        first / last kernel on the MFMA path."""
        B, L, dim = x.shape
        H, W = size
        assert H * W == L
        if self._hip_ok(x) and (residual is None or (
                residual.dtype == torch.float32 and residual.shape[-1] == self.out_dim)):
            return self._hip_forward(x, size, residual, pre_ln)
        if pre_ln is not None:
            x = pre_ln(x)
        if residual is not None:
            return self.forward(x, size) + residual
        x = x.permute(0, 2, 1).reshape(B, dim, H, W).contiguous()
        x = self.gelu(self.conv1(x))
        x = self.ln1(x.reshape(B, self.h_dim, L).permute(0, 2, 1))
        x = x.permute(0, 2, 1).reshape(B, self.h_dim, H, W).contiguous()
        x = self.conv2(x)
        return self.ln2(x.reshape(B, self.out_dim, L).permute(0, 2, 1))


class HighresSideAdaptorBlock(nn.Module):
    """:108-135 -- x += ConvBlock(ln_3(x)); the last tokens += the (projected,
    nearest-resized) CLIP feature map; ln_4."""

    def __init__(self, dim, mlp_dim=960, neck_dim=0, pre_norm=False, use_add=False,
                 use_checkpoint=False):
        super().__init__()
        self.ff = ConvBlock(dim, mlp_dim)
        self.use_checkpoint = use_checkpoint
        self.neck_add = nn.Linear(neck_dim, dim, bias=False) \
            if neck_dim > 0 and use_add else nn.Identity()
        self.use_add = use_add
        self.pre_norm = nn.LayerNorm(dim) if pre_norm else nn.Identity()
        self.ln_3 = nn.LayerNorm(dim)
        self.ln_4 = nn.LayerNorm(dim)

    def forward(self, x, x_pos, ext, ext_pos, offset=None, offset_shape=(1, 1)):
        B, C_clip, h_ext, w_ext = ext.shape
        native = self.ff.conv_dtype == _half.dtype() and not self.training
        x = _ln(self.pre_norm, x, native)
        x = self.ff(x, offset_shape, residual=x, pre_ln=self.ln_3)
        if offset is not None:
            offset = self.neck_add(offset.reshape(B, C_clip, -1).permute(0, 2, 1))
            if (native and x.is_cuda and not torch.is_grad_enabled()
                    and isinstance(self.ln_4, nn.LayerNorm) and x.dtype == torch.float32
                    and offset.dtype == torch.float32 and x.shape[-1] % 128 == 0
                    and x.shape[-1] <= 1024
                    and offset_shape[0] * offset_shape[1] <= x.shape[1]):
                # nearest resize + add onto the last tokens + ln_4 in one pass
                return vit_ops.layernorm_f32_add_nearest(
                    x.contiguous(), offset.contiguous(), offset_shape, (h_ext, w_ext),
                    self.ln_4.weight.detach(), self.ln_4.bias.detach(), self.ln_4.eps)
            offset = _interp(offset.permute(0, 2, 1).reshape(B, -1, h_ext, w_ext),
                             size=offset_shape)
            offset = offset.reshape(B, offset.shape[1], -1).permute(0, 2, 1)
            x = torch.cat([x[:, :-offset.shape[1]], x[:, -offset.shape[1]:] + offset], 1)
        return _ln(self.ln_4, x, native)


class AttnManipulateBlock(nn.Module):
    """:138-193 -- ConvBlock, then two token-wise heads: per-layer / per-head
    embeddings whose Gram matrices are CLIP's dense attention biases
    (layers, B, heads, hw, hw), and the "supp" feature map for the lift."""

    def __init__(self, dim, mlp_dim=768, clip_dim=1024, heads=16, dim_head=64,
                 attn_layers=6, add_layers=2, supp_dim=384, pre_norm=False,
                 use_checkpoint=False):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        self.pre_norm = nn.LayerNorm(dim) if pre_norm else nn.Identity()
        self.ff = ConvBlock(dim, mlp_dim, mlp_dim)
        self.dim, self.mlp_dim, self.clip_dim = dim, mlp_dim, clip_dim
        self.add_layers, self.attn_layers = add_layers, attn_layers
        self.heads, self.dim_head = heads, dim_head
        self.attn_out = attn_layers * heads * dim_head
        self.head_attn = FeedForward(mlp_dim, mlp_dim, self.attn_out)
        self.head_supp = FeedForward(mlp_dim, mlp_dim, supp_dim)
        self.ln_3 = nn.LayerNorm(dim)
        self.ln_4 = nn.LayerNorm(mlp_dim)
        # veon_amd extension (inference only): emit the attention biases already
        # bordered for the class token, (layers, B, heads, hw + 1, hw + 1);
        # ClipRecHead.build_attn_bias passes such matrices through
        self.pad_class_token = False

    def forward(self, x, side_shape=(1, 1), new_shape=(1, 1)):
        native = self.ff.conv_dtype == _half.dtype() and not self.training
        x = _ln(self.pre_norm, x, native)
        x = _ln(self.ln_4, self.ff(x, side_shape, pre_ln=self.ln_3), native)
        supp = self.head_supp(x)
        H, W = side_shape
        h, w = new_shape
        B = x.shape[0]
        # head -> (B, C, H, W) map -> bilinear resize to CLIP's token grid; the
        # reference then reshapes the NCHW result to (B, h, w, -1) without a
        # permute -- kept as is (:177-178)
        attns = self.head_attn.forward_resized(x, (H, W), (h, w)).reshape(B, h, w, -1)
        attns = attns.reshape(B, h * w, self.attn_layers, self.heads, self.dim_head)
        # Gram matrices per (layer, sample, head): the reference's
        # einsum('bmahd,bnahd->bmnah').permute(3, 0, 4, 1, 2), i.e. (a, b, h, m, n),
        # formed as one batched matmul that writes that layout contiguously (the
        # einsum writes it with the layer / head axes innermost -- 36 floats apart
        # -- and every consumer then copies it)
        q = attns.permute(2, 0, 3, 1, 4)                       # (a, b, h, m, d)
        if self.pad_class_token and not torch.is_grad_enabled():
            # Gram matrices of the embeddings with a ZERO row in front: the zero row /
            # column of the class token that RecWithAttnbiasHead's bias builder
            # (clip_utils/visual.py:287-292) adds afterwards comes out of the same batched
            # matmul, instead of a zero fill and a strided copy of every (L+1)^2 matrix
            key = (tuple(q.shape), q.dtype, q.device)
            pad = self.__dict__.get('_qpad')
            if pad is None or pad[0] != key:
                a_, b_, h_, m_, d_ = q.shape
                pad = (key, torch.zeros(a_, b_, h_, m_ + 1, d_, dtype=q.dtype,
                                        device=q.device))
                self.__dict__['_qpad'] = pad
            qp = pad[1]
            qp[..., 1:, :].copy_(q)
            attns = torch.matmul(qp, qp.transpose(-1, -2))
        else:
            attns = torch.matmul(q, q.transpose(-1, -2))
        supp = supp.permute(0, 2, 1).reshape(B, -1, H, W)
        return None, attns, supp


class PatchEmbed(nn.Module):
    """:196-229."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 norm_layer=True, flatten=True, bias=True):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size,
                              stride=patch_size, bias=bias)
        self.norm = nn.LayerNorm(embed_dim) if norm_layer else nn.Identity()

    def _native_ok(self, x):
        p = self.patch_size
        return (x.is_cuda and not self.training and not torch.is_grad_enabled()
                and self.flatten and p[0] == p[1]
                and (self.proj.in_channels * p[0] * p[1]) % 64 == 0
                and self.proj.out_channels % 8 == 0 and x.dtype == torch.float32)

    def forward(self, x):
        if self._native_ok(x):
            # kernel = stride: the patches as bf16 GEMM rows, one MFMA GEMM onto a
            # zeroed fp32 token matrix (+ bias)
            from ... import vit_ops
            if '_hip' not in self.__dict__ or self.__dict__['_hip'] is None:
                d = self.proj.out_channels
                self.__dict__['_hip'] = (
                    vit_ops.to_bf16(self.proj.weight.detach().float().view(d, -1)),
                    None if self.proj.bias is None
                    else self.proj.bias.detach().float().contiguous())
            w, b = self.__dict__['_hip']
            B, _, H, W = x.shape
            p = self.patch_size[0]
            h, wd = H // p, W // p
            a = vit_ops.patchify(x, p, 0, w.shape[1])
            out = torch.zeros((B, h * wd, w.shape[0]), dtype=torch.float32, device=x.device)
            vit_ops.linear_residual_(out.view(B * h * wd, -1), a, w, b)
            return self.norm(out), (h, wd)
        x = self.proj(x)
        _, c, h, w = x.shape
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x), (h, w)

    def train(self, mode=True):
        self.__dict__['_hip'] = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__['_hip'] = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_hip'] = None
        return super()._apply(fn, *args, **kwargs)


class HighresSideAdaptorNetwork(nn.Module):
    """:232-300.  ``cr_map``: layer -> (cross-attention source, add source) CLIP
    layer indices (``FUSION_MAP`` strings 'i->j->k')."""

    def __init__(self, patch_embed, hsa_net_body, rear_block, cr_map,
                 use_checkpoint=False):
        super().__init__()
        self.patch_embed = patch_embed
        self.hsa_net_body = hsa_net_body
        self.rear_block = rear_block
        self.cr_map = cr_map
        self.use_checkpoint = use_checkpoint

    @classmethod
    def build(cls, dim=384, clip_dim=768, mlp_dim=384, input_size=(512, 1408),
              patch_shape=(8, 8), num_heads=12, fusion_map=('0->3->3', '1->6->6', '2->9->9'),
              manip_dim_head=32, manip_attn_layers=6, manip_add_layers=2,
              manip_supp_dim=384):
        """``from_config`` (:245-283) with explicit arguments; the defaults are
        configs/san_config.py:78-93."""
        cr_map = {int(i): (int(j), int(k)) for i, j, k in
                  [x.split('->') for x in fusion_map]}
        patch_embed = PatchEmbed(input_size, patch_shape, embed_dim=dim, norm_layer=False)
        body = nn.ModuleList([
            HighresSideAdaptorBlock(dim=dim, neck_dim=clip_dim, mlp_dim=mlp_dim,
                                    pre_norm=(i == 0), use_add=cr_map[i][1] >= 0,
                                    use_checkpoint=True)
            for i in range(len(fusion_map))])
        rear = AttnManipulateBlock(dim=dim, mlp_dim=mlp_dim, clip_dim=clip_dim,
                                   heads=num_heads, dim_head=manip_dim_head,
                                   attn_layers=manip_attn_layers,
                                   add_layers=manip_add_layers, supp_dim=manip_supp_dim,
                                   pre_norm=False, use_checkpoint=True)
        return cls(patch_embed, body, rear, cr_map)

    def set_conv_dtype(self, dtype):
        """``torch.bfloat16``: ConvBlocks on the MFMA conv kernel at inference;
        ``None``: the reference's fp32 PyTorch convolutions."""
        for m in self.modules():
            if isinstance(m, (ConvBlock, FeedForward)):
                m.conv_dtype = dtype
        return self

    def forward(self, image, clip_features: Dict):
        x, (H, W) = self.patch_embed(image)
        h, w = clip_features[1].shape[2], clip_features[1].shape[3]
        for layer_id, blk in enumerate(self.hsa_net_body):
            ca_id, add_id = self.cr_map[layer_id]
            # (the cross-attention source is read for its shape only; the add source is
            # flattened to token rows by the block -- neither needs an NCHW copy)
            x = blk(x, None, clip_features[ca_id], None,
                    clip_features[add_id] if blk.use_add else None, (H, W))
        return self.rear_block(x, (H, W), (h, w))

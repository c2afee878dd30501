"""DINOv2 ViT encoder of DepthAnythingV2 -- mirror of
mmdet3d/models/depth_anything/dinov2.py and dinov2_layers/*.

Module / parameter names follow the reference (``patch_embed.proj``,
``cls_token``, ``pos_embed``, ``blocks.{i}.norm1|attn.qkv|attn.proj|ls1.gamma|
norm2|mlp.fc1|mlp.fc2|ls2.gamma``, ``norm``; LoRA ``lora_A`` / ``lora_B``), so
DepthAnythingV2 checkpoints load unchanged.

Two execution paths share the parameters:

* ``forward`` on a ROCm device in eval mode -> the MFMA block kernels
  (csrc/vit_block.hip): bf16 operands, fp32 accumulation, fp32 residual
  stream; weights are cast to bf16 once (LoRA merged, the q scaling folded into
  the qkv weights -- 64^-0.5 is a power of two, so that folding is exact);
* otherwise (CPU, training) -> the same arithmetic in plain fp32 torch ops,
  which is the reference's own formulation (attention.py:56-69, block.py:85-110).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import vit_ops


class LoRALinear(nn.Linear):
    """nn.Linear + low-rank update W + (B @ A) * alpha / r
    (dinov2_layers/lora_layers.py:91-152, lora_alpha = 1, merge on eval)."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1, bias=True):
        super().__init__(in_features, out_features, bias=bias)
        self.r = r
        self.lora_alpha = lora_alpha
        self.merged = False
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def train(self, mode=True):
        super().train(mode)
        if self.r > 0:
            if mode and self.merged:
                self.weight.data -= (self.lora_B @ self.lora_A) * self.scaling
                self.merged = False
            elif not mode and not self.merged:
                self.weight.data += (self.lora_B @ self.lora_A) * self.scaling
                self.merged = True
        return self

    def effective_weight(self):
        if self.r > 0 and not self.merged:
            return self.weight + (self.lora_B @ self.lora_A) * self.scaling
        return self.weight

    def forward(self, x):
        out = F.linear(x, self.weight, self.bias)
        if self.r > 0 and not self.merged:
            out = out + (x @ self.lora_A.t() @ self.lora_B.t()) * self.scaling
        return out


def _linear(in_f, out_f, bias, lora_r):
    if lora_r > 0:
        return LoRALinear(in_f, out_f, r=lora_r, bias=bias)
    return nn.Linear(in_f, out_f, bias=bias)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, lora_r=-1):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = _linear(dim, dim * 3, qkv_bias, lora_r)
        self.proj = _linear(dim, dim, proj_bias, lora_r)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 bias=True, lora_r=-1):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        r = lora_r if lora_r > 1 else -1  # the reference's `> 1` (mlp.py:30)
        self.fc1 = _linear(in_features, hidden_features, bias, r)
        self.fc2 = _linear(hidden_features, out_features, bias, r)
        self.act = nn.GELU()

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    """Pre-norm block: x += ls1(attn(norm1 x)); x += ls2(mlp(norm2 x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False,
                 proj_bias=True, ffn_bias=True, init_values=None, lora_r=-1):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, qkv_bias, proj_bias, lora_r)
        self.ls1 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), bias=ffn_bias, lora_r=lora_r)
        self.ls2 = LayerScale(dim, init_values) if init_values else nn.Identity()

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        hw = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        ps = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = hw, ps
        self.patches_resolution = (hw[0] // ps[0], hw[1] // ps[1])
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=ps, stride=ps)

    def forward(self, x):
        _, _, H, W = x.shape
        assert H % self.patch_size[0] == 0 and W % self.patch_size[1] == 0
        return self.proj(x).flatten(2).transpose(1, 2)


class _HipBlockWeights:
    """bf16 weights + fp32 vectors of one block, prepared for the kernels."""

    def __init__(self, blk):
        def eff(lin):
            return lin.effective_weight() if isinstance(lin, LoRALinear) else lin.weight

        def vec(t, n, dev):
            return (t.detach().float().contiguous() if t is not None
                    else torch.zeros(n, device=dev))
        a = blk.attn
        d = a.qkv.in_features
        dev = a.qkv.weight.device
        wq = eff(a.qkv).detach().float().clone()
        bq = vec(a.qkv.bias, 3 * d, dev).clone()
        # fold q * scale (attention.py:60) and log2(e): the attention kernel then works
        # in the exp2 domain without spending an instruction on the scores
        wq[:d] *= a.scale * vit_ops.LOG2E
        bq[:d] *= a.scale * vit_ops.LOG2E
        self.heads = a.num_heads
        self.w_qkv, self.b_qkv = vit_ops.to_bf16(wq), bq
        self.w_proj = vit_ops.to_bf16(eff(a.proj).detach().float())
        self.b_proj = vec(a.proj.bias, d, dev)
        self.w_fc1 = vit_ops.to_bf16(eff(blk.mlp.fc1).detach().float())
        self.b_fc1 = vec(blk.mlp.fc1.bias, blk.mlp.fc1.out_features, dev)
        self.w_fc2 = vit_ops.to_bf16(eff(blk.mlp.fc2).detach().float())
        self.b_fc2 = vec(blk.mlp.fc2.bias, d, dev)
        self.n1 = (blk.norm1.weight.detach().float().contiguous(),
                   blk.norm1.bias.detach().float().contiguous(), blk.norm1.eps)
        self.n2 = (blk.norm2.weight.detach().float().contiguous(),
                   blk.norm2.bias.detach().float().contiguous(), blk.norm2.eps)
        self.g1 = (blk.ls1.gamma.detach().float().contiguous()
                   if isinstance(blk.ls1, LayerScale) else None)
        self.g2 = (blk.ls2.gamma.detach().float().contiguous()
                   if isinstance(blk.ls2, LayerScale) else None)
        # the same tensors behind one native call per block (veon_vit_block)
        self.packed = vit_ops.BlockWeights(
            self.heads, self.n1, self.w_qkv, self.b_qkv, self.w_proj, self.b_proj,
            self.g1, self.n2, self.w_fc1, self.b_fc1, self.w_fc2, self.b_fc2,
            self.g2, vit_ops.EPI_GELU, q_log2=True)


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True,
                 ffn_bias=True, proj_bias=True, init_values=None,
                 num_register_tokens=0, interpolate_offset=0.1, lora_r=-1,
                 **unused):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.num_register_tokens = num_register_tokens
        self.interpolate_offset = interpolate_offset
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, embed_dim))
        assert num_register_tokens == 0, 'DepthAnythingV2 uses no register tokens'
        self.register_tokens = None
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias,
                  init_values, lora_r) for _ in range(depth)])
        self.chunked_blocks = False
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.use_hip = True           # MFMA path on ROCm devices in eval mode
        self._hip_weights = None
        self._pos_cache = {}
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ------------------------------------------------------------- tokens
    def interpolate_pos_encoding(self, x, w, h):
        """Bicubic resize of the patch position grid with the +0.1 offset
        (dinov2.py:181-212; ``w`` / ``h`` are the image's H / W there too)."""
        npatch = x.shape[1] - 1
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        # the resized grid depends only on (w, h) and the weights: torch's
        # bicubic kernel takes ~5 ms for 768 channels, more than the 12 MFMA
        # blocks together, so it is computed once per input size at inference
        frozen = not self.training and not torch.is_grad_enabled()
        key = (w, h, x.dtype, x.device)
        if frozen and key in self._pos_cache:
            return self._pos_cache[key]
        pos = self.pos_embed.float()
        dim = x.shape[-1]
        w0 = w // self.patch_size + self.interpolate_offset
        h0 = h // self.patch_size + self.interpolate_offset
        sq = math.sqrt(N)
        patch = F.interpolate(
            pos[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
            scale_factor=(float(w0) / sq, float(h0) / sq), mode='bicubic')
        assert int(w0) == patch.shape[-2] and int(h0) == patch.shape[-1]
        patch = patch.permute(0, 2, 3, 1).view(1, -1, dim)
        out = torch.cat((pos[:, :1], patch), dim=1).to(x.dtype)
        if frozen:
            self._pos_cache[key] = out
        return out

    def prepare_tokens_with_masks(self, x, masks=None):
        B, nc, w, h = x.shape
        x = self.patch_embed(x)
        if masks is not None:
            x = torch.where(masks.unsqueeze(-1),
                            self.mask_token.to(x.dtype).unsqueeze(0), x)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x, w, h)

    def _native_tokens(self, x):
        """``prepare_tokens_with_masks`` at inference as: position rows copied into
        the fp32 stream, patches gathered as bf16 GEMM rows (class-token slots zero),
        one MFMA GEMM accumulating conv-weight x patch + bias onto the stream.  Row 0
        of the copied block holds cls + pos[0] - bias, so the GEMM's bias lands it on
        cls + pos[0].  -> fp32 [B*T, d]."""
        B, _, H, W = x.shape
        p, d = self.patch_size, self.embed_dim
        key = ('tok', H, W, x.device)
        if key not in self._pos_cache:
            proj = self.patch_embed.proj
            k = proj.in_channels * p * p
            kpad = (k + 63) // 64 * 64
            wp = torch.zeros(d, kpad, device=x.device)
            wp[:, :k] = proj.weight.detach().float().view(d, k)
            bias = (proj.bias.detach().float().contiguous() if proj.bias is not None
                    else torch.zeros(d, device=x.device))
            T = 1 + (H // p) * (W // p)
            probe = torch.empty(1, T, d, device=x.device)
            base = self.interpolate_pos_encoding(probe, H, W).float()[0].clone()
            base[0] += self.cls_token.detach().float().view(d) - bias
            self._pos_cache[key] = (vit_ops.to_bf16(wp), bias, base.contiguous(), kpad)
        wp, bias, base, kpad = self._pos_cache[key]
        # always a fresh buffer: the blocks run in place on it, and for B == 1 an
        # expand().reshape() would be a VIEW of the cached rows
        s = base.repeat(B, 1)
        assert s.data_ptr() != base.data_ptr()
        a = vit_ops.patchify(x, p, 1, kpad)
        return vit_ops.linear_residual_(s, a, wp, bias)

    # ------------------------------------------------------------- blocks
    def invalidate_hip_cache(self):
        """Call after changing weights when the inference caches are in use."""
        self._hip_weights = None
        self._pos_cache = {}

    def train(self, mode=True):
        self._hip_weights = None
        self._pos_cache = {}
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_weights = None
        self._pos_cache = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .half()
        self._hip_weights = None
        self._pos_cache = {}
        return super()._apply(fn, *args, **kwargs)

    def _use_hip(self, x):
        return (self.use_hip and x.is_cuda and not self.training
                and not torch.is_grad_enabled()
                and self.embed_dim % 64 == 0
                and self.embed_dim // self.num_heads == 64)

    def _run_blocks(self, x, taps):
        """x [B,T,d] fp32 -> list of block outputs at indices ``taps`` (and
        the final stream)."""
        outs = []
        if self._use_hip(x):
            if self._hip_weights is None:
                self._hip_weights = [_HipBlockWeights(b) for b in self.blocks]
            B, T, d = x.shape
            s = x.float().contiguous().view(B * T, d).clone()
            ws = vit_ops.block_workspace(B, T, d, self._hip_weights[0].packed.mlp_dim,
                                         x.device)
            for i, w in enumerate(self._hip_weights):
                vit_ops.block_forward_(s, w.packed, B, T, ws)
                if i in taps:
                    outs.append(s.view(B, T, d).clone())
            return outs, s.view(B, T, d)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in taps:
                outs.append(x)
        return outs, x

    def intermediate_rows(self, x, taps):
        """Native twin of ``get_intermediate_layers(x, taps, norm=True)``: the final
        LayerNorm of every tapped block output, as bf16 ROWS [B*T, d] (class token
        first in every image) straight off the fp32 residual stream -- no clone of
        the stream, no fp32 normalised copy.  None when the MFMA path is off."""
        if not self._use_hip(x):
            return None
        if self._hip_weights is None:
            self._hip_weights = [_HipBlockWeights(b) for b in self.blocks]
        s = self._native_tokens(x)
        B, d = x.shape[0], self.embed_dim
        T = s.shape[0] // B
        ws = vit_ops.block_workspace(B, T, d, self._hip_weights[0].packed.mlp_dim, x.device)
        nw, nb = self.norm.weight.detach().float(), self.norm.bias.detach().float()
        outs = []
        for i, w in enumerate(self._hip_weights):
            vit_ops.block_forward_(s, w.packed, B, T, ws)
            if i in taps:
                outs.append(vit_ops.layernorm(s, nw, nb, self.norm.eps))
        return outs

    def forward_features(self, x, masks=None):
        x = self.prepare_tokens_with_masks(x, masks)
        _, x = self._run_blocks(x, ())
        xn = self.norm(x)
        return {'x_norm_clstoken': xn[:, 0], 'x_norm_regtokens': xn[:, 1:1],
                'x_norm_patchtokens': xn[:, 1:], 'x_prenorm': x, 'masks': masks}

    def get_intermediate_layers(self, x, n=1, reshape=False,
                                return_class_token=False, norm=True):
        """dinov2.py:299-323."""
        img = x
        tokens = self.prepare_tokens_with_masks(x)
        total = len(self.blocks)
        taps = list(range(total - n, total)) if isinstance(n, int) else list(n)
        outputs, _ = self._run_blocks(tokens, taps)
        assert len(outputs) == len(taps)
        if norm:
            outputs = [self.norm(o) for o in outputs]
        cls = [o[:, 0] for o in outputs]
        outputs = [o[:, 1:] for o in outputs]
        if reshape:
            B, _, w, h = img.shape
            outputs = [o.reshape(B, w // self.patch_size, h // self.patch_size, -1)
                       .permute(0, 3, 1, 2).contiguous() for o in outputs]
        if return_class_token:
            return tuple(zip(outputs, cls))
        return tuple(outputs)

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret['x_norm_clstoken'])


_ZOO = {'vits': dict(embed_dim=384, depth=12, num_heads=6),
        'vitb': dict(embed_dim=768, depth=12, num_heads=12),
        'vitl': dict(embed_dim=1024, depth=24, num_heads=16)}


def DINOv2Adaptor(model_name, lora_r=-1):
    """dinov2.py:420-436: patch 14, 518 px position grid, LayerScale 1.0."""
    if model_name not in _ZOO:
        raise NotImplementedError('%s (swiglu vitg is not used by VEON)' % model_name)
    return DinoVisionTransformer(img_size=518, patch_size=14, init_values=1.0,
                                 mlp_ratio=4, lora_r=lora_r, **_ZOO[model_name])


def DINOv2(model_name):
    return DINOv2Adaptor(model_name, lora_r=-1)

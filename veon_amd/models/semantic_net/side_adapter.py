"""SAN's region-wise side adapter network -- the 2-D mask-proposal branch of
SANInVeonTemporal (semantic_net/side_adapter/side_adaptor_in_veon.py:30-263,
timm_wrapper.py:8-84, layers.py:33-101; wiring san_in_veon_temporal.py:125-139,
176-186, 238-255).

A small ViT (``vit_w240n6d8_patch16``: width 240, 8 blocks, 6 heads, patch 16) runs
on the FULL-resolution image with 100 learned query tokens in front of the patch
tokens; CLIP feature maps are added in at blocks 0 / 1 / 2 / 3 (``AddFusion``); the
query and patch tokens of the last block feed ``MLPMaskDecoder``: mask proposals
(B, Q, h, w) and per-head attention biases (B, heads, Q, h, w) for the CLIP
recognition head.  The 3-D occupancy path takes nothing but a SHAPE from this branch
(``sem_embed_ds``), so it is optional in ``VeonOccupancyPath`` (``side_adapter=True``).
At inference on a ROCm device the eight blocks run on the MFMA kernels of
csrc/vit_block.hip (``_native_blocks``): width 240 and head_dim 40 do not fit their
64-wide tiles as they are, so the residual stream is padded to 256 columns and every
head to 64 (zero weight rows / columns, packed once; q still scaled by 40^-0.5, so the
softmax is unchanged), LayerNorm takes its statistics over the 240 real columns
(``veon_vit_layernorm_padded``).

Parameter names follow the reference (and timm's VisionTransformer for
``vit_model.*``: ``patch_embed.proj``, ``pos_embed``, ``blocks.N.norm1 / attn.qkv /
attn.proj / norm2 / mlp.fc1 / mlp.fc2``), so SAN checkpoints map one to one.  timm is
absent from the build image (and unpinned in the reference): the ViT block is a
restatement of timm's pre-norm block (LayerNorm eps 1e-6, qkv bias, GELU, ratio 4),
parity UNPINNED; the reference-owned parts (token layout, position-embedding resize,
fusion, mask decoder) are pinned by tests/golden/side_adapter_tiny.npz.
"""
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .resize import interpolate

from .fusion_layers import LayerNorm


class MLP(nn.Module):
    """layers.py:33-49: ``num_layers`` affine maps, ReLU after all but the last."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, affine_func=nn.Linear):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(affine_func(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = F.relu(layer(x))
        return self.layers[-1](x)


class AddFusion(nn.Module):
    """layers.py:75-101: x (N,L,C) += bilinear-resized 1x1 projection of the
    channel-LayerNormed CLIP map y (N,C',H,W)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.input_proj = nn.Sequential(LayerNorm(in_channels),
                                        nn.Conv2d(in_channels, out_channels, kernel_size=1))

    def forward(self, x, y, spatial_shape):
        y = interpolate(self.input_proj(y.contiguous()), size=spatial_shape,
                        mode='bilinear', align_corners=False)
        return x + y.permute(0, 2, 3, 1).reshape(x.shape)


class PatchEmbed(nn.Module):
    """timm_wrapper.py:8-45: conv patchify that also returns the patch grid."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size,
                              bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        h, w = x.shape[-2:]
        return self.norm(x.flatten(2).transpose(1, 2)), (h, w)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    """timm's pre-norm ViT block without LayerScale / drop-path (the SAN configs
    use none): x += attn(norm1(x)); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


def _pad64(n):
    return (n + 63) // 64 * 64


class _PaddedBlock:
    """One ``_Block`` packed for the MFMA kernels: width d -> dp = ceil64(d), every
    head hd -> 64, MLP width -> ceil64; zero rows / columns in the padding, q rows
    pre-scaled by hd^-0.5 (the attention kernel's contract).  Half-precision weights
    [out, in], fp32 vectors."""

    def __init__(self, blk, dev):
        from ... import vit_ops
        d = blk.norm1.weight.numel()
        H = blk.attn.num_heads
        hd = d // H
        dp, m = _pad64(d), blk.mlp.fc1.out_features
        mp = _pad64(m)
        f32 = dict(dtype=torch.float32, device=dev)
        wq = torch.zeros(3, H, 64, dp, **f32)
        bq = torch.zeros(3, H, 64, **f32)
        wq[:, :, :hd, :d] = blk.attn.qkv.weight.detach().float().view(3, H, hd, d)
        bq[:, :, :hd] = blk.attn.qkv.bias.detach().float().view(3, H, hd)
        wq[0] *= hd ** -0.5 * vit_ops.LOG2E     # exp2-domain attention (q_log2)
        bq[0] *= hd ** -0.5 * vit_ops.LOG2E
        wp = torch.zeros(dp, H, 64, **f32)
        wp[:d, :, :hd] = blk.attn.proj.weight.detach().float().view(d, H, hd)
        w1 = torch.zeros(mp, dp, **f32)
        w1[:m, :d] = blk.mlp.fc1.weight.detach().float()
        w2 = torch.zeros(dp, mp, **f32)
        w2[:d, :m] = blk.mlp.fc2.weight.detach().float()

        def vec(t, n):
            o = torch.zeros(n, **f32)
            o[:t.numel()] = t.detach().float().view(-1)
            return o
        self.d, self.dp, self.heads = d, dp, H
        self.n1 = (vec(blk.norm1.weight, dp), vec(blk.norm1.bias, dp), blk.norm1.eps)
        self.n2 = (vec(blk.norm2.weight, dp), vec(blk.norm2.bias, dp), blk.norm2.eps)
        self.w_qkv = vit_ops.to_bf16(wq.view(3 * H * 64, dp))
        self.b_qkv = bq.view(-1).contiguous()
        self.w_proj = vit_ops.to_bf16(wp.view(dp, H * 64))
        self.b_proj = vec(blk.attn.proj.bias, dp)
        self.w_fc1, self.b_fc1 = vit_ops.to_bf16(w1), vec(blk.mlp.fc1.bias, mp)
        self.w_fc2, self.b_fc2 = vit_ops.to_bf16(w2), vec(blk.mlp.fc2.bias, dp)

    def forward_(self, s, B, T):
        """x += attn(norm1(x)); x += mlp(norm2(x)) on the padded fp32 stream s
        [B*T, dp], in place: seven launches (LayerNorm, qkv GEMM, attention, proj GEMM
        onto the stream, LayerNorm, fc1 GEMM + GELU, fc2 GEMM onto the stream)."""
        from ... import vit_ops
        h = vit_ops.layernorm_padded(s, self.n1[0], self.n1[1], self.d, self.n1[2])
        qkv = vit_ops.linear(h, self.w_qkv, self.b_qkv)
        o = vit_ops.attention(qkv.view(B, T, -1), self.heads, q_log2=True)
        vit_ops.linear_residual_(s, o.view(B * T, -1), self.w_proj, self.b_proj)
        h = vit_ops.layernorm_padded(s, self.n2[0], self.n2[1], self.d, self.n2[2])
        u = vit_ops.linear(h, self.w_fc1, self.b_fc1, vit_ops.EPI_GELU)
        vit_ops.linear_residual_(s, u, self.w_fc2, self.b_fc2)
        return s


class SideAdapterViT(nn.Module):
    """The timm VisionTransformer as SAN leaves it (side_adaptor_in_veon.py:103-112:
    class token dropped from ``pos_embed``, output norm replaced by Identity)."""

    use_hip = True   # inference on a ROCm device: blocks on the MFMA kernels

    def __init__(self, img_size=640, patch_size=16, embed_dim=240, depth=8, num_heads=6):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = None
        self.pos_embed = nn.Parameter(
            torch.randn(1, self.patch_embed.num_patches, embed_dim) * 0.02)
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.Identity()
        self.__dict__['_packed'] = None

    def packed_blocks(self, dev):
        """The blocks packed for the native path (cached with the half flavour they
        were packed in; dropped by train() / load_state_dict() / .to())."""
        from ... import half as _half
        pk = self.__dict__['_packed']
        if pk is None or pk[0] != (str(dev), _half.name()):
            pk = ((str(dev), _half.name()), [_PaddedBlock(b, dev) for b in self.blocks])
            self.__dict__['_packed'] = pk
        return pk[1]

    def native_ok(self, x):
        return (self.use_hip and x.is_cuda and not self.training
                and not torch.is_grad_enabled())

    def invalidate_hip_cache(self):
        self.__dict__['_packed'] = None

    def train(self, mode=True):
        self.__dict__['_packed'] = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__['_packed'] = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_packed'] = None
        return super()._apply(fn, *args, **kwargs)


class MLPMaskDecoder(nn.Module):
    """side_adaptor_in_veon.py:30-89."""

    def __init__(self, *, in_channels, total_heads=1, total_layers=1, embed_channels=256,
                 mlp_channels=256, mlp_num_layers=3, rescale_attn_bias=False):
        super().__init__()
        self.total_heads, self.total_layers = total_heads, total_layers

        def dense(n, k):
            return nn.Conv2d(n, k, kernel_size=1)
        self.query_mlp = MLP(in_channels, mlp_channels, embed_channels, mlp_num_layers)
        self.pix_mlp = MLP(in_channels, mlp_channels, embed_channels, mlp_num_layers,
                           affine_func=dense)
        self.attn_mlp = MLP(in_channels, mlp_channels,
                            embed_channels * total_heads * total_layers, mlp_num_layers,
                            affine_func=dense)
        self.bias_scaling = nn.Linear(1, 1) if rescale_attn_bias else nn.Identity()

    def forward(self, query, x):
        query = self.query_mlp(query)                 # (B, Q, c)
        pix = self.pix_mlp(x)                         # (B, c, h, w)
        b, c, h, w = pix.shape
        mask_preds = torch.einsum('bqc,bchw->bqhw', query, pix)
        attn = self.attn_mlp(x).reshape(b, self.total_layers, self.total_heads, c, h, w)
        if torch.is_grad_enabled():
            attn_bias = torch.einsum('bqc,blnchw->blnqhw', query, attn)
        else:
            # the same contraction as a batched matmul over (b, l, n): the einsum lowers
            # to ONE skinny fp32 mm (0.3 ms at 6 x 12 heads x 100 queries x 16 x 44)
            attn_bias = torch.matmul(query[:, None, None], attn.flatten(4)).reshape(
                b, self.total_layers, self.total_heads, query.shape[1], h, w)
        if isinstance(self.bias_scaling, nn.Linear) and not torch.is_grad_enabled():
            # Linear(1, 1) over five million scalars: as the multiply-add it is (the
            # K = 1 GEMM torch makes of it takes 0.34 ms)
            attn_bias = attn_bias * self.bias_scaling.weight.view(()) + \
                self.bias_scaling.bias.view(())
        else:
            attn_bias = self.bias_scaling(attn_bias[..., None]).squeeze(-1)
        return mask_preds, [a.squeeze(1) for a in attn_bias.chunk(self.total_layers, dim=1)]


class RegionwiseSideAdapterNetwork(nn.Module):
    """side_adaptor_in_veon.py:91-263 (inference form: the last block's features
    only go to the mask decoder, :186-187)."""

    def __init__(self, vit_model, fusion_layers, mask_decoder, num_queries,
                 fusion_map: Dict[int, int], deep_supervision_idxs: List[int]):
        super().__init__()
        self.vit_model = vit_model
        self.num_queries = num_queries
        self.num_features = vit_model.num_features
        self.query_embed = nn.Parameter(torch.zeros(1, num_queries, self.num_features))
        self.query_pos_embed = nn.Parameter(torch.zeros(1, num_queries, self.num_features))
        nn.init.normal_(self.query_embed, std=0.02)
        nn.init.normal_(self.query_pos_embed, std=0.02)
        self.fusion_layers = fusion_layers
        self.fusion_map = fusion_map
        self.mask_decoder = mask_decoder
        self.deep_supervision_idxs = deep_supervision_idxs

    @classmethod
    def build(cls, clip_dim=768, image_size=640, width=240, depth=8, num_heads=6,
              num_queries=100, fusion_map=('0->0', '3->1', '6->2', '9->3'),
              deep_supervision_idxs=(8,), attn_heads=12, attn_layers=1, embed_channels=256,
              mlp_channels=256, mlp_num_layers=3, rescale_attn_bias=True):
        """``from_config`` (:135-178) with explicit arguments; defaults are
        configs/san_config.py:58-75."""
        vit = SideAdapterViT(image_size, 16, width, depth, num_heads)
        x2side = {int(j): int(i) for i, j in [x.split('->') for x in fusion_map]}
        fusion = nn.ModuleDict({'layer_%d' % tgt: AddFusion(clip_dim, width)
                                for tgt in x2side})
        dec = MLPMaskDecoder(in_channels=width, total_heads=attn_heads,
                             total_layers=attn_layers, embed_channels=embed_channels,
                             mlp_channels=mlp_channels, mlp_num_layers=mlp_num_layers,
                             rescale_attn_bias=rescale_attn_bias)
        return cls(vit, fusion, dec, num_queries, x2side, list(deep_supervision_idxs))

    def forward(self, image, clip_features):
        out_features, san_features = self.forward_features(image, clip_features)
        mask_preds, attn_biases = self.decode_masks(out_features)
        return mask_preds, attn_biases, san_features

    def decode_masks(self, features):
        if not self.training:
            features = [features[-1]]
        mask_preds, attn_biases = [], []
        for feature in features:
            m, a = self.mask_decoder(**feature)
            mask_preds.append(m)
            attn_biases.append(a)
        return mask_preds, attn_biases

    def forward_features(self, image, clip_features):
        vit = self.vit_model
        x, (h, w) = vit.patch_embed(image)
        L = x.shape[1]
        pos_embed = vit.pos_embed
        ori_h, ori_w = vit.patch_embed.grid_size
        if pos_embed.shape[1] != L:
            # (the resize costs more than a block: cached at inference, like CLIP's)
            inference = not (self.training or torch.is_grad_enabled())
            key = (h, w, pos_embed.device, pos_embed.dtype, pos_embed._version,
                   pos_embed.data_ptr())
            cache = self.__dict__.setdefault('_pos_resized', {})
            if inference and key in cache:
                pos_embed = cache[key]
            else:
                pos_embed = F.interpolate(
                    pos_embed.reshape(1, ori_h, ori_w, -1).permute(0, 3, 1, 2), size=[h, w],
                    mode='bicubic', align_corners=False).flatten(2).permute(0, 2, 1)
                if inference:
                    cache.clear()
                    cache[key] = pos_embed.detach()
        pos_embed = torch.cat(
            [self.query_pos_embed.expand(pos_embed.shape[0], -1, -1), pos_embed], dim=1)
        x = torch.cat([self.query_embed.expand(x.shape[0], -1, -1), x], dim=1)  # B, Q+L, C
        x = vit.norm_pre(x + pos_embed)
        x = self.fuse(0, x, clip_features, (h, w))
        if vit.native_ok(x):
            return self._native_blocks(x, pos_embed, clip_features, (h, w), L)
        outs, san_feats = [], []
        n_blocks = len(vit.blocks)
        for i, blk in enumerate(vit.blocks, start=1):
            x = self.fuse(i, blk(x), clip_features, (h, w))
            grid = x[:, -L:, ...].permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], h, w)
            if i in self.deep_supervision_idxs:
                outs.append({'query': x[:, :-L, ...], 'x': grid})
            san_feats.append(grid.contiguous())
            if i < n_blocks:
                x = x + pos_embed
        return outs, san_feats

    def _native_blocks(self, x, pos_embed, clip_features, hw, L):
        """The block loop of ``forward_features`` on the MFMA kernels: the tokens live
        in ONE padded fp32 stream [B*T, ceil64(width)] that the blocks update in place;
        the CLIP fusion and the position embedding are added onto its real columns."""
        vit = self.vit_model
        B, T, d = x.shape
        h, w = hw
        blocks = vit.packed_blocks(x.device)
        dp = blocks[0].dp
        s = torch.zeros((B * T, dp), dtype=torch.float32, device=x.device)
        xs = s.view(B, T, dp)[..., :d]          # the real columns, a view of the stream
        xs.copy_(x)
        outs, san_feats = [], []
        n_blocks = len(blocks)
        for i, blk in enumerate(blocks, start=1):
            blk.forward_(s, B, T)
            if i in self.fusion_map:            # AddFusion onto the patch tokens, in place
                layer = self.fusion_layers['layer_%d' % i]
                y = interpolate(layer.input_proj(clip_features[self.fusion_map[i]].contiguous()),
                                size=(h, w), mode='bilinear', align_corners=False)
                xs[:, -L:] += y.permute(0, 2, 3, 1).reshape(B, L, d)
            # (reshape alone would be a VIEW of the stream the next block overwrites)
            grid = xs[:, -L:].permute(0, 2, 1).reshape(B, d, h, w).contiguous()
            if i in self.deep_supervision_idxs:
                outs.append({'query': xs[:, :-L].clone(), 'x': grid})
            san_feats.append(grid)
            if i < n_blocks:
                xs += pos_embed
        return outs, san_feats

    def fuse(self, block_idx, x, clip_features, spatial_shape):
        if block_idx in self.fusion_map:
            src = self.fusion_map[block_idx]
            L = spatial_shape[0] * spatial_shape[1]
            x = torch.cat([x[:, :-L, ...],
                           self.fusion_layers['layer_%d' % block_idx](
                               x[:, -L:, ...], clip_features[src], spatial_shape)], dim=1)
        return x


def semantic_inference_2d_w_embed(mask_cls, mask_embed, mask_pred):
    """san_in_veon_temporal.py:238-255: class scores (softmax without the void
    class) and mask embeddings spread over the sigmoid mask proposals."""
    mask_cls = F.softmax(mask_cls, dim=-1)[..., :-1]
    mask_pred = mask_pred.sigmoid()
    semseg = torch.einsum('bqc,bqhw->bchw', mask_cls, mask_pred)
    semembed = torch.einsum('bqc,bqhw->bchw', mask_embed, mask_pred)
    return semseg, semembed


def semantic_branch_2d(side_net, rec_head, ov_classifier_weight, images, clip_feats):
    """The 2-D open-vocabulary branch of SANInVeonTemporal.forward
    (san_in_veon_temporal.py:123-139, 176-186) on one flattened camera batch
    ``images`` (B*N,3,H,W) and the CLIP feature dict of layers 0..K: mask proposals
    and attention biases from the side adapter, mask embeddings from the CLIP
    recognition head, class logits against the text embeddings, and the
    down-sampled / full-resolution semantic maps."""
    mask_preds, attn_biases, san_feats = side_net(images, clip_feats)
    mask_embs = [rec_head(clip_feats, ab, normalize=True) for ab in attn_biases]
    mask_logits = [torch.einsum('bqc,nc->bqn', e, ov_classifier_weight) for e in mask_embs]
    sem_seg_ds, sem_embed_ds = semantic_inference_2d_w_embed(mask_logits[-1], mask_embs[-1],
                                                             mask_preds[-1])
    up = F.interpolate(mask_preds[-1], size=images.shape[-2:], mode='bilinear',
                       align_corners=False)
    sem_seg = torch.einsum('bqc,bqhw->bchw', F.softmax(mask_logits[-1], dim=-1)[..., :-1],
                           up.sigmoid())
    return {'mask_preds': mask_preds[-1], 'attn_biases': attn_biases[-1],
            'mask_embs': mask_embs[-1], 'mask_logits': mask_logits[-1],
            'sem_seg_ds': sem_seg_ds, 'sem_embed_ds': sem_embed_ds, 'sem_seg': sem_seg,
            'san_features': san_feats}

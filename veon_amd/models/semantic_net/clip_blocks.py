"""CLIP ViT residual-attention blocks of SAN-CLIP on the MFMA kernels.

In the reference the CLIP transformer arithmetic is NOT in the tree: it lives in
third-party ``open_clip`` (``open_clip.transformer.ResidualAttentionBlock`` /
``VisionTransformer``; version unpinned anywhere in the reference -- SURVEY 8c),
and VEON only wraps it: ``FeatureExtractor`` runs ``conv1`` / class + position
embeddings / ``ln_pre`` / the first K ``resblocks``
(mmdet3d/models/semantic_net/clip_utils/visual.py:57-91) and
``update_remaining_clip_feats`` runs the tail blocks again over all tokens with
a dense additive ``attn_mask`` of shape (B*heads, L+1, L+1) (:258-285).
This module restates the published block
    x = x + attn(ln_1(x), attn_mask);  x = x + mlp(ln_2(x))
with ``nn.MultiheadAttention``'s packed ``in_proj_weight`` / ``in_proj_bias`` /
``out_proj`` and ``mlp.c_fc`` / ``mlp.c_proj`` parameter names (what an
``open_clip`` checkpoint holds), QuickGELU for OpenAI weights.
**Parity unpinned**: ``open_clip`` is not installed and the reference holds no
vectors for it; tests compare against ``torch.nn.MultiheadAttention`` semantics.
The 100-query cross attention with the extra "self" logit
(attn_helper.py:34-300) stays in PyTorch for now.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import vit_ops


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio=4.0, quick_gelu=True):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ls_1 = nn.Identity()
        self.ln_2 = nn.LayerNorm(d_model)
        width = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([
            ('c_fc', nn.Linear(d_model, width)),
            ('gelu', QuickGELU() if quick_gelu else nn.GELU()),
            ('c_proj', nn.Linear(width, d_model))]))
        self.ls_2 = nn.Identity()
        self.n_head = n_head
        self.quick_gelu = quick_gelu

    def forward(self, x, attn_mask=None):
        """x (L, N, D) sequence-first, attn_mask additive (N*heads, L, L) or
        (L, L), as open_clip passes it."""
        h = self.ln_1(x)
        a = self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        x = x + self.ls_1(a)
        return x + self.ls_2(self.mlp(self.ln_2(x)))


def cross_attn_with_self_bias(attn, query, key, value, attn_mask=None):
    """The SOS/query-token attention of SAN's ``RecWithAttnbiasHead``
    (mmdet3d/models/semantic_net/attn_helper.py:34-300, called from
    ``cross_attn_layer`` :303-314): queries (K, N, D) attend over the memory
    tokens (L, N, D) with an additive per-head bias (N*heads, K, L) and, in
    addition, over THEMSELVES through one extra logit q.k_self whose value is
    the query's own v projection -- a softmax over L + 1 logits per query.
    ``attn`` is an nn.MultiheadAttention holding the packed projections.
    PyTorch implementation (100 queries: not a hot loop)."""
    K, N, D = query.shape
    L = key.shape[0]
    H = attn.num_heads
    hd = D // H
    w, b = attn.in_proj_weight, attn.in_proj_bias
    q = F.linear(query, w[:D], b[:D]) * hd ** -0.5
    k = F.linear(key, w[D:2 * D], b[D:2 * D])
    v = F.linear(value, w[2 * D:], b[2 * D:])
    q_k = F.linear(query, w[D:2 * D], b[D:2 * D])   # the query as its own key
    q_v = F.linear(query, w[2 * D:], b[2 * D:])     # ... and value

    def heads(t):  # (T, N, D) -> (N*H, T, hd)
        return t.contiguous().view(t.shape[0], N * H, hd).transpose(0, 1)
    q, k, v, q_k, q_v = heads(q), heads(k), heads(v), heads(q_k), heads(q_v)
    logits = torch.bmm(q, k.transpose(1, 2))                   # (N*H, K, L)
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            logits = logits.masked_fill(attn_mask, float('-inf'))
        else:
            logits = logits + attn_mask
    self_logit = (q * q_k).sum(dim=-1, keepdim=True)           # (N*H, K, 1)
    p = F.softmax(torch.cat([logits, self_logit], dim=-1), dim=-1)
    out = torch.bmm(p[:, :, :-1], v) + p[:, :, -1:] * q_v      # (N*H, K, hd)
    out = out.transpose(0, 1).contiguous().view(K, N, D)
    return F.linear(out, attn.out_proj.weight, attn.out_proj.bias)


def cross_attn_layer(block, x, mem, attn_bias):
    """attn_helper.py:303-314: x (K,N,D) queries, mem (L,N,D), attn_bias
    (N*heads, K, L)."""
    q_x = block.ln_1(x)
    k_x = block.ln_1(mem)
    x = x + block.ls_1(cross_attn_with_self_bias(block.attn, q_x, k_x, k_x,
                                                 attn_mask=attn_bias))
    return x + block.ls_2(block.mlp(block.ln_2(x)))


class _HipClipWeights:
    def __init__(self, blk):
        a = blk.attn
        d = a.embed_dim
        scale = (d // a.num_heads) ** -0.5
        w = a.in_proj_weight.detach().float().clone()
        b = a.in_proj_bias.detach().float().clone()
        w[:d] *= scale       # F.multi_head_attention_forward scales q
        b[:d] *= scale
        self.heads = a.num_heads
        self.w_qkv, self.b_qkv = vit_ops.to_bf16(w), b.contiguous()
        self.w_proj = vit_ops.to_bf16(a.out_proj.weight.detach().float())
        self.b_proj = a.out_proj.bias.detach().float().contiguous()
        self.w_fc1 = vit_ops.to_bf16(blk.mlp.c_fc.weight.detach().float())
        self.b_fc1 = blk.mlp.c_fc.bias.detach().float().contiguous()
        self.w_fc2 = vit_ops.to_bf16(blk.mlp.c_proj.weight.detach().float())
        self.b_fc2 = blk.mlp.c_proj.bias.detach().float().contiguous()
        self.n1 = (blk.ln_1.weight.detach().float().contiguous(),
                   blk.ln_1.bias.detach().float().contiguous(), blk.ln_1.eps)
        self.n2 = (blk.ln_2.weight.detach().float().contiguous(),
                   blk.ln_2.bias.detach().float().contiguous(), blk.ln_2.eps)
        self.act = vit_ops.EPI_QUICKGELU if blk.quick_gelu else vit_ops.EPI_GELU
        self.packed = vit_ops.BlockWeights(
            self.heads, self.n1, self.w_qkv, self.b_qkv, self.w_proj, self.b_proj,
            None, self.n2, self.w_fc1, self.b_fc1, self.w_fc2, self.b_fc2, None,
            self.act)


def run_blocks(blocks, x_lnd, attn_masks=None, cache=None):
    """Run ``blocks`` over x (L, N, D).  On a ROCm device in no-grad eval mode
    the MFMA kernels are used (one transpose in, one out); otherwise the torch
    blocks.  attn_masks: None or one additive mask per block, each
    (N*heads, L, L) / (N, heads, L, L) / (L, L).  Returns the list of per-block
    outputs (L, N, D)."""
    L, N, D = x_lnd.shape
    hip = (x_lnd.is_cuda and not torch.is_grad_enabled()
           and not any(b.training for b in blocks)
           and D % 64 == 0 and all(D // b.n_head == 64 for b in blocks))
    outs = []
    if not hip:
        x = x_lnd
        for i, blk in enumerate(blocks):
            m = None if attn_masks is None else attn_masks[i]
            if m is not None and m.dim() == 4:
                m = m.reshape(-1, L, L)
            x = blk(x, attn_mask=m)
            outs.append(x)
        return outs
    if cache is None:
        cache = {}
    # private batch-major fp32 copy of the stream (the kernels update it in place)
    s = torch.empty((N, L, D), dtype=torch.float32, device=x_lnd.device)
    s.copy_(x_lnd.permute(1, 0, 2))
    s = s.view(N * L, D)
    ws = None
    for i, blk in enumerate(blocks):
        w = cache.get(id(blk))
        if w is None:
            w = cache[id(blk)] = _HipClipWeights(blk)
        m = None if attn_masks is None else attn_masks[i]
        if m is not None:
            m = m.float()
            if m.dim() == 2:
                m = m.view(1, 1, L, L)
            elif m.dim() == 3:
                m = m.view(N, -1, L, L)
            m = m.contiguous()
        if ws is None:
            ws = vit_ops.block_workspace(N, L, D, w.packed.mlp_dim, s.device)
        vit_ops.block_forward_(s, w.packed, N, L, ws, m)
        outs.append(s.view(N, L, D).permute(1, 0, 2).contiguous())
    return outs


class ClipVisualTrunk(nn.Module):
    """conv1 patchify + class / position embeddings + ln_pre + resblocks: what
    ``FeatureExtractor.forward`` runs (clip_utils/visual.py:57-91), with
    open_clip's VisionTransformer parameter names."""

    def __init__(self, image_size=224, patch_size=16, width=768, layers=12,
                 heads=12, mlp_ratio=4.0, quick_gelu=True):
        super().__init__()
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.patch_size = patch_size
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(
            scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.resblocks = nn.ModuleList(
            [ResidualAttentionBlock(width, heads, mlp_ratio, quick_gelu)
             for _ in range(layers)])
        self._hip_cache = {}
        self._pos_cache = {}

    def train(self, mode=True):
        self._hip_cache = {}
        self._pos_cache = {}
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_cache = {}
        self._pos_cache = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def _pos_embed(self, h, w):
        pe = self.positional_embedding
        if (h, w) == self.grid_size:
            return pe
        inference = not (self.training or torch.is_grad_enabled())
        key = (h, w, pe.device, pe.dtype)
        if inference and key in self._pos_cache:
            return self._pos_cache[key]
        cls, grid = pe[:1], pe[1:]
        grid = grid.reshape(1, self.grid_size[0], self.grid_size[1], -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(h, w), mode='bicubic', align_corners=False)
        out = torch.cat([cls, grid.permute(0, 2, 3, 1).reshape(h * w, -1)], 0)
        if inference:  # the resize costs more than a block at these sizes
            self._pos_cache[key] = out
        return out

    def _patchify(self, x):
        """conv1 (kernel = stride = patch): at inference the non-overlapping
        patches are gathered and multiplied as one GEMM -- the same sums without
        a convolution library call."""
        p = self.patch_size
        N, C, H, W = x.shape
        if (self.training or torch.is_grad_enabled() or H % p or W % p):
            x = self.conv1(x)
            _, _, h, w = x.shape
            return x.flatten(2).permute(0, 2, 1), (h, w)
        h, w = H // p, W // p
        cols = x.view(N, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(N * h * w, C * p * p)
        out = cols @ self.conv1.weight.view(self.conv1.out_channels, -1).t()
        return out.view(N, h * w, -1), (h, w)

    def tokens(self, x):
        x, (h, w) = self._patchify(x)
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self._pos_embed(h, w).to(x.dtype)
        return self.ln_pre(x).permute(1, 0, 2), (h, w)     # LND

    def forward(self, x, last_layer_idx=-1, attn_masks=None):
        """-> (list of per-block token tensors (L,N,D), (h, w))."""
        t, hw = self.tokens(x)
        blocks = list(self.resblocks if last_layer_idx == -1
                      else self.resblocks[:last_layer_idx])
        return [t] + run_blocks(blocks, t, attn_masks, self._hip_cache), hw

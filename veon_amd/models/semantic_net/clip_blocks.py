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
**Block parity unpinned**: ``open_clip`` is not installed and the reference holds
no vectors for it; tests compare against ``torch.nn.MultiheadAttention``
semantics.  What the reference itself owns around the block IS pinned by vectors
generated from its unmodified files (oracle/tools/gen_golden_clip_*.py): the
query-token cross attention with the extra "self" logit (attn_helper.py:34-300,
PyTorch here: 100 queries are not a hot loop), the trunk wiring
(``FeatureExtractor``) and the recognition head (``RecWithAttnbiasHead``).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import vit_ops
from .resize import interpolate


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
        # F.multi_head_attention_forward scales q; log2(e) rides along (exp2-domain
        # attention kernel, BlockWeights(q_log2=True))
        w[:d] *= scale * vit_ops.LOG2E
        b[:d] *= scale * vit_ops.LOG2E
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
            self.act, q_log2=True)


def hip_blocks_ok(blocks, x, D):
    """The condition of run_blocks' MFMA path."""
    return (x.is_cuda and not torch.is_grad_enabled()
            and not any(b.training for b in blocks)
            and D % 64 == 0 and all(D // b.n_head == 64 for b in blocks))


def run_blocks(blocks, x_lnd, attn_masks=None, cache=None, keep=None, stream=None):
    """Run ``blocks`` over x (L, N, D).  On a ROCm device in no-grad eval mode
    the MFMA kernels are used (one transpose in, one out); otherwise the torch
    blocks.  attn_masks: None or one additive mask per block, each
    (N*heads, L, L) / (N, heads, L, L) / (L, L).  Returns the list of per-block
    outputs (L, N, D).  ``keep``: indices of the blocks whose output the caller reads
    (the last one always is); the native path leaves the others as None instead of
    copying the stream out after every block."""
    L, N, D = x_lnd.shape
    hip = hip_blocks_ok(blocks, x_lnd, D)
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
    # private batch-major fp32 copy of the stream (the kernels update it in place);
    # ``stream``: the caller already holds one ((N, L, D) fp32, its to overwrite) of
    # which x_lnd is the (L, N, D) view
    if stream is not None:
        assert stream.shape == (N, L, D) and stream.dtype == torch.float32 \
            and stream.is_contiguous()
        s = stream
    else:
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
        if keep is None or i in keep or i == len(blocks) - 1:
            # (L, N, D)-shaped VIEW of a batch-major snapshot (the stream itself after
            # the last block): ClipRecHead._save's permute / reshape to an (N, C, h, w)
            # map is then a view too (channels-last strides) -- one copy per tapped
            # layer instead of two, none for the last
            snap = s.view(N, L, D) if i == len(blocks) - 1 else s.view(N, L, D).clone()
            outs.append(snap.permute(1, 0, 2))
        else:
            outs.append(None)
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

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .half()
        self._hip_cache = {}
        self._pos_cache = {}
        return super()._apply(fn, *args, **kwargs)

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
        if self.training or torch.is_grad_enabled():
            x = self.conv1(x)
            _, _, h, w = x.shape
            return x.flatten(2).permute(0, 2, 1), (h, w)
        h, w = H // p, W // p
        if H % p or W % p:   # a stride-p conv without padding never reads the remainder
            x = x[:, :, :h * p, :w * p].contiguous()     # (ViT-L/14 on 128 x 352: 9 x 25)
        cols = x.view(N, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(N * h * w, C * p * p)
        out = cols @ self.conv1.weight.view(self.conv1.out_channels, -1).t()
        return out.view(N, h * w, -1), (h, w)

    def tokens(self, x):
        x, (h, w) = self._patchify(x)
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self._pos_embed(h, w).to(x.dtype)
        return self.ln_pre(x).permute(1, 0, 2), (h, w)     # LND

    def _native_stream(self, x):
        """``tokens`` at inference as four launches: the position rows (+ the class
        embedding in row 0) repeated into a fresh fp32 stream, the patches gathered as
        bf16 GEMM rows (class-token slots zero), one MFMA GEMM accumulating conv-weight x
        patch onto the stream, ln_pre fp32 -> fp32.  -> ((N, L, D) fp32, (h, w)).
        (The patch product runs on bf16 operands here; everything after it does anyway.)"""
        N, C, H, W = x.shape
        p, D = self.patch_size, self.conv1.out_channels
        h, w = H // p, W // p
        key = ('tok', h, w, x.device)
        if key not in self._pos_cache:
            k = C * p * p
            kpad = (k + 63) // 64 * 64
            wp = torch.zeros(D, kpad, device=x.device)
            wp[:, :k] = self.conv1.weight.detach().float().view(D, k)
            base = self._pos_embed(h, w).detach().float().clone()
            base[0] += self.class_embedding.detach().float()
            self._pos_cache[key] = (vit_ops.to_bf16(wp), base.contiguous(), kpad)
        wp, base, kpad = self._pos_cache[key]
        s = base.repeat(N, 1)           # always a fresh buffer: the blocks run in place on it
        assert s.data_ptr() != base.data_ptr()
        vit_ops.linear_residual_(s, vit_ops.patchify(x, p, 1, kpad), wp, None)
        s = vit_ops.layernorm_f32(s, self.ln_pre.weight.detach(), self.ln_pre.bias.detach(),
                                  self.ln_pre.eps)
        return s.view(N, 1 + h * w, D), (h, w)

    def forward(self, x, last_layer_idx=-1, attn_masks=None, taps=None):
        """-> (list of per-block token tensors (L,N,D), (h, w)); entry 0 = the tokens
        after ``ln_pre``, entry i = the output of block i.  ``taps``: the entries the
        caller reads (None = all); untapped entries may come back as None."""
        blocks = list(self.resblocks if last_layer_idx == -1
                      else self.resblocks[:last_layer_idx])
        keep = None if taps is None else {i - 1 for i in taps if i >= 1}
        D = self.conv1.out_channels
        if (hip_blocks_ok(blocks, x, D) and not self.training and D % 128 == 0
                and D <= 1024 and x.dtype == torch.float32 and blocks):
            s, hw = self._native_stream(x)
            t = (s.clone() if taps is None or 0 in taps else s).permute(1, 0, 2)
            return [t] + run_blocks(blocks, t, attn_masks, self._hip_cache, keep,
                                    stream=s), hw
        t, hw = self.tokens(x)
        return [t] + run_blocks(blocks, t, attn_masks, self._hip_cache, keep), hw


class ClipRecHead(nn.Module):
    """Mirror of ``RecWithAttnbiasHead`` (clip_utils/visual.py:112-292) for the
    configuration VEON uses (``cross_attn=True``): the tail blocks
    ``resblocks[first_layer_idx:]`` of the CLIP visual transformer, ``ln_post``
    and ``proj``.

    * ``forward(features, attn_bias)``: the SOS/query tokens attend over the
      patch tokens of layer ``first_layer_idx`` through ``cross_attn_layer``
      (per-head additive bias + self logit) while the patch tokens run through
      the plain blocks; returns the projected query tokens (:164-216).
    * ``update_remaining_clip_feats(clip_outputs, offsets, attns)``: the tail
      blocks once more over all tokens, with feature offsets added before the
      first and the middle block and a dense additive (B*heads, L+1, L+1) mask
      per block; fills the per-layer outputs and ``clip_feat_proj`` (:258-285).
      On a ROCm device at inference the blocks run on the MFMA kernels
      (``run_blocks``).

    ``features`` / ``clip_outputs``: dict with ``[i]`` = (N,C,h,w) patch map and
    ``['%d_cls_token' % i]`` = (1,N,C), as ``ClipOutput`` in the reference.
    """

    def __init__(self, resblocks, ln_post, proj, first_layer_idx=0,
                 sos_token_format='cls_token', sos_token_num=1,
                 downsample_method='bilinear'):
        super().__init__()
        if first_layer_idx < 0:
            raise NotImplementedError('first_layer_idx < 0 is not implemented yet.')
        self.first_layer_idx = first_layer_idx
        self.resblocks = nn.ModuleList(list(resblocks)[first_layer_idx:])
        self.ln_post = ln_post
        self.proj = proj                       # (width, output_dim) parameter
        self.sos_token_format = sos_token_format
        self.sos_token_num = sos_token_num
        self.downsample_method = downsample_method
        if sos_token_format in ('learnable_token', 'pos_embedding'):
            self.sos_token = nn.Parameter(
                torch.randn(sos_token_num, 1, proj.shape[0]) * 0.02)
        self._hip_cache = {}

    def train(self, mode=True):
        self._hip_cache = {}
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_cache = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._hip_cache = {}
        return super()._apply(fn, *args, **kwargs)

    @staticmethod
    def _save(outputs, idx, tokens, hw):
        l, n, c = tokens.shape
        outputs[idx] = tokens[1:].permute(1, 2, 0).reshape(n, c, *hw)
        outputs['%d_cls_token' % idx] = tokens[0:1]

    def _build_attn_biases(self, attn_biases, target_shape):
        """[N, heads|1, num_sos, H, W] per entry -> (N*heads, num_sos, h*w)
        (:218-256, the cross_attn branch)."""
        out = []
        true_heads = self.resblocks[0].attn.num_heads
        for ab in attn_biases:
            n, num_head, num_sos, h, w = ab.shape
            ab = ab.reshape(n, num_head * num_sos, h, w)
            if self.downsample_method in ('bicubic', 'bilinear', 'nearest'):
                ab = interpolate(ab, size=target_shape, mode=self.downsample_method,
                                 align_corners=False)
            elif self.downsample_method == 'avg':
                ab = F.adaptive_avg_pool2d(ab, output_size=target_shape)
            else:
                ab = F.adaptive_max_pool2d(ab, output_size=target_shape)
            ab = ab.reshape(n, num_head, num_sos, *target_shape)
            assert num_head in (1, true_heads), 'num_head=%d is not supported.' % num_head
            if num_head == 1:
                ab = ab.repeat(1, true_heads, 1, 1, 1)
            out.append(ab.reshape(n * true_heads, num_sos, -1))
        if len(out) == 1:
            out = [out[0] for _ in self.resblocks]
        return out

    def forward(self, features, attn_bias, normalize=False, clip_outputs=None):
        k0 = self.first_layer_idx
        cls_token = features['%d_cls_token' % k0]        # 1,n,c
        pix = features[k0]                               # n,c,h,w
        n, c, h, w = pix.shape
        x = torch.cat([cls_token, pix.reshape(n, c, -1).permute(2, 0, 1)])
        if self.sos_token_format == 'cls_token':
            sos = cls_token.repeat(self.sos_token_num, 1, 1)
        elif self.sos_token_format == 'learnable_token':
            sos = self.sos_token.expand(-1, n, -1)
        else:
            sos = self.sos_token.expand(-1, n, -1) + cls_token
        biases = self._build_attn_biases(attn_bias, (h, w))
        blocks = list(self.resblocks)
        # the patch tokens run through the plain blocks (the last one only when its
        # output is saved): all at once, on the MFMA kernels where run_blocks can
        run = blocks if clip_outputs is not None else blocks[:-1]
        xs = [x] + (run_blocks(run, x, None, self._hip_cache) if run else [])
        for i, blk in enumerate(blocks):
            sos = cross_attn_layer(blk, sos, xs[i][1:], biases[i])
            if clip_outputs is not None:
                self._save(clip_outputs, i + k0 + 1, xs[i + 1], (h, w))
        sos = self.ln_post(sos.permute(1, 0, 2))
        if self.proj is not None:
            sos = sos @ self.proj
        if normalize:
            sos = F.normalize(sos, dim=-1)
        if clip_outputs is not None:
            clip_outputs['clip_feat_proj'] = torch.einsum(
                'bchw,cd->bdhw', clip_outputs[len(self.resblocks) + k0], self.proj)
            return sos, clip_outputs
        return sos

    @staticmethod
    def build_attn_bias(attn):
        """(B, heads, L, L) patch-to-patch bias -> (B*heads, L+1, L+1) with a
        zero row / column for the class token (:287-292)."""
        B, H, L, _ = attn.shape
        new = torch.zeros((B, H, L + 1, L + 1), device=attn.device, dtype=attn.dtype)
        new[:, :, 1:, 1:] = attn
        return new.reshape(B * H, L + 1, L + 1)

    def update_remaining_clip_feats(self, clip_outputs, offsets=None, attns=None,
                                    keep_layers=None):
        """``keep_layers`` (veon_amd extension): the CLIP layers the caller reads
        afterwards (None = every tail layer, as the reference saves them)."""
        k0 = self.first_layer_idx
        cls_token = clip_outputs['%d_cls_token' % k0]
        x = clip_outputs[k0]
        hw = tuple(x.shape[2:])
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(2, 0, 1)
        x = torch.cat([cls_token, x], dim=0)
        nblk = len(self.resblocks)
        # the offsets (added before block 0 and before block nblk // 2) split the
        # tail into runs of blocks that go to the MFMA path back to back
        cuts = [0, nblk] if offsets is None else sorted({0, nblk // 2, nblk})
        for a, b in zip(cuts[:-1], cuts[1:]):
            if offsets is not None:
                if a == 0:
                    x = torch.cat([x[:1], x[1:] + offsets[0].permute(1, 0, 2)], dim=0)
                if a == nblk // 2:
                    x = torch.cat([x[:1], x[1:] + offsets[1].permute(1, 0, 2)], dim=0)
                    if a > 0:
                        # reference behaviour kept: ClipOutput.save stores a VIEW
                        # of the block output, and the in-place offset add of the
                        # next iteration (:268-271) writes through it, so the
                        # saved map of the block before the middle carries the
                        # second offset too
                        self._save(clip_outputs, a + k0, x, hw)
            # (biases that already carry the class token's zero row / column --
            # AttnManipulateBlock.pad_class_token -- pass through)
            L1 = x.shape[0]
            masks = None if attns is None else \
                [attns[t].reshape(-1, L1, L1) if attns[t].shape[-1] == L1
                 else self.build_attn_bias(attns[t]) for t in range(a, b)]
            keep = None if keep_layers is None else \
                {t - a for t in range(a, b) if t + k0 + 1 in keep_layers}
            outs = run_blocks(list(self.resblocks)[a:b], x, masks, self._hip_cache, keep)
            for t, o in zip(range(a, b), outs):
                if o is not None and (keep_layers is None or t + k0 + 1 in keep_layers
                                      or t + 1 == nblk):
                    self._save(clip_outputs, t + k0 + 1, o, hw)
            x = outs[-1]
        clip_outputs['clip_feat_proj'] = torch.einsum(
            'bchw,cd->bdhw', clip_outputs[nblk + k0], self.proj)
        return clip_outputs

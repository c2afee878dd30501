#!/usr/bin/env python
"""Where does the CLIP trunk's time go at 6 x 177 tokens?"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import vit_ops  # noqa: E402
from veon_amd.models.semantic_net import ClipVisualTrunk  # noqa: E402
from veon_amd.models.semantic_net.clip_blocks import run_blocks  # noqa: E402
from tools.hotpath_bench import timeit  # noqa: E402

dev = 'cuda:0'
clip = ClipVisualTrunk(224, 16, 768, 12, 12).to(dev).eval()
x = torch.randn(6, 3, 128, 352, device=dev)
with torch.no_grad():
    print('trunk total        %.3f ms' % timeit(lambda: clip(x)))
    t, hw = clip.tokens(x)
    print('tokens()           %.3f ms' % timeit(lambda: clip.tokens(x)))
    blocks = list(clip.resblocks)
    print('run_blocks         %.3f ms' % timeit(lambda: run_blocks(blocks, t, None, clip._hip_cache)))
    L, N, D = t.shape
    s = t.permute(1, 0, 2).contiguous().float().view(N * L, D)
    ws = vit_ops.block_workspace(N, L, D, 3072, dev)
    packed = [clip._hip_cache[id(b)].packed for b in blocks]

    def only_blocks():
        for w in packed:
            vit_ops.block_forward_(s, w, N, L, ws)
    print('12 x block_forward_ %.3f ms' % timeit(only_blocks))
    w0 = packed[0]
    print('1 x block_forward_  %.3f ms' % timeit(lambda: vit_ops.block_forward_(s, w0, N, L, ws)))
    h = vit_ops.layernorm(s, *clip._hip_cache[id(blocks[0])].n1)
    cw = clip._hip_cache[id(blocks[0])]
    print('  layernorm   %.1f us' % (1e3 * timeit(lambda: vit_ops.layernorm(s, *cw.n1), 50)))
    qkv = vit_ops.linear(h, cw.w_qkv, cw.b_qkv)
    print('  qkv gemm    %.1f us' % (1e3 * timeit(lambda: vit_ops.linear(h, cw.w_qkv, cw.b_qkv), 50)))
    print('  attention   %.1f us' % (1e3 * timeit(lambda: vit_ops.attention(qkv.view(N, L, -1), 12), 50)))
    u = vit_ops.linear(h, cw.w_fc1, cw.b_fc1, cw.act)
    print('  fc1 gemm    %.1f us' % (1e3 * timeit(lambda: vit_ops.linear(h, cw.w_fc1, cw.b_fc1, cw.act), 50)))
    print('  fc2 gemm    %.1f us' % (1e3 * timeit(lambda: vit_ops.linear_residual_(s, u, cw.w_fc2, cw.b_fc2), 50)))
    print('  copy out    %.1f us' % (1e3 * timeit(lambda: s.view(N, L, D).permute(1, 0, 2).contiguous(), 50)))

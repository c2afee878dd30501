#!/usr/bin/env python
"""Attention kernel timing (DA-V2 / CLIP shapes) + correctness vs fp32 math."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import vit_ops  # noqa: E402
from tools.vit_bench import timeit, PEAK  # noqa: E402


LOG2Q = os.environ.get('VEON_ATT_RAW_Q') != '1'    # q arrives multiplied by log2(e) (the packers' form)


def ref(qkv, H, bias=None):
    B, T, _ = qkv.shape
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    if LOG2Q:
        q = q / 1.4426950408889634
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    return (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, T, H * 64)


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    for B, T, H, with_bias in ((6, 901, 12, False), (6, 901, 16, False), (6, 705, 12, False),
                               (6, 705, 12, True), (2, 64, 2, False), (1, 130, 3, True)):
        qkv = torch.randn(B, T, 3 * H * 64, device=dev) * 0.5
        if LOG2Q:
            qkv.view(B, T, 3, H * 64)[:, :, 0] *= 1.4426950408889634
        qkv = qkv.bfloat16()
        bias = torch.randn(B, H, T, T, device=dev) if with_bias else None
        got = vit_ops.attention(qkv, H, bias, q_log2=LOG2Q).float()
        want = ref(qkv, H, bias)
        err = ((got - want).norm() / want.norm()).item()
        us = timeit(lambda: vit_ops.attention(qkv, H, bias, q_log2=LOG2Q))
        fl = 4.0 * B * H * T * T * 64
        q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        tus = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(
            q, k, v, attn_mask=bias.bfloat16() if with_bias else None,
            scale=1.0 / 1.4426950408889634 if LOG2Q else 1.0))
        print('B%d T%d H%d bias=%d  rel_l2 %.2e  %8.1f us %6.1f TF/s (%4.1f%% peak) | SDPA %8.1f us %6.1f TF/s'
              % (B, T, H, with_bias, err, us, fl / us / 1e6, 100 * fl / us / 1e6 / PEAK,
                 tus, fl / tus / 1e6))


if __name__ == '__main__':
    main()

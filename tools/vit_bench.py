#!/usr/bin/env python
"""Per-kernel timing of the ViT block kernels at DepthAnythingV2 shapes
(6 images x 901 tokens) with TFLOP/s against the bf16 dense MFMA peak, next to
torch's own bf16 ops (hipBLASLt / SDPA) on the same shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import vit_ops  # noqa: E402
from veon_amd.models.depth_anything import dinov2  # noqa: E402
from veon_amd.graphs import GraphedCallable  # noqa: E402

PEAK = 2500.0  # TFLOP/s dense bf16, MI355X_MICROARCH.md


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    dev = 'cuda:0'
    B, T = 6, 901
    for name, d, H in (('vitb', 768, 12), ('vitl', 1024, 16)):
        M = B * T
        print('== %s: M=%d d=%d heads=%d' % (name, M, d, H))
        x = torch.randn(M, d, device=dev)
        a = torch.randn(M, d, device=dev).bfloat16()
        for label, N, K, epi in (('qkv', 3 * d, d, 0), ('proj+res', d, d, 3),
                                 ('fc1+gelu', 4 * d, d, 1), ('fc2+res', d, 4 * d, 3)):
            w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
            bias = torch.randn(N, device=dev)
            gamma = torch.ones(N, device=dev)
            ak = torch.randn(M, K, device=dev).bfloat16()
            res = torch.randn(M, N, device=dev)
            if epi == 3:
                f = lambda: vit_ops.linear_residual_(res, ak, w, bias, gamma)
            else:
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                f = lambda: vit_ops.linear(ak, w, bias, epi, out=out)
            us = timeit(f)
            fl = 2.0 * M * N * K
            tus = timeit(lambda: torch.nn.functional.linear(ak, w))
            print('  gemm %-9s %4dx%4dx%4d  %8.1f us  %6.1f TF/s (%4.1f%% peak) | torch bf16 linear %8.1f us %6.1f TF/s'
                  % (label, M, N, K, us, fl / us / 1e6, 100 * fl / us / 1e6 / PEAK, tus, fl / tus / 1e6))
        qkv = (torch.randn(B, T, 3 * d, device=dev) * 0.5).bfloat16()
        us = timeit(lambda: vit_ops.attention(qkv, H))
        fl = 4.0 * B * H * T * T * 64
        q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        tus = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0))
        print('  attention  B=%d T=%d H=%d      %8.1f us  %6.1f TF/s (%4.1f%% peak) | torch SDPA %8.1f us %6.1f TF/s'
              % (B, T, H, us, fl / us / 1e6, 100 * fl / us / 1e6 / PEAK, tus, fl / tus / 1e6))
        w = torch.ones(d, device=dev)
        us = timeit(lambda: vit_ops.layernorm(x, w, w))
        print('  layernorm                      %8.1f us  %6.0f GB/s' % (us, M * d * 6 / us / 1e3))
        # whole block + whole encoder
        enc = dinov2.DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=d,
                                           depth=12 if name == 'vitb' else 24,
                                           num_heads=H, init_values=1.0, lora_r=16).to(dev).eval()
        img = torch.randn(B, 3, 252, 700, device=dev)
        with torch.no_grad():
            us = timeit(lambda: enc.forward_features(img), iters=10)
            blk_fl = 2.0 * M * (4 * d * d + 8 * d * d) + 4.0 * B * H * T * T * 64
            nb = len(enc.blocks)
            print('  encoder (%d blocks, 6 imgs)    %8.1f us  %6.1f TF/s  -> %.1f 6-cam samples/s'
                  % (nb, us, nb * blk_fl / us / 1e6, 1e6 / us))
            g = GraphedCallable(lambda im: enc.forward_features(im)['x_prenorm'], (img,))
            gus = timeit(lambda: g(img), iters=20)
            print('  encoder, hipGraph replay       %8.1f us  %6.1f TF/s  -> %.1f 6-cam samples/s'
                  % (gus, nb * blk_fl / gus / 1e6, 1e6 / gus))
            enc.use_hip = False
            enc16 = enc.bfloat16()
            tus = timeit(lambda: enc16.forward_features(img.bfloat16()), iters=10)
            print('  torch bf16 eager encoder       %8.1f us  %6.1f TF/s' % (tus, nb * blk_fl / tus / 1e6))


if __name__ == '__main__':
    main()

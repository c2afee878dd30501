#!/usr/bin/env python
"""veon_vit_gemm tile configurations on the encoder shapes (M = 5406 tokens of a
6-camera sample): small-tile kernel (0) and the ring-kernel tiles (1..6) against
torch's bf16 linear (hipBLASLt), interleaved rounds, HIP-event timed, each checked
against an fp32 reference on the bf16-rounded operands.

    python tools/gemm_bench.py [rounds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import _lib, vit_ops  # noqa: E402

SHAPES = [('B qkv', 5406, 2304, 768), ('B proj', 5406, 768, 768), ('B fc1', 5406, 3072, 768),
          ('B fc2', 5406, 768, 3072), ('L qkv', 5406, 3072, 1024), ('L proj', 5406, 1024, 1024),
          ('L fc1', 5406, 4096, 1024), ('L fc2', 5406, 1024, 4096),
          ('clipB qkv', 1062, 2304, 768), ('clipB fc1', 1062, 3072, 768),
          ('hsa head', 16896, 384, 384)]
if os.environ.get('SPLITK_EMU'):   # split-K emulated as more rows with a shorter K
    SHAPES = [('L fc2 /2', 10812, 1024, 2048), ('L fc2 /4', 21624, 1024, 1024),
              ('B fc2 /2', 10812, 768, 1536), ('B fc2 /3', 16218, 768, 1024),
              ('L qkv /2', 10812, 3072, 512), ('B qkv /2', 10812, 2304, 384)]
NAMES = {0: 'small', 1: '256x256', 2: '192x192', 3: '128x192', 4: '128x256', 5: '256x192',
         6: '128x128', 7: '256x128', 8: 'w16-256x256', 9: 'w16-256x128', 10: 'w16-128x256', 11: 'g256x256', 12: 'g256x128', 13: 'g128x256',
         14: 'g128x128'}
if os.environ.get('GEMM_CFGS'):
    NAMES = {int(c): NAMES[int(c)] for c in os.environ['GEMM_CFGS'].split(',')}
CFGS = sorted(NAMES)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = 'cuda:0'
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K in SHAPES:
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).bfloat16()
        w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=dev, generator=g)
        ref = a.float() @ w.float().t() + b
        flops = 2.0 * M * N * K
        res = {}
        for cfg in CFGS:
            L.veon_gemm_ring_set(cfg)
            out = vit_ops.linear(a, w, b)
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            res[cfg] = [err]
        auto = None
        for _ in range(rounds):
            for cfg in CFGS:
                L.veon_gemm_ring_set(cfg)
                res[cfg].append(timeit(lambda: vit_ops.linear(a, w, b)))
            L.veon_gemm_ring_set(-1)
            t = timeit(lambda: vit_ops.linear(a, w, b))
            auto = t if auto is None else min(auto, t)
            res.setdefault('torch', [0.0]).append(
                timeit(lambda: torch.nn.functional.linear(a, w, b.bfloat16())))
        L.veon_gemm_ring_set(-1)
        sk = ''
        if L.veon_vit_gemm_splitk_plan(M, N, K, None) > 0:   # the residual form, split-K
            xr = torch.randn(M, N, device=dev, generator=g)
            need = L.veon_vit_gemm_splitk_plan(M, N, K, None)
            wsk = (torch.empty(need, dtype=torch.uint8, device=dev),
                   torch.zeros(1024, dtype=torch.int32, device=dev))
            t_sk = min(timeit(lambda: vit_ops.linear_residual_splitk_(xr, a, w, b, None, wsk))
                       for _ in range(rounds))
            t_pl = min(timeit(lambda: vit_ops.linear_residual_(xr, a, w, b, None))
                       for _ in range(rounds))
            sk = ' | resid: split-K %6.1f plain %6.1f' % (t_sk, t_pl)
        if os.environ.get('RESID_CFGS'):      # the residual epilogue under forced tiles
            xr = torch.randn(M, N, device=dev, generator=g)
            parts = []
            for cfg in CFGS:
                L.veon_gemm_ring_set(cfg)
                parts.append('%s %.1f' % (NAMES[cfg], min(
                    timeit(lambda: vit_ops.linear_residual_(xr, a, w, b, None))
                    for _ in range(rounds))))
            L.veon_gemm_ring_set(-1)
            sk += ' | resid by tile: ' + ' '.join(parts)
        line = '%-10s %5dx%4dx%4d |' % (name, M, N, K)
        for cfg in CFGS:
            t = min(res[cfg][1:])
            line += ' %s %6.1f%s' % (NAMES[cfg], t, '!' if res[cfg][0] > 1e-2 else '')
        tt = min(res['torch'][1:])
        best = min(CFGS, key=lambda c: min(res[c][1:]))
        line += ' | auto %6.1f | torch %6.1f | best %s %.0f TF/s' % (
            auto, tt, NAMES[best], flops / min(res[best][1:]) / 1e6) + sk
        print(line, flush=True)


if __name__ == '__main__':
    main()

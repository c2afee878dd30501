#!/usr/bin/env python
"""One GEMM shape / configuration in a loop (rocprofv3 --pmc target):
    python tools/gemm_case.py M N K cfg [resid|bf16] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import _lib, vit_ops  # noqa: E402

M, N, K, cfg = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else 'bf16'
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).bfloat16()
w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * K ** -0.5).bfloat16()
b = torch.randn(N, device=dev, generator=g)
x = torch.randn(M, N, device=dev, generator=g)
L = _lib.lib()
L.veon_gemm_ring_set(cfg)
for _ in range(iters):
    if mode == 'resid':
        vit_ops.linear_residual_(x, a, w, b, None)
    elif mode == 'torch':
        torch.nn.functional.linear(a, w, b.bfloat16())
    else:
        vit_ops.linear(a, w, b)
torch.cuda.synchronize()

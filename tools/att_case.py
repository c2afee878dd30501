#!/usr/bin/env python
"""One attention shape in a loop (rocprofv3 target):
    python tools/att_case.py B T H [bias] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import vit_ops  # noqa: E402

B, T, H = (int(v) for v in sys.argv[1:4])
with_bias = len(sys.argv) > 4 and sys.argv[4] == '1'
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = 'cuda:0'
torch.manual_seed(0)
qkv = (torch.randn(B, T, 3 * H * 64, device=dev) * 0.5).bfloat16()
bias = torch.randn(B, H, T, T, device=dev) if with_bias else None
for _ in range(iters):
    vit_ops.attention(qkv, H, bias, q_log2=True)
torch.cuda.synchronize()

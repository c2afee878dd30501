#!/usr/bin/env python
"""DPT head (PyTorch/MIOpen) timing: fp32, bf16 autocast, channels_last."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd.models import build_neck  # noqa: E402
from tools.hotpath_bench import timeit  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
dav2 = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0, use_lora=True, lora_r=16,
                       encoder='vitb', features=128, out_channels=[96, 192, 384, 768])).to(dev).eval()
x = torch.randn(6, 3, 252, 700, device=dev)
with torch.no_grad():
    feats = dav2.encode(x)
    feats = [(a.clone(), b.clone()) for a, b in feats]
    print('fp32                 %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    dav2.head_dtype = torch.bfloat16
    print('bf16 autocast        %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    dav2.head_dtype = torch.float16
    print('fp16 autocast        %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    dav2.depth_head.to(memory_format=torch.channels_last)
    dav2.head_dtype = torch.bfloat16
    print('bf16 + channels_last %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    dav2.head_dtype = None
    print('fp32 + channels_last %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))

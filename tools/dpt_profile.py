#!/usr/bin/env python
"""rocprofv3 target: the DPT head alone under bf16 autocast, 20 iterations."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd.models import build_neck  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
dav2 = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0, use_lora=True, lora_r=16,
                       encoder='vitb', features=128, out_channels=[96, 192, 384, 768])).to(dev).eval()
dav2.head_dtype = torch.bfloat16
x = torch.randn(6, 3, 252, 700, device=dev)
with torch.no_grad():
    feats = [(a.clone(), b.clone()) for a, b in dav2.encode(x)]
    for _ in range(3):
        dav2.decode(feats, 18, 50)
    torch.cuda.synchronize()
    for _ in range(20):
        dav2.decode(feats, 18, 50)
    torch.cuda.synchronize()

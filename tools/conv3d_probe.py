#!/usr/bin/env python
"""How fast is torch/MIOpen on the AlignNetOcc3D body conv (3x3x3, 256->256 on
8x100x100, align_net_occ3d.py:363-399)?  Baseline for an implicit-GEMM kernel."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.vit_bench import timeit  # noqa: E402

dev = 'cuda:0'
C = 256
x = torch.randn(1, C, 8, 100, 100, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.01
fl = 2.0 * 8 * 100 * 100 * C * C * 27
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for cl in (False, True):
        xx, ww = x.to(dt), w.to(dt)
        if cl:
            xx = xx.contiguous(memory_format=torch.channels_last_3d)
            ww = ww.contiguous(memory_format=torch.channels_last_3d)
        try:
            f = lambda: torch.nn.functional.conv3d(xx, ww, padding=1)
            us = timeit(f, iters=10)
            print('%-9s channels_last=%d  %9.1f us  %7.1f TF/s' % (str(dt)[6:], cl, us, fl / us / 1e6), flush=True)
        except Exception as e:  # noqa
            print(dt, cl, 'failed', str(e)[:100])
# the same contraction as one dense GEMM (upper bound for an implicit-GEMM kernel)
a = torch.randn(104040, 6912, device=dev, dtype=torch.bfloat16)
b = torch.randn(256, 6912, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: torch.nn.functional.linear(a, b), iters=10)
print('dense bf16 GEMM 104040x256x6912  %9.1f us  %7.1f TF/s' % (us, 2.0 * 104040 * 256 * 6912 / us / 1e6))
from veon_amd import vit_ops  # noqa: E402
out = torch.empty(104040, 256, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: vit_ops.linear(a, b, None, 0, out=out), iters=10)
print('veon k_gemm_bf16 same shape      %9.1f us  %7.1f TF/s' % (us, 2.0 * 104040 * 256 * 6912 / us / 1e6))

# ---- the implicit-GEMM kernel itself and the whole 4-block body
from veon_amd import conv3d_ops  # noqa: E402
from veon_amd.models.semantic_net import AlignBody3D  # noqa: E402
vol = conv3d_ops.pack(x)
wp = conv3d_ops.pack_weight(w)
out = vol.like()
sc = torch.ones(C, device=dev)
us = timeit(lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, relu=True, out=out), iters=20)
print('veon conv3d_k3 (BN+ReLU fused)   %9.1f us  %7.1f TF/s (%.1f%% of 2.5 PF)' % (us, fl / us / 1e6, fl / us / 1e6 / 25))
us = timeit(lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, resid=vol, relu=True, out=out), iters=20)
print('veon conv3d_k3 (+identity)       %9.1f us  %7.1f TF/s' % (us, fl / us / 1e6))
print('pack %.1f us  unpack %.1f us' % (timeit(lambda: conv3d_ops.pack(x, out=vol)), timeit(lambda: conv3d_ops.unpack(vol))))
body = AlignBody3D(256, 4).to(dev).eval()
with torch.no_grad():
    us = timeit(lambda: body(x), iters=10)
    print('AlignBody3D x4 blocks (HIP)      %9.1f us  %7.1f TF/s' % (us, 8 * fl / us / 1e6))
    body.use_hip = False
    tus = timeit(lambda: body(x), iters=3)
    print('AlignBody3D torch fp32 eager     %9.1f us' % tus)
    b16 = body.bfloat16()
    xb = x.bfloat16()
    tus = timeit(lambda: b16(xb), iters=3)
    print('AlignBody3D torch bf16 eager     %9.1f us' % tus)

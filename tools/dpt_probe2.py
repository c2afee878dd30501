#!/usr/bin/env python
"""DPT head: bf16 autocast on MIOpen vs the MFMA 2-D conv path; per-stage."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import _lib, conv3d_ops  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402
from veon_amd.models.depth_anything import dpt  # noqa: E402
from tools.hotpath_bench import timeit  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
dav2 = build_neck(dict(type='DepthAnythingV2Adaptor', max_depth=80.0, use_lora=True, lora_r=16,
                       encoder='vitb', features=128, out_channels=[96, 192, 384, 768])).to(dev).eval()
x = torch.randn(6, 3, 252, 700, device=dev)
with torch.no_grad():
    feats = [(a.clone(), b.clone()) for a, b in dav2.encode(x)]
    dav2.head_dtype = torch.bfloat16
    c0 = _lib.CALLS.get('veon_conv2d_k3_bf16', 0)
    dav2.decode(feats, 18, 50)
    print('conv2d calls per decode:', _lib.CALLS.get('veon_conv2d_k3_bf16', 0) - c0)
    print('MFMA convs   %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    ok = dpt._hip_convs_ok
    dpt._hip_convs_ok = lambda *a, **k: False
    print('MIOpen bf16  %.2f ms' % timeit(lambda: dav2.decode(feats, 18, 50)))
    dpt._hip_convs_ok = ok
    # single pieces at the big shapes
    for mt in (1, 2, 3, 4):     # forced tile heights (veon_conv_debug_set bits 8..15)
        _lib.lib().veon_conv_debug_set(mt << 8)
        print('forced mt = %d: head %.3f ms' % (mt, timeit(lambda: dav2.decode(feats, 18, 50))))
        for (B, Cin, Cout, H, W) in ((6, 128, 128, 72, 200), (6, 128, 128, 144, 400),
                                     (6, 128, 64, 144, 400), (6, 64, 32, 252, 700)):
            xx = torch.randn(B, Cin, H, W, device=dev).bfloat16()
            hc = dpt._HipConv(torch.nn.Conv2d(Cin, Cout, 3, 1, 1).to(dev))
            img = conv3d_ops.pack_image(xx)
            fl = 2.0 * B * H * W * Cin * Cout * 9
            t_c = timeit(lambda: hc(img, relu=True))
            print('   %s: conv %.1f us (%.0f TF/s)' % ((B, Cin, Cout, H, W), t_c * 1e3,
                                                       fl / t_c / 1e9))
    _lib.lib().veon_conv_debug_set(0)
    for (B, Cin, Cout, H, W) in ((6, 128, 128, 72, 200), (6, 128, 64, 144, 400), (6, 64, 32, 252, 700)):
        xx = torch.randn(B, Cin, H, W, device=dev).bfloat16()
        conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1).to(dev)
        hc = dpt._HipConv(conv)
        img = conv3d_ops.pack_image(xx)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        t_c = timeit(lambda: hc(img, relu=True))
        t_p = timeit(lambda: conv3d_ops.pack_image(xx, out=img))
        o = hc(img)
        t_u = timeit(lambda: conv3d_ops.unpack_image(o, torch.bfloat16, Cout))
        cb = conv.bfloat16()
        t_m = timeit(lambda: cb(xx))
        print('%s: conv %.1f us (%.0f TF/s) pack %.1f unpack %.1f | MIOpen bf16 %.1f us' % (
            (B, Cin, Cout, H, W), t_c * 1e3, fl / t_c / 1e9, t_p * 1e3, t_u * 1e3, t_m * 1e3))

#!/usr/bin/env python
"""AlignNetOcc3D Conv3d body on the HIP implicit-GEMM kernel at the VEON shape
(4 x ResBlock3D, 256 ch, 8x100x100): per-conv and whole-body time, TFLOP/s
against the dense bf16 MFMA peak.  `--torch` adds the MIOpen baselines (slow)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import conv3d_ops  # noqa: E402
from veon_amd.models.semantic_net import (AlignBody3D, PredHead3DOcc,  # noqa: E402
                                          PredHead3DSem)
from tools.vit_bench import timeit, PEAK  # noqa: E402


def main():
    dev = 'cuda:0'
    C, Z, Y, X = 256, 8, 100, 100
    torch.manual_seed(0)
    x = torch.randn(1, C, Z, Y, X, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * (27 * C) ** -0.5
    fl = 2.0 * Z * Y * X * C * C * 27
    vol = conv3d_ops.pack(x)
    wp = conv3d_ops.pack_weight(w)
    out = vol.like()
    sc = torch.ones(C, device=dev)
    us = timeit(lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, relu=True, out=out), iters=20)
    print('conv3d_k3 BN+ReLU        %8.1f us  %6.1f TF/s (%4.1f%% of %.0f)' % (us, fl / us / 1e6, 100 * fl / us / 1e6 / PEAK, PEAK))
    us = timeit(lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, resid=vol, relu=True, out=out), iters=20)
    print('conv3d_k3 BN+id+ReLU     %8.1f us  %6.1f TF/s (%4.1f%%)' % (us, fl / us / 1e6, 100 * fl / us / 1e6 / PEAK))
    from veon_amd import _lib
    for flags, what in ((1, 'A slab every 3rd tap'), (2, 'W slab once'), (3, 'both'),
                        (4, 'no MFMA'), (7, 'no MFMA, few loads')):
        _lib.lib().veon_conv_debug_set(flags)
        us = timeit(lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, relu=True, out=out), iters=20)
        print('  ablation %d (%s): %8.1f us' % (flags, what, us))
    _lib.lib().veon_conv_debug_set(0)
    print('pack %.1f us  unpack %.1f us' % (timeit(lambda: conv3d_ops.pack(x, out=vol)),
                                            timeit(lambda: conv3d_ops.unpack(vol))))
    body = AlignBody3D(C, 4).to(dev).eval()
    with torch.no_grad():
        us = timeit(lambda: body(x), iters=10)
        print('AlignBody3D 4 blocks     %8.1f us  %6.1f TF/s (%4.1f%%)  -> %.1f samples/s' % (
            us, 8 * fl / us / 1e6, 100 * 8 * fl / us / 1e6 / PEAK, 1e6 / us))
        occ = PredHead3DOcc(C, 2).to(dev).eval()
        sem = PredHead3DSem(C, 768).to(dev).eval()

        def body_heads():
            vol = body(x, return_volume=True)
            return occ(vol), sem(vol)
        us2 = timeit(body_heads, iters=10)
        print('body + occ/sem heads     %8.1f us  (heads %.1f us: 5 GEMMs on the padded rows + 2 unpacks)' % (us2, us2 - us))
        if '--torch' in sys.argv:
            body.use_hip = False
            print('torch fp32 eager body    %8.1f us' % timeit(lambda: body(x), iters=3))
            b16, xb = body.bfloat16(), x.bfloat16()
            print('torch bf16 eager body    %8.1f us' % timeit(lambda: b16(xb), iters=3))


if __name__ == '__main__':
    main()

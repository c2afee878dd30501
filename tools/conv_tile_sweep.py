#!/usr/bin/env python
"""Tile sweep of the 2-D / 3-D MFMA conv on the shapes the VEON-B path runs at 256x704
(veon_conv_debug_set bits 16..27 force the tile): time per launch for every
instantiated tile against the launcher's own choice.

    python tools/conv_tile_sweep.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import _lib, conv3d_ops  # noqa: E402
from tools.vit_bench import timeit  # noqa: E402

TILES = [(3, 4, 7), (4, 4, 4), (4, 4, 3), (4, 3, 3), (4, 3, 2), (4, 2, 4), (4, 2, 3),
         (4, 2, 2), (4, 2, 1), (8, 1, 2), (8, 1, 1)]
# (tag, B, Cin, Cout, Y, X, stride, act)
SHAPES = [('HSA 384->384 16x44 gelu', 6, 384, 384, 16, 44, 1, 'gelu'),
          ('DPT s2 768->768 18x50', 6, 768, 768, 18, 50, 2, None),
          ('DPT rn4 768->128 9x25', 6, 768, 128, 9, 25, 1, None),
          ('DPT rn3 768->128 18x50', 6, 768, 128, 18, 50, 1, None),
          ('DPT rn2 192->128 36x100', 6, 192, 128, 36, 100, 1, None),
          ('DPT rn1 128->128 72x200', 6, 128, 128, 72, 200, 1, None),
          ('HSA 384->384 32x88 (512x1408)', 6, 384, 384, 32, 88, 1, 'gelu'),
          ('DPT fuse 128->128 36x100 relu', 6, 128, 128, 36, 100, 1, 'relu'),
          ('DPT fuse 128->128 72x200 relu', 6, 128, 128, 72, 200, 1, 'relu'),
          ('DPT out1 128->64 144x400', 6, 128, 64, 144, 400, 1, None),
          ('body 256->256 8x100x100', 1, 256, 256, 100, 100, 3, 'relu')]


def main():
    dev = torch.device('cuda:0')
    L = _lib.lib()
    torch.manual_seed(0)
    for tag, B, Cin, Cout, Y, X, stride, act in SHAPES:
        if stride == 3:   # the 3-D body conv
            x = torch.randn(B, Cin, 8, Y, X, device=dev)
            w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * (27 * Cin) ** -0.5
            vin, wp = conv3d_ops.pack(x), conv3d_ops.pack_weight(w)
            out = vin.like(Cout)
            fn = lambda: conv3d_ops.conv3d_k3(vin, wp, None, None, act=act, out=out)  # noqa: E731
        else:
            x = torch.randn(B, Cin, Y, X, device=dev)
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5
            img, wp = conv3d_ops.pack_image(x), conv3d_ops.pack_weight2d(w)
            if stride == 2:
                out = conv3d_ops.PaddedImage(B, wp.shape[0], (Y + 1) // 2, (X + 1) // 2, dev)
                fn = lambda: conv3d_ops.conv2d_k3s2(img, wp, None, None, out=out, act=act)  # noqa: E731
            else:
                out = conv3d_ops.PaddedImage(B, wp.shape[0], Y, X, dev)
                fn = lambda: conv3d_ops.conv2d_k3(img, wp, None, None, out=out, act=act)  # noqa: E731
        L.veon_conv_debug_set(0)
        fn()
        torch.cuda.synchronize()
        ref = out.rows.clone()
        base = min(timeit(fn, iters=30) for _ in range(2))
        res = []
        for wm, wn, mt in TILES:
            if 64 * wn > max(64, wp.shape[0]) and wn > 1:
                continue
            L.veon_conv_debug_set((wm << 16) | (wn << 20) | (mt << 24))
            try:
                fn()
                torch.cuda.synchronize()
                ok = (torch.equal(out.rows, ref) or
                      (out.rows.float() - ref.float()).abs().max().item() < 0.05)
                us = min(timeit(fn, iters=30) for _ in range(2))
                res.append(('%dx%d' % (wm * 16 * mt, 64 * wn), us, ok))
            except Exception as e:  # an LDS size the launcher refuses, ...
                res.append(('%dx%d' % (wm * 16 * mt, 64 * wn), float('nan'), False))
        L.veon_conv_debug_set(0)
        best = min((r for r in res if r[2]), key=lambda r: r[1])
        print('%-32s auto %6.1f us | best %s %6.1f us | %s' % (
            tag, base, best[0], best[1],
            ' '.join('%s %.1f%s' % (n, u, '' if ok else '!') for n, u, ok in res)), flush=True)


if __name__ == '__main__':
    main()

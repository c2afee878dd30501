#!/usr/bin/env python
"""Where the worker workgroups of the row max-pool kernel spend their time (SV, dense,
bf16 padded): per-worker stamps from the kernel itself (debug bit 20).
    python tools/mp_prof.py [dbg-ablation-bits]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


def main():
    abl = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
    dev = torch.device('cuda:0')
    C = 256
    cs = lift_case(synthetic.GRID_VEON, (512, 1408), 6, C, str(dev))
    X, Y, Z = cs['gsize']
    vs = bp.build_voxel_table(cs['rb'], cs['st'], 1, Z * Y * X, attach=False)
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)
    fb = cs['feat_nhwc'].to(vol.rows.dtype)
    L = _lib.lib()
    if '--hot-rows' in sys.argv:     # every gather reads row 0: memory out of the picture
        cs['rf'] = torch.zeros_like(cs['rf'])
    for _ in range(3):
        bp.rows_maxpool(cs['depth'], fb, cs['rd'], cs['rf'], vs, (1, Z, Y, X, C), (2, 2, 2),
                        out_volume=vol, chunk_order=None)
    def run():
        bp.rows_maxpool(cs['depth'], fb, cs['rd'], cs['rf'], vs, (1, Z, Y, X, C), (2, 2, 2),
                        out_volume=vol, chunk_order=None)
    L.veon_pool_debug_set(abl | (1 << 20) | (1023 << 21))
    run()
    torch.cuda.synchronize()
    L.veon_pool_debug_set(0)
    n = 512
    buf = np.zeros(n * 8, dtype=np.uint64)
    _lib.check(L.veon_pool_prof_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), 'prof')
    p = buf.reshape(n, 8).astype(np.int64)
    t0 = p[:, 0].min()
    scan = (p[:, 1] - p[:, 0]) / 100.0
    life = (p[:, 2] - p[:, 0]) / 100.0
    start = (p[:, 0] - t0) / 100.0
    end = (p[:, 2] - t0) / 100.0
    short = p[:, 7] / 100.0
    print('workers %d: start %.1f..%.1f us, end median %.1f max %.1f us' %
          (n, start.min(), start.max(), np.median(end), end.max()))
    mhz = p[:, 3] / np.maximum(life, 1e-3)
    print('shader clock during the workers: median %.0f MHz (min %.0f, max %.0f)' % (np.median(mhz), mhz.min(), mhz.max()))
    print('scan     median %.2f max %.2f us' % (np.median(scan), scan.max()))
    print('lifetime median %.2f max %.2f us' % (np.median(life), life.max()))
    print('lists    median %d max %d; chain slots median %d max %d; segments median %d max %d'
          % (np.median(p[:, 4]), p[:, 4].max(), np.median(p[:, 5]), p[:, 5].max(),
             np.median(p[:, 6]), p[:, 6].max()))
    o = np.argsort(-life)[:8]
    for i in o:
        print('  worker %3d: life %.1f us (scan %.1f), lists %d chain slots %d segments %d'
              % (i, life[i], scan[i], p[i, 4], p[i, 5], p[i, 6]))
    trace(int(o[0]), abl, run, L)


def trace(worker, abl, run, L):
    L.veon_pool_debug_set(abl | (1 << 20) | (worker << 21))
    run()
    torch.cuda.synchronize()
    L.veon_pool_debug_set(0)
    b2 = np.zeros(256 * 8, dtype=np.uint64)
    _lib.check(L.veon_pool_prof_read2(b2.ctypes.data_as(ctypes.c_void_p), b2.nbytes), 'trace')
    t = b2.reshape(256, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    z = t[:, 0].min()
    print('worker %d, first sub-pass: seg  start  issued  token  row0  row15  done (us from the '
          'first start)  points  index-in-chain' % worker)
    for k, r in enumerate(t[:40]):
        print('   %3d  %6.2f  %6.2f  %6.2f  %6.2f  %6.2f  %6.2f   %2d  %d' % (
            (k,) + tuple((r[[0, 1, 2, 4, 5, 3]] - z) / 100.0) + (r[6], r[7])))


if __name__ == '__main__':
    main()

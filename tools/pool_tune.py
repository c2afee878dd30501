#!/usr/bin/env python
"""Sweep the row max-pool kernel's tuning knobs (worker workgroups, cold / warm
list lengths) on the SV workload:  python tools/pool_tune.py [SV]"""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from tools.poolbench import timeit  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'SV'
    grid, size, cams, C = {'S2': (synthetic.GRID_S2, (256, 704), 6, 80),
                           'SV': (synthetic.GRID_VEON, (512, 1408), 6, 256)}[tag]
    dev = torch.device('cuda:0')
    cs = lift_case(grid, size, cams, C, str(dev))
    depth, feat = cs['depth'], cs['feat_nhwc']
    rb, rd, rf, st, ln = (cs[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = cs['gsize']
    vpb = Z * Y * X
    shape = (1, Z, Y, X, C)
    L = _lib.lib()
    vs = bp.build_voxel_table(rb, st, 1, vpb, attach=False)
    fb = feat.bfloat16()
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)

    def run_b():
        bp.rows_maxpool(depth, fb, rd, rf, vs, shape, (2, 2, 2), out_volume=vol)

    def run_f():
        bp.rows_maxpool(depth, feat, rd, rf, vs, shape, (2, 2, 2))

    res = []
    for workers, cold, warm in itertools.product((512, 1024, 2048, 4096), (16, 32, 64),
                                                 (96, 128, 256, 512)):
        L.veon_pool_tune_set(workers, cold, warm)
        tb = np.median([timeit(run_b, 10) for _ in range(3)])
        tf = np.median([timeit(run_f, 10) for _ in range(3)])
        res.append((tb, tf, workers, cold, warm))
        print('workers %4d cold %3d warm %3d: bf16 padded %7.2f us   f32 %7.2f us' %
              (workers, cold, warm, tb, tf), flush=True)
    L.veon_pool_tune_set(0, 0, 0)
    print('best bf16:', sorted(res)[:5])
    print('best f32:', sorted(res, key=lambda r: r[1])[:5])


if __name__ == '__main__':
    main()

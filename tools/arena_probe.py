#!/usr/bin/env python
"""Placement of the 205 MB output inside big arenas: is 'fast' a property of a
large (physically contiguous?) allocation as a whole?"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    grid, size, cams, C = synthetic.GRID_S2, (256, 704), 6, 80
    case = lift_case(grid, size, cams, C, dev)
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    gsize = case['gsize']
    vpb = gsize[0] * gsize[1] * gsize[2]
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))
    plan = bp.build_plan(rb, st, 1, vpb)
    L = _lib.lib()
    s = _lib.stream_ptr(torch.device(dev))
    nbytes = vpb * C * 4

    def run(ptr):
        def f():
            r = L.veon_bev_pool_v2_fwd_fused(
                C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(rd),
                _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln), _lib.ptr(plan),
                ctypes.c_void_p(ptr), _lib.LAYOUT_BCZYX, s)
            assert r == 0
        return f

    for gb in (1, 1, 1, 1, 4, 8):
        arena = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        offs = list(range(0, (gb << 30) - nbytes, 224 << 20))[:24]
        ts = [timeit(run(base + o), 20) for o in offs]
        print('arena %d GiB @%#x: %s' % (gb, base, ' '.join('%.1f' % t for t in ts)), flush=True)
        globals().setdefault('_keep', []).append(arena)


if __name__ == '__main__':
    main()

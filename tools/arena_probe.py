#!/usr/bin/env python
"""Placement of the 205 MB output inside big arenas: is 'fast' a property of a
large (physically contiguous?) allocation as a whole?"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from veon_amd import _lib, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    grid, size, cams, C = synthetic.GRID_S2, (256, 704), 6, 80
    ranks, coor, rig, fr, gsize = helpers.oracle_ranks(grid, size, cams)
    rb, rd, rf, st, ln = ranks
    D = fr.shape[0]
    hf, wf = size[0] // 16, size[1] // 16
    depth, feat = synthetic.make_depth_feat(1, cams, D, C, hf, wf, 0)
    depth = depth.to(dev)
    feat = feat.permute(0, 1, 3, 4, 2).contiguous().to(dev)
    rb, rd, rf, st, ln = (torch.from_numpy(x).to(dev) for x in (rb, rd, rf, st, ln))
    vpb = int(gsize[2]) * int(gsize[1]) * int(gsize[0])
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

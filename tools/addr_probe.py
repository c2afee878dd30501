#!/usr/bin/env python
"""Does the placement of the output volume change the fused pool kernel's time?
Runs the product kernel (cached plan, S2) into (a) views of one arena at
different byte offsets and (b) separately allocated buffers.  Not a test."""
import ctypes
import os
import sys

import numpy as np
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
    Z, Y, X = int(gsize[2]), int(gsize[1]), int(gsize[0])
    vpb = Z * Y * X
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

    arena = torch.empty(nbytes + (64 << 20), dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    print('arena base %#x (mod 2MiB = %#x)' % (base, base % (2 << 20)))
    offs = [k << 20 for k in range(0, 36)] + [(k << 19) + (8 << 20) for k in range(-3, 4)]
    res = {}
    for rnd in range(2):
        for o in offs:
            res.setdefault(o, []).append(timeit(run(base + o)))
    for o in offs:
        a_ = base + o
        print('arena +%5.1f MiB  (addr>>20)&31=%2d  %s us' % (
            o / 2**20, (a_ >> 20) & 31, ' '.join('%6.2f' % t for t in res[o])))
    keep = []
    for i in range(8):
        pad = torch.empty((3 + 5 * i) << 20, dtype=torch.uint8, device=dev)
        b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        keep += [pad, b]
        t = [timeit(run(b.data_ptr())) for _ in range(2)]
        print('separate buf %d @%#x  %6.2f %6.2f us' % (i, b.data_ptr(), t[0], t[1]))
    # output of the product wrapper (allocator-reused block), as kbench measures it
    shape = (1, Z, Y, X, C)
    f = lambda: bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)
    o = f()
    print('wrapper out @%#x' % o.data_ptr())
    del o
    print('wrapper  %6.2f %6.2f us' % (timeit(f), timeit(f)))
    # fill for reference
    v = arena[:nbytes].view(torch.float32)
    print('zero_    %6.2f us' % timeit(lambda: v.zero_()))


if __name__ == '__main__':
    main()

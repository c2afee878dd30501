#!/usr/bin/env python
"""Is the placement bimodality of the S2 pool kernel a capacity effect of the 256 MiB
Infinity Cache?  The same kernel and ranks with C = 40 ... 112 channel planes (volume
102 ... 287 MB) into NBUF fresh allocations each; prints us per launch and per 100 MB.
Not a test."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from tools.kbench import timeit  # noqa: E402
from veon_amd import _lib, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


def main():
    dev = 'cuda:0'
    nbuf = int(os.environ.get('NBUF', 10))
    case = lift_case(synthetic.GRID_S2, (256, 704), 6, 80, dev)
    depth = case['depth']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = case['gsize']
    vpb = X * Y * Z
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))
    plan = bp.build_plan(rb, st, 1, vpb)
    L = _lib.lib()
    s = _lib.stream_ptr(torch.device(dev))
    for C in (40, 60, 80, 96, 112):
        feat = torch.randn(1, 6, 16, 44, C, device=dev)
        nbytes = vpb * C * 4

        def run(ptr):
            def f():
                r = L.veon_bev_pool_v2_fwd_fused(
                    C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(rd),
                    _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln), _lib.ptr(plan),
                    ctypes.c_void_p(ptr), _lib.LAYOUT_BCZYX, s)
                assert r == 0
            return f
        zplan = torch.zeros_like(plan)

        def run_empty(ptr):   # every tile empty: the kernel's store pattern, nothing else
            def f():
                r = L.veon_bev_pool_v2_fwd_fused(
                    C, 0, 1, vpb, _lib.ptr(depth), _lib.ptr(feat), _lib.ptr(rd),
                    _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln), _lib.ptr(zplan),
                    ctypes.c_void_p(ptr), _lib.LAYOUT_BCZYX, s)
                assert r == 0
            return f
        keep, ts, fills, stores = [], [], [], []
        for i in range(nbuf):
            b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            keep.append(b)
            ts.append(min(timeit(run(b.data_ptr()), 20) for _ in range(2)))
            stores.append(timeit(run_empty(b.data_ptr()), 20))
            v = b.view(torch.float32)
            fills.append(timeit(lambda: v.zero_(), 20))
        per = [t / (nbytes / 1e8) for t in ts]
        print('C=%3d  %6.1f MB: pool us %s | us per 100 MB min %.2f max %.2f | fill %.1f-%.1f us'
              % (C, nbytes / 1e6, ' '.join('%5.1f' % t for t in ts), min(per), max(per),
                 min(fills), max(fills)), flush=True)
        print('        store pattern only (all tiles empty): %s'
              % ' '.join('%5.1f' % t for t in stores), flush=True)
        del keep
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

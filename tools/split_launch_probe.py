#!/usr/bin/env python
"""Does splitting the S2 pool launch over channel groups remove the placement
bimodality?  (Volumes of 102 / 154 MB time the same on every allocation, 205 MB is
bimodal: tools/mall_probe.py.)  On NBUF fresh 205 MB allocations: the one C = 80 launch
against 2 x C = 40 and 4 x C = 20 launches writing the channel groups of the same
volume back to back (each with its own contiguous feature rows), outputs compared.
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
    nbuf = int(os.environ.get('NBUF', 8))
    C = 80
    case = lift_case(synthetic.GRID_S2, (256, 704), 6, C, dev)
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = case['gsize']
    vpb = X * Y * Z
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))
    plan = bp.build_plan(rb, st, 1, vpb)
    L = _lib.lib()
    s = _lib.stream_ptr(torch.device(dev))

    def launcher(groups):
        feats = [feat[..., lo:hi].contiguous() for lo, hi in groups]

        def make(ptr):
            def f():
                for (lo, hi), fg in zip(groups, feats):
                    r = L.veon_bev_pool_v2_fwd_fused(
                        hi - lo, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(fg),
                        _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                        _lib.ptr(plan), ctypes.c_void_p(ptr + 4 * lo * vpb),
                        _lib.LAYOUT_BCZYX, s)
                    assert r == 0
            return f
        return make
    variants = {'1 x 80': launcher([(0, 80)]),
                '2 x 40': launcher([(0, 40), (40, 80)]),
                '4 x 20': launcher([(0, 20), (20, 40), (40, 60), (60, 80)])}
    res = {k: [] for k in variants}
    keep, same = [], True
    for i in range(nbuf):
        b = torch.empty(vpb * C * 4, dtype=torch.uint8, device=dev)
        keep.append(b)
        ref = None
        for k, mk in variants.items():
            f = mk(b.data_ptr())
            b.zero_()
            f()
            torch.cuda.synchronize()
            got = b.view(torch.float32).clone()
            if ref is None:
                ref = got
            else:
                same = same and torch.equal(ref, got)
            res[k].append(min(timeit(f, 20) for _ in range(2)))
    for k, v in res.items():
        print('%s launches: us per volume %s | min %.1f max %.1f' % (
            k, ' '.join('%5.1f' % t for t in v), min(v), max(v)), flush=True)
    print('outputs equal:', same)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Is the placement sensitivity of the channels-first pool kernel a matter of how
its C plane streams (4*Z*Y*X bytes apart) fall onto HBM channels?  Runs the S2
kernel into several allocations with the planes padded apart by various amounts
(veon_bev_pool_v2_fwd_fused_strided).  Not a test."""
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
    grid, size, cams, C = synthetic.GRID_S2, (256, 704), 6, 80
    case = lift_case(grid, size, cams, C, dev)
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = case['gsize']
    vpb = X * Y * Z
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))
    plan = bp.build_plan(rb, st, 1, vpb)
    L = _lib.lib()
    s = _lib.stream_ptr(torch.device(dev))

    def run(ptr, stride):
        def f():
            r = L.veon_bev_pool_v2_fwd_fused_strided(
                C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(feat), _lib.FEAT_F32,
                _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                _lib.ptr(plan), ctypes.c_void_p(ptr), stride, s)
            assert r == 0
        return f

    pads = [0, 64, 256, 1024, 4096, 16384, 65536 + 64, 262144 + 1024]
    nbuf = int(os.environ.get('NBUF', 10))
    maxb = C * (vpb + max(pads)) * 4
    bufs = [torch.empty(maxb, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    # correctness of the strided variant once
    ref = torch.empty(C * vpb, dtype=torch.float32, device=dev)
    run(ref.data_ptr(), vpb)()
    o = bufs[0].view(torch.float32)[:C * (vpb + 1024)].view(C, vpb + 1024)
    run(o.data_ptr(), vpb + 1024)()
    torch.cuda.synchronize()
    assert torch.equal(o[:, :vpb].reshape(-1), ref), 'strided result differs'
    print('plane pad (floats): ' + ' '.join('%8d' % p for p in pads))
    for i, b in enumerate(bufs):
        ts = [timeit(run(b.data_ptr(), vpb + p), 20) for p in pads]
        print('buf %2d @%#x   ' % (i, b.data_ptr()) + ' '.join('%8.2f' % t for t in ts))


if __name__ == '__main__':
    main()

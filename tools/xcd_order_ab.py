#!/usr/bin/env python
"""Sweep of the tile order of the fused pool kernels on the GPU box: runs of 2^lg
consecutive tiles per XCD (veon_pool_debug_set((lg + 1) << 8)) against tile =
blockIdx (flag 16), interleaved rounds in one process, HIP-event timed (minimum
over the rounds, us), outputs compared bit for bit.

    python tools/xcd_order_ab.py [rounds] [lg,lg,...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


def timeit(fn, iters=40):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def cases_for(tag, dev):
    grid, size, cams, C = {'S2': (synthetic.GRID_S2, (256, 704), 6, 80),
                           'SV': (synthetic.GRID_VEON, (512, 1408), 6, 256)}[tag]
    cs = lift_case(grid, size, cams, C, str(dev))
    depth, feat = cs['depth'], cs['feat_nhwc']
    rb, rd, rf, st, ln = (cs[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    X, Y, Z = cs['gsize']
    vpb = Z * Y * X
    shape = (1, Z, Y, X, C)
    L = _lib.lib()
    s = _lib.stream_ptr(dev)
    out = torch.empty((1, C, Z, Y, X), dtype=torch.float32, device=dev)
    if tag == 'S2':
        plan = bp.build_plan(rb, st, 1, vpb, attach=False)

        def slab():
            _lib.check(L.veon_bev_pool_v2_fwd_fused_ex(
                C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(feat), bp._feat_code(feat),
                _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                _lib.ptr(plan), _lib.ptr(out), _lib.LAYOUT_BCZYX, s), 'slab_cf')
            return out
        return [('S2 k_pool_fused_cf f32', slab)]
    vs = bp.build_voxel_table(rb, st, 1, vpb, attach=False)
    fb = feat.bfloat16()
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)
    return [
        ('SV k_rows_fused_cf f32', lambda: bp.rows_forward(depth, feat, rd, rf, vs, shape, out=out)),
        ('SV k_rows_fused_cf bf16', lambda: bp.rows_forward(depth, fb, rd, rf, vs, shape, out=out)),
        ('SV k_rows_maxpool f32', lambda: bp.rows_maxpool(depth, feat, rd, rf, vs, shape, (2, 2, 2))),
        ('SV k_rows_maxpool bf16 padded',
         lambda: bp.rows_maxpool(depth, fb, rd, rf, vs, shape, (2, 2, 2), out_volume=vol).rows),
    ]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    lgs = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6]
    dev = torch.device('cuda:0')
    L = _lib.lib()
    flags = [16] + [(lg + 1) << 8 for lg in lgs]
    for tag in ('S2', 'SV'):
        for name, fn in cases_for(tag, dev):
            ref, same = None, True
            for fl in flags:
                L.veon_pool_debug_set(fl)
                r = fn()
                torch.cuda.synchronize()
                if ref is None:
                    ref = r.clone()
                else:
                    same = same and torch.equal(ref, r)
            ref = None
            t = {fl: [] for fl in flags}
            for _ in range(rounds):
                for fl in flags:
                    L.veon_pool_debug_set(fl)
                    t[fl].append(timeit(fn))
            L.veon_pool_debug_set(0)
            print('%-30s blockIdx %6.1f |' % (name, min(t[16])) +
                  ' '.join(' lg%d %6.1f' % (lg, min(t[(lg + 1) << 8])) for lg in lgs) +
                  ' | %s' % ('bit-exact' if same else 'MISMATCH'), flush=True)


if __name__ == '__main__':
    main()

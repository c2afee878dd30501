#!/usr/bin/env python
"""Pool-kernel microbench on the GPU box: the slab kernels (csrc/bev_pool_v2.hip)
against the row kernels (csrc/bev_pool_rows.hip) on one workload, interleaved
rounds in one process, HIP-event timed on the launch stream.  Every case is
checked bit-for-bit against the slab kernels' output first.

    python tools/poolbench.py [S2|SV] [rounds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


def timeit(fn, iters=30):
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


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'SV'
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3] if len(sys.argv) > 3 else ''
    grid, size, cams, C = {'S2': (synthetic.GRID_S2, (256, 704), 6, 80),
                           'SV': (synthetic.GRID_VEON, (512, 1408), 6, 256)}[tag]
    dev = torch.device('cuda:0')
    case = lift_case(grid, size, cams, C, str(dev))
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    D = case['D']
    hf, wf = size[0] // 16, size[1] // 16
    X, Y, Z = case['gsize']
    vpb = Z * Y * X
    shape = (1, Z, Y, X, C)
    alg = 4 * (cams * hf * wf * C + cams * D * hf * wf + 3 * rb.numel() +
               2 * st.numel() + vpb * C)
    alg_mp = alg - 4 * vpb * C + 4 * vpb * C // 8
    print('%s: P_kept=%d I=%d out=%.1f MB alg=%.1f MB alg_maxpool=%.1f MB' %
          (tag, rb.numel(), st.numel(), vpb * C * 4 / 1e6, alg / 1e6, alg_mp / 1e6))
    L = _lib.lib()
    s = _lib.stream_ptr(dev)
    plan = bp.build_plan(rb, st, 1, vpb, attach=False)
    rows = bp.build_row_table(rb, st, 1, vpb, X, attach=False)
    vs = bp.build_voxel_table(rb, st, 1, vpb, attach=False)
    feat_b = feat.bfloat16()
    out = torch.empty((1, C, Z, Y, X), dtype=torch.float32, device=dev)
    mp = torch.empty((1, C, Z // 2, Y // 2, X // 2), dtype=torch.float32, device=dev)
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)

    def slab_cf(f):
        def run():
            _lib.check(L.veon_bev_pool_v2_fwd_fused_ex(
                C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(f), bp._feat_code(f),
                _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                _lib.ptr(plan), _lib.ptr(out), _lib.LAYOUT_BCZYX, s), 'slab_cf')
        return run

    def slab_mp(f, padded):
        def run():
            fn = (L.veon_bev_pool_v2_fwd_maxpool_padded if padded
                  else L.veon_bev_pool_v2_fwd_maxpool_ex)
            _lib.check(fn(C, st.numel(), 1, Z, Y, X, 2, 2, 2, _lib.ptr(depth), _lib.ptr(f),
                          bp._feat_code(f), _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb),
                          _lib.ptr(st), _lib.ptr(ln), _lib.ptr(rows),
                          _lib.ptr(vol.rows if padded else mp), s), 'slab_mp')
        return run

    def rows_cf(f, variant, ds=True):
        def run():
            _lib.check(L.veon_bev_pool_v2_fwd_rows(
                C, 1, vpb, _lib.ptr(depth), _lib.ptr(f), bp._feat_code(f), _lib.ptr(rd),
                _lib.ptr(rf), _lib.ptr(vs), _lib.ptr(out), 0, f.numel(), variant, s), 'rows_cf')
        return run

    def rows_mp(f, padded, ds=True, dbg=0):
        def run():
            L.veon_pool_debug_set(dbg)
            _lib.check(L.veon_bev_pool_v2_fwd_rows_maxpool(
                C, 1, Z, Y, X, 2, 2, 2, _lib.ptr(depth), _lib.ptr(f), bp._feat_code(f),
                _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(vs),
                _lib.ptr(vol.rows if padded else mp), 1 if padded else 0, f.numel(), s),
                'rows_mp')
            L.veon_pool_debug_set(0)
        return run

    cases = [('fill(torch)', out.zero_, alg, None)]
    cases += [('slab_cf f32', slab_cf(feat), alg, 'cf32'),
              ('slab_cf bf16', slab_cf(feat_b), alg, 'cfb')]
    for v in range(4):
        cases.append(('rows_cf f32 v%d' % v, rows_cf(feat, v), alg, 'cf32'))
    cases.append(('rows_cf bf16 v0', rows_cf(feat_b, 0), alg, 'cfb'))
    cases.append(('rows_cf bf16 v1', rows_cf(feat_b, 1), alg, 'cfb'))
    cases += [('slab_mp f32', slab_mp(feat, False), alg_mp, 'mp32'),
              ('rows_mp f32', rows_mp(feat, False), alg_mp, 'mp32'),
              ('slab_mp bf16 padded', slab_mp(feat_b, True), alg_mp, 'mpb'),
              ('rows_mp bf16 padded', rows_mp(feat_b, True), alg_mp, 'mpb'),
              ('rows_mp bf16 ABL cold only', rows_mp(feat_b, True, False, 1), alg_mp, None),
              ('rows_mp bf16 ABL workers only', rows_mp(feat_b, True, False, 2), alg_mp, None),
              ('rows_mp bf16 ABL -warm', rows_mp(feat_b, True, False, 4), alg_mp, None),
              ('rows_mp bf16 ABL -hot', rows_mp(feat_b, True, False, 8), alg_mp, None),
              ('rows_mp bf16 ABL onlyhot', rows_mp(feat_b, True, False, 6), alg_mp, None),
              ('rows_mp bf16 ABL onlywarm', rows_mp(feat_b, True, False, 10), alg_mp, None),
              ('rows_mp f32 ABL cold only', rows_mp(feat, False, False, 1), alg_mp, None),
              ('rows_mp f32 ABL workers only', rows_mp(feat, False, False, 2), alg_mp, None)]
    if only:
        cases = [c for c in cases if only in c[0]]
    # parity between the two families (the oracle comparison lives in tests/)
    refs = {}
    for name, fn, _, key in cases:
        if key is None:
            continue
        out.fill_(float('nan'))
        mp.fill_(float('nan'))
        fn()
        torch.cuda.synchronize()
        got = {'c': out, 'm': (vol.rows if key == 'mpb' else mp)}[key[0]].clone()
        if key not in refs:
            refs[key] = got
            print('%-22s reference for %s' % (name, key))
        else:
            print('%-22s %s' % (name, 'bit-exact' if torch.equal(got, refs[key]) else 'MISMATCH'))
        del got
    refs.clear()
    res = {c[0]: [] for c in cases}
    for _ in range(rounds):
        for name, fn, _, _ in cases:
            res[name].append(timeit(fn))
    for name, _, nbytes, _ in cases:
        t = np.array(res[name])
        print('%-22s min %8.2f us  med %8.2f us  %6.0f GB/s (alg, med)  frac %.3f' %
              (name, t.min(), np.median(t), nbytes / np.median(t) / 1e3,
               nbytes / np.median(t) / 1e3 / 8000.0))


if __name__ == '__main__':
    main()

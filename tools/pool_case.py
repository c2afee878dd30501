#!/usr/bin/env python
"""Run ONE pool kernel case in a loop (for rocprofv3 passes):
    python tools/pool_case.py SV rows_mp_bf16 [dbg] [iters]
    python tools/pool_case.py ALL        (the dominant kernels of S2 and SV, 10 launches each)
    python tools/pool_case.py ROOFLINE   (what bench.py's rocprofv3 --kernel-trace child runs:
                                          the S2 kernel round-robin into ROOFLINE_BUFFERS output
                                          allocations held at once, ROOFLINE_ROUNDS rounds -- launch
                                          i of the kernel writes buffer i % ROOFLINE_BUFFERS --, then
                                          the SV kernels, 30 launches each)
cases: rows_mp_bf16 (padded), rows_mp_f32, rows_cf_f32, rows_cf_bf16, slab_cf_f32."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402


ROOFLINE_BUFFERS, ROOFLINE_ROUNDS = 8, 25


def main():
    if sys.argv[1] == 'ROOFLINE':
        run('S2', 'slab_cf_f32', 0, ROOFLINE_ROUNDS, buffers=ROOFLINE_BUFFERS)
        for case in ('rows_cf_f32', 'rows_mp_f32', 'rows_mp_bf16'):
            run('SV', case, 0, 30)
        return
    if sys.argv[1] == 'ALL':   # what bench.py's PMC child passes run
        for tag, case in (('S2', 'slab_cf_f32'), ('SV', 'rows_cf_f32'), ('SV', 'rows_mp_f32'),
                          ('SV', 'rows_mp_bf16')):
            run(tag, case, 0, 10)
        return
    tag, case = sys.argv[1], sys.argv[2]
    dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    run(tag, case, dbg, iters)


def run(tag, case, dbg, iters, buffers=1):
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
    f = feat.bfloat16() if case.endswith('bf16') else feat
    out = torch.empty((1, C, Z, Y, X), dtype=torch.float32, device=dev)
    L.veon_pool_debug_set(dbg)
    if case.startswith('rows_mp'):
        vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)
        mp = torch.empty((1, C, Z // 2, Y // 2, X // 2), dtype=torch.float32, device=dev)
        for _ in range(iters):
            bp.rows_maxpool(depth, f, rd, rf, vs, shape, (2, 2, 2),
                            out_volume=vol if case.endswith('bf16') else None)
    elif case.startswith('rows_cf'):
        for _ in range(iters):
            bp.rows_forward(depth, f, rd, rf, vs, shape, out=out)
    else:
        plan = bp.build_plan(rb, st, 1, vpb, attach=False)
        s = _lib.stream_ptr(dev)
        outs = [out] + [torch.empty_like(out) for _ in range(buffers - 1)]
        for _ in range(iters):
            for o in outs:
                _lib.check(L.veon_bev_pool_v2_fwd_fused_ex(
                    C, st.numel(), 1, vpb, _lib.ptr(depth), _lib.ptr(f), bp._feat_code(f),
                    _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(ln),
                    _lib.ptr(plan), _lib.ptr(o), _lib.LAYOUT_BCZYX, s), 'slab_cf')
    torch.cuda.synchronize()
    L.veon_pool_debug_set(0)


if __name__ == '__main__':
    main()

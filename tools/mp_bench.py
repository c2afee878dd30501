#!/usr/bin/env python
"""The fused pool + 2x2x2 max-pool row kernel at the VEON shape (SV), bf16 rows into
the Conv3d body's padded input (what the VEON path runs) and fp32 -> fp32:

  dense     cached ranks of the whole frustum, softmax(randn) depth (859 k points)
  twohot60  two-hot lift by construction, metric depth ~ U(1, 60) m, eps 1e-6:
            a quarter of the pixels lie beyond the depth range and keep all their bins
  twohot45  the same with every pixel inside the depth range (U(1, 45))

each with the kernel's built-in chunk order (default) and the azimuth order,
interleaved rounds in one process, HIP events; outputs compared bit for bit.

    python tools/mp_bench.py [rounds] [--tune]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from tools.poolbench import timeit  # noqa: E402
from veon_amd import _lib, conv3d_ops, synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402

GRID, SIZE, CAMS, C = synthetic.GRID_VEON, (512, 1408), 6, 256


def twohot_case(dev, hi, seed=0):
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=GRID, input_size=SIZE,
                         out_channels=C, collapse_z=False, ds_feat=[2, 2, 2])).to(dev).eval()
    vt.sync_free = True
    hf, wf = SIZE[0] // 16, SIZE[1] // 16
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, CAMS, SIZE))]
    g = torch.Generator().manual_seed(seed)
    metric = (1.0 + (hi - 1.0) * torch.rand(1, CAMS, hf, wf, generator=g)).to(dev)
    feat = torch.randn(1, CAMS, C, hf, wf, generator=g).to(dev)
    with torch.no_grad():
        tw = vt.get_two_hot_windows(metric, eps=1e-6)
        vt([feat] + geom, tw)
    ws = list(vt.__dict__['_veon_lift_workspaces'].values())[-1]
    kept = int(ws.counts[0])
    return dict(depth=tw.wts, feat_nhwc=feat.permute(0, 1, 3, 4, 2).contiguous(),
                rd=ws.ranks_depth, rf=ws.ranks_feat, vs=ws.vstart, kept=kept,
                gsize=tuple(int(v) for v in vt.grid_size), keep=(vt, tw, ws))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    dev = torch.device('cuda:0')
    cases = {}
    cs = lift_case(GRID, SIZE, CAMS, C, str(dev))
    X, Y, Z = cs['gsize']
    vs = bp.build_voxel_table(cs['rb'], cs['st'], 1, Z * Y * X, attach=False)
    cases['dense'] = dict(depth=cs['depth'], feat_nhwc=cs['feat_nhwc'], rd=cs['rd'],
                          rf=cs['rf'], vs=vs, kept=cs['rb'].numel())
    cases['twohot60'] = twohot_case(dev, 60.0)
    cases['twohot45'] = twohot_case(dev, 45.0)
    shape = (1, Z, Y, X, C)
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)
    order = bp.cold_chunk_order(1, Z // 2, Y // 2, X // 2, dev)
    runs, res = [], {}
    for tag, cse in cases.items():
        fb = cse['feat_nhwc'].to(vol.rows.dtype)
        alg_b = (fb.numel() * 2 + cse['depth'].numel() * 4 + 8 * cse['kept'] +
                 4 * (Z * Y * X + 1) + (Z * Y * X // 8) * C * 2)
        for oname, o in (('builtin', None), ('azimuth', order)):
            def run_b(cse=cse, fb=fb, o=o):
                bp.rows_maxpool(cse['depth'], fb, cse['rd'], cse['rf'], cse['vs'], shape,
                                (2, 2, 2), out_volume=vol, chunk_order=o)

            def run_f(cse=cse, o=o):
                return bp.rows_maxpool(cse['depth'], cse['feat_nhwc'], cse['rd'], cse['rf'],
                                       cse['vs'], shape, (2, 2, 2), chunk_order=o)
            runs.append(('%s bf16 padded %s' % (tag, oname), run_b, alg_b, (tag, 'b')))
            runs.append(('%s f32 %s' % (tag, oname), run_f, None, (tag, 'f')))
    refs = {}
    for name, fn, _, key in runs:
        vol.rows.fill_(0)
        r = fn()
        torch.cuda.synchronize()
        got = (vol.rows if key[1] == 'b' else r).clone()
        if key in refs:
            print('%-32s %s' % (name, 'bit-exact' if torch.equal(got, refs[key]) else 'MISMATCH'))
        else:
            refs[key] = got
    refs.clear()
    res = {r[0]: [] for r in runs}
    for _ in range(rounds):
        for name, fn, _, _ in runs:
            res[name].append(timeit(fn, 20))
    for name, _, alg, key in runs:
        t = np.array(res[name])
        extra = ''
        if alg:
            extra = '  %6.0f GB/s alg (%.1f MB)  frac %.3f' % (
                alg / np.median(t) / 1e3, alg / 1e6, alg / np.median(t) / 1e3 / 8000)
        print('%-32s %7d pts  min %7.2f us  med %7.2f us%s'
              % (name, cases[key[0]]['kept'], t.min(), np.median(t), extra))
    if '--tune' in sys.argv:
        L = _lib.lib()
        for tag in ('dense', 'twohot60'):
            cse = cases[tag]
            fb = cse['feat_nhwc'].to(vol.rows.dtype)
            for workers in (256, 512, 1024, 2048):
                for cold in (32, 48, 64, 96):
                    L.veon_pool_tune_set(workers, cold, 256)
                    t = np.median([timeit(lambda: bp.rows_maxpool(
                        cse['depth'], fb, cse['rd'], cse['rf'], cse['vs'], shape, (2, 2, 2),
                        out_volume=vol), 20) for _ in range(3)])
                    print('%s workers %4d cold %3d: %7.2f us' % (tag, workers, cold, t), flush=True)
            L.veon_pool_tune_set(0, 0, 0)


if __name__ == '__main__':
    main()

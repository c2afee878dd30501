#!/usr/bin/env python
"""The per-call lift of one workload in a loop (for a rocprofv3 kernel trace):
view_transform(accelerate=False, sync_free) = geometry + counting-sort prepare + pool.
    python tools/percall_case.py [S2|SV] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'S2'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    grid, size, cams, C, ds = {'S2': (synthetic.GRID_S2, (256, 704), 6, 80, [1, 1, 1]),
                               'SV': (synthetic.GRID_VEON, (512, 1408), 6, 256, [2, 2, 2])}[tag]
    dev = torch.device('cuda:0')
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=grid, input_size=size,
                         downsample=16, out_channels=C, accelerate=False, collapse_z=False,
                         ds_feat=ds)).to(dev).eval()
    vt.sync_free = True
    hf, wf = size[0] // 16, size[1] // 16
    rig = synthetic.make_rig(1, cams, size)
    geom = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    depth5, feat5 = synthetic.make_depth_feat(1, cams, vt.D, C, hf, wf, seed=0, device=dev)
    depth = depth5.view(cams, vt.D, hf, wf)
    tran_feat = feat5.view(cams, C, hf, wf)
    inp = [feat5] + geom
    with torch.no_grad():
        for _ in range(iters):
            vt.view_transform(inp, depth, tran_feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(iters):
            vt.view_transform(inp, depth, tran_feat)
        e1.record()
    torch.cuda.synchronize()
    print('%s per-call lift, eager: %.1f us / call' % (tag, e0.elapsed_time(e1) / iters * 1e3))


if __name__ == '__main__':
    main()

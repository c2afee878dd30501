#!/usr/bin/env python
"""Time LSSViewTransformerRaw.forward (VEON shape, ds_feat=[2,2,2]) on the GPU
box: reference structure (pool -> amax) vs the fused pool+max-pool kernel, with
cached ranks, per-call prepare, and sync-free per-call prepare."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'SV'
    grid, size, C = {'SV': (synthetic.GRID_VEON, (512, 1408), 256),
                     'S2': (synthetic.GRID_S2, (256, 704), 80)}[tag]
    dev = 'cuda:0'
    rig = synthetic.make_rig(1, 6, size)
    inp = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    hf, wf = size[0] // 16, size[1] // 16
    res = {}
    for mode in ('accelerate', 'percall', 'sync_free'):
        vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=grid,
                             input_size=size, out_channels=C, collapse_z=False,
                             accelerate=(mode == 'accelerate'),
                             ds_feat=[2, 2, 2])).to(dev)
        vt.sync_free = mode == 'sync_free'
        depth, feat = synthetic.make_depth_feat(1, 6, vt.D, C, hf, wf, 0, dev)
        with torch.no_grad():
            for fused in (False, True):
                vt.fuse_ds = fused
                f = lambda: vt([feat] + inp, depth)
                res[(mode, fused)] = timeit(f)
                if mode == 'sync_free':
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        f()
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        f()
                    res[(mode + '+graph', fused)] = timeit(g.replay)
    for k, v in res.items():
        print('%-20s fused=%-5s %9.1f us  (%.0f samples/s)' % (k[0], k[1], v, 1e6 / v))


if __name__ == '__main__':
    main()

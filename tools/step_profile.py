#!/usr/bin/env python
"""cProfile of the host side of the S2 step (view_transform with cached ranks):
where do the ~40 us of Python per step go?  Not a test."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    C, size, cams = 80, (256, 704), 6
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=synthetic.GRID_S2,
                         input_size=size, downsample=16, out_channels=C, accelerate=True,
                         collapse_z=False, ds_feat=[1, 1, 1])).to(dev).eval()
    vt.persistent_output = True
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, cams, size))]
    hf, wf = size[0] // 16, size[1] // 16
    depth5, feat5 = synthetic.make_depth_feat(1, cams, vt.D, C, hf, wf, seed=0, device=dev)
    depth, tran = depth5.view(cams, vt.D, hf, wf), feat5.view(cams, C, hf, wf)
    inp = [feat5] + geom
    with torch.no_grad():
        for _ in range(20):
            vt.view_transform(inp, depth, tran)
        torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n):
            vt.view_transform(inp, depth, tran)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print('per step: issue %.1f us, wall %.1f us' % (t_issue / n * 1e6, t_all / n * 1e6))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            vt.view_transform(inp, depth, tran)
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(22)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Does any lift kernel consume uninitialised memory?  Run the sync-free lift
(SV shape) in a clean process, then fill the caching allocator's free blocks
with small in-range integer patterns, run it again and compare bit for bit.
Patterns are chosen to be valid indices everywhere so a stray read shows up as
a mismatch, not as a fault."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402


def poison(pattern):
    torch.cuda.synchronize()
    blocks = []
    size = 512
    while size <= (160 << 20):
        for mult in (1.0, 1.5):
            n = int(size * mult) // 4
            for _ in range(3):
                blocks.append(torch.full((n,), pattern, dtype=torch.int32, device='cuda:0'))
        size *= 2
    torch.cuda.synchronize()
    del blocks


def main():
    dev = 'cuda:0'
    size = (512, 1408)
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=synthetic.GRID_VEON,
                         input_size=size, out_channels=256, collapse_z=False,
                         ds_feat=[2, 2, 2])).to(dev).eval()
    vt.sync_free = True
    rig = synthetic.make_rig(1, 6, size)
    geom = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    depth, feat = synthetic.make_depth_feat(1, 6, vt.D, 256, size[0] // 16, size[1] // 16, 0)
    depth, feat = depth.to(dev), feat.to(dev)
    with torch.no_grad():
        for fuse in (True, False):
            vt.fuse_ds = fuse
            ref = vt([feat] + geom, depth).clone()
            for pattern in (1, 16256, 0):
                poison(pattern)
                out = vt([feat] + geom, depth)
                torch.cuda.synchronize()
                same = torch.equal(out, ref)
                print('fuse_ds=%s pattern %5d: %s' % (fuse, pattern, 'identical' if same else
                      'MISMATCH (%d elements)' % int((out != ref).sum())), flush=True)
                del out


if __name__ == '__main__':
    main()

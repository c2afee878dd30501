"""Inputs for the GPU micro-benchmarks, produced by the PRODUCT path (HIP
prepare of the view transformer) -- the oracle is test infrastructure and is
not used by tools."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402


def lift_case(grid, size, cams, C, dev='cuda:0', seed=0):
    """-> dict(depth (1,N,D,h,w), feat_nhwc (1,N,h,w,C), rb, rd, rf, st, ln
    (device int32), gsize (X,Y,Z), D)."""
    vt = build_neck(dict(type='LSSViewTransformerRaw', grid_config=grid, input_size=size,
                         out_channels=C, collapse_z=False, accelerate=True,
                         ds_feat=[1, 1, 1])).to(dev).eval()
    rig = synthetic.make_rig(1, cams, size)
    geom = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    hf, wf = size[0] // 16, size[1] // 16
    depth, feat = synthetic.make_depth_feat(1, cams, vt.D, C, hf, wf, seed)
    depth, feat = depth.to(dev), feat.to(dev)
    with torch.no_grad():
        vt.pre_compute([feat] + geom)
    return dict(depth=depth, feat_nhwc=feat.permute(0, 1, 3, 4, 2).contiguous(),
                rb=vt.ranks_bev, rd=vt.ranks_depth, rf=vt.ranks_feat,
                st=vt.interval_starts, ln=vt.interval_lengths,
                gsize=tuple(int(v) for v in vt.grid_size), D=vt.D)

"""Pin LSSViewTransformerBEVDepth with vectors from the
reference's own class (mmdet3d/models/necks/view_transformer.py:694-791), run
here on CPU, unmodified:

  * get_mlp_input  (:703-724)  -- the 27-d camera vector;
  * forward        (:780-791)  -- depth_net -> D / out_channels channel split ->
                                   softmax(dim=1) -> view_transform -> (bev_feat, depth);
  * get_downsampled_gt_depth / get_depth_loss (:726-778).

The reference's DepthNet cannot be built here (mmdet's BasicBlock and mmcv's DCN
are absent), and its conv stack is out of scope anyway: the class is
instantiated with a deterministic stand-in `depth_net` (a 1x1 conv on the image
features plus a linear map of mlp_input), whose weights travel in the fixture, so
the test pins the reference's WIRING around the depth net.  `bev_pool_v2` is this
repo's CPU restatement (the reference op is CUDA-only), as in gen_golden.py.

    python oracle/tools/gen_golden_bevdepth.py     (needs /root/reference)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.tools import ref_import  # noqa: E402
from oracle.tools.gen_golden import GOLD, cpu_bev_pool_v2, np_, perturbed_rig  # noqa: E402
from veon_amd import synthetic  # noqa: E402


class StandInDepthNet(nn.Module):
    """depth_net(x, mlp_input, stereo_metas) -> (B*N, D + C_out, H, W); same call
    signature as the reference's DepthNet.forward (:603-630)."""

    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, **kw):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, depth_channels + context_channels, 1)
        self.mlp = nn.Linear(27, depth_channels + context_channels)
        # mlp_input carries focal lengths (~1e3): keep the logits O(1) so that the
        # softmax of the fixture is not saturated
        nn.init.normal_(self.mlp.weight, std=3e-4)

    def forward(self, x, mlp_input, stereo_metas=None):
        y = self.conv(x)
        return y + self.mlp(mlp_input.reshape(-1, mlp_input.shape[-1]))[:, :, None, None]


def main():
    raw, vtmod = ref_import.load_view_transformers(cpu_bev_pool_v2)
    vtmod.DepthNet = StandInDepthNet      # looked up by name in __init__ (:699)
    torch.manual_seed(7)
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    size, B, N, Cin, Cout = (64, 176), 2, 3, 16, 8
    for sid in (False, True):
        vt = vtmod.LSSViewTransformerBEVDepth(
            grid_config=grid, input_size=size, downsample=16, in_channels=Cin,
            out_channels=Cout, accelerate=False, sid=sid, collapse_z=False)
        vt.eval()
        rig = perturbed_rig(B, N, size, seed=11)
        inp = synthetic.rig_inputs(rig)
        hf, wf = size[0] // 16, size[1] // 16
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, N, Cin, hf, wf, generator=g)
        with torch.no_grad():
            mlp_input = vt.get_mlp_input(*inp)
            bev_feat, depth = vt.forward([x] + list(inp) + [mlp_input])
            gt = 14.0 * torch.rand(B, N, size[0], size[1], generator=g)
            gt[gt < 2.0] = 0.0
            onehot = vt.get_downsampled_gt_depth(gt)
            loss = vt.get_depth_loss(gt, depth)
        sd = {k: np_(v) for k, v in vt.depth_net.state_dict().items()}
        np.savez_compressed(
            os.path.join(GOLD, 'bevdepth_tiny%s.npz' % ('_sid' if sid else '')),
            grid_x=np.array(grid['x']), grid_y=np.array(grid['y']),
            grid_z=np.array(grid['z']), grid_depth=np.array(grid['depth']),
            input_size=np.array(size), sid=np.array(sid), D=np.array(vt.D),
            in_channels=np.array(Cin), out_channels=np.array(Cout),
            sensor2ego=np_(rig['sensor2ego']), ego2global=np_(rig['ego2global']),
            intrins=np_(rig['intrins']), post_rots=np_(rig['post_rots']),
            post_trans=np_(rig['post_trans']), bda=np_(rig['bda']),
            x=np_(x), mlp_input=np_(mlp_input), bev_feat=np_(bev_feat), depth=np_(depth),
            gt_depth=np_(gt), gt_onehot=np_(onehot), depth_loss=np_(loss),
            **{'depth_net.' + k: v for k, v in sd.items()})
        print('sid=%s D=%d mlp_input %s bev_feat %s depth %s loss %.6f' % (
            sid, vt.D, tuple(mlp_input.shape), tuple(bev_feat.shape), tuple(depth.shape),
            float(loss)))


if __name__ == '__main__':
    main()

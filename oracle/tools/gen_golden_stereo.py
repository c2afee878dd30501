"""Golden vectors for BEVStereo's cost volume (view_transformer.py:543-601).

TEST INFRASTRUCTURE (fixture generation, build container only).  The reference's
``DepthNet`` cannot be constructed here (mmdet's BasicBlock is absent), but
``gen_grid`` and ``calculate_cost_volumn`` only use ``self.bias``: they are run
from the reference's unmodified view_transformer.py on an instance created with
``object.__new__`` -- no reference code is copied.

    python oracle/tools/gen_golden_stereo.py  ->  tests/golden/stereo_cost_volume.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    _, vt = ref_import.load_view_transformers(lambda *a, **k: None)
    net = object.__new__(vt.DepthNet)
    torch.nn.Module.__init__(net)
    net.bias = 5.0
    g = torch.Generator().manual_seed(0)
    B, N, D, H, W, C = 1, 2, 7, 6, 9, 8
    hi, wi = H * 4, W * 4
    xs = torch.linspace(0, wi - 1, W).view(1, 1, W).expand(D, H, W)
    ys = torch.linspace(0, hi - 1, H).view(1, H, 1).expand(D, H, W)
    ds = (torch.arange(D).float() * 2 + 2).view(D, 1, 1).expand(D, H, W)
    frustum = torch.stack([xs, ys, ds], -1)
    ang = torch.tensor([0.02, -0.03])
    k2s = torch.eye(4).repeat(B, N, 1, 1)
    k2s[0, :, 0, 0] = torch.cos(ang)
    k2s[0, :, 0, 2] = torch.sin(ang)
    k2s[0, :, 2, 0] = -torch.sin(ang)
    k2s[0, :, 2, 2] = torch.cos(ang)
    k2s[0, :, :3, 3] = torch.tensor([[0.3, 0.0, -0.8], [-0.2, 0.05, 4.0]])
    intr = torch.tensor([[30.0, 0, wi / 2], [0, 30.0, hi / 2], [0, 0, 1]]).repeat(B, N, 1, 1)
    post_rots = (torch.eye(3) * 0.9).repeat(B, N, 1, 1)
    post_rots[..., 2, 2] = 1
    post_trans = torch.tensor([[1.5, -2.0, 0.0], [0.0, 1.0, 0.0]]).view(B, N, 3)
    metas = dict(frustum=frustum, post_trans=post_trans, post_rots=post_rots,
                 k2s_sensor=k2s, intrins=intr,
                 cv_feat_list=[torch.randn(B * N, C, H, W, generator=g),
                               torch.randn(B * N, C, H, W, generator=g)])
    with torch.no_grad():
        grid = net.gen_grid(metas, B, N, D, H, W, hi, wi)
        cv = net.calculate_cost_volumn(metas)
    out = {k: v.numpy() for k, v in metas.items() if torch.is_tensor(v)}
    out['prev'], out['curr'] = (t.numpy() for t in metas['cv_feat_list'])
    out['grid'], out['cost_volume'], out['bias'] = grid.numpy(), cv.numpy(), np.float32(5.0)
    path = os.path.join(ROOT, 'tests', 'golden', 'stereo_cost_volume.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, grid.shape, cv.shape, float(cv.sum()))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Temporal fusion at VEON's shape (1, 256, 8, 100, 100): MFMA path on
PaddedVolumes vs the PyTorch module (fp32 and bf16 autocast), plus the
align_after_lss warp.  Not a test."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.kbench import timeit  # noqa: E402
from veon_amd import conv3d_ops  # noqa: E402
from veon_amd.models.semantic_net import temporal_fusion as tfm  # noqa: E402


def main():
    dev = 'cuda:0'
    C, shape = 256, (8, 100, 100)
    grid = {'x': [-40.0, 40.0, 0.4], 'y': [-40.0, 40.0, 0.4], 'z': [-1.0, 5.4, 0.4]}
    eye = torch.eye(4, device=dev)[None, None]
    move = eye.clone()
    move[0, 0, :3, 3] = torch.tensor([1.7, 0.3, 0.02])
    for T in (1, 2):
        net = tfm.TemporalFusionMultiFrame(C, seqs=T).eval().to(dev)
        cur = torch.randn(1, C, *shape, device=dev)
        prevs = [torch.randn(1, C, *shape, device=dev) for _ in range(T)]
        vc, vp = conv3d_ops.pack(cur), [conv3d_ops.pack(p) for p in prevs]
        with torch.no_grad():
            t_hip = timeit(lambda: net(vc, vp), 10)
            t_f32 = t_bf = float('nan')
            if not os.environ.get('ONLY_HIP'):   # keep a rocprof trace to our kernels
                t_f32 = timeit(lambda: net(cur, prevs), 3)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    t_bf = timeit(lambda: net(cur, prevs), 3)
            d = net.deform_fusion_layer.t_deform
            kv = d.project_kv(vc)
            q = conv3d_ops.pack(torch.randn(1, C, *shape, device=dev))
            off = conv3d_ops.pack(torch.randn(1, 96, *shape, device=dev) * 1.5)
            t_da = timeit(lambda: conv3d_ops.deform_attention(kv, q, off, 4), 10)
        print('T=%d past frames: MFMA path %.2f ms | torch fp32 %.1f ms | torch bf16 autocast '
              '%.1f ms | deform gather kernel alone %.3f ms' % (T, t_hip / 1e3, t_f32 / 1e3,
                                                                 t_bf / 1e3, t_da / 1e3))
    with torch.no_grad():
        t_w = timeit(lambda: tfm.align_after_lss(vc, [eye, move], grid, (2, 2, 2)), 10)
        t_wt = float('nan') if os.environ.get('ONLY_HIP') else timeit(
            lambda: tfm.align_after_lss(cur, [eye, move], grid, (2, 2, 2)), 5)
    print('align_after_lss warp: HIP %.3f ms (incl. host 4x4 algebra) | torch grid_sample fp32 '
          '%.2f ms' % (t_w / 1e3, t_wt / 1e3))


if __name__ == '__main__':
    main()

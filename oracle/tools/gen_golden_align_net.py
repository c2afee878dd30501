"""Golden vectors for the whole occupancy decoder ``AlignNetOcc3D`` and its 2-D
fusion layers (SURVEY 8 rows a5/a11/a12 + f1/f2, single frame).

TEST INFRASTRUCTURE (fixture generation, build container only).  The reference's
own files -- ``semantic_net/layers.py`` (LayerNorm, CatFusionLift),
``side_adapter/align_net_occ3d.py`` (AlignNetOcc3D, ResBlock3D, PredHead3D*) and
``necks/view_transformer_raw.py`` (LSSViewTransformerRaw) -- are loaded
unmodified and run end to end on CPU.  Stand-ins, as elsewhere: name-only stubs
for detectron2 / fvcore / mmdet imports (``detectron2.layers.Conv2d`` is used
without norm / activation there, i.e. as ``nn.Conv2d``), the ``ConvModule``
stand-in of gen_golden_body.py (mmcv absent), and the torch ``index_add_`` port
for the CUDA-only ``bev_pool_v2``.

    python oracle/tools/gen_golden_align_net.py -> tests/golden/align_net_tiny.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from gen_golden_body import ConvModuleStandIn, randomise  # noqa: E402
from oracle import lss_torch  # noqa: E402
from veon_amd import synthetic  # noqa: E402


def cpu_bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    bev_feat_shape, interval_starts, interval_lengths):
    return lss_torch.pool(depth.float(), feat.float(), ranks_depth, ranks_feat,
                          ranks_bev, bev_feat_shape)


GRID = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
        'depth': [1.0, 13.0, 1.0]}
SIZE = (64, 176)


def main():
    raw, _ = ref_import.load_view_transformers(cpu_bev_pool_v2)
    ref_import._mod('mmcv.cnn.bricks')
    ref_import._mod('mmcv.cnn.bricks.conv_module', ConvModule=ConvModuleStandIn)
    ref_import._mod('mmdet3d.models.necks.view_transformer_raw',
                    LSSViewTransformerRaw=raw.LSSViewTransformerRaw)
    ref_import._mod('mmdet3d.utils')
    ref_import._mod('mmdet3d.utils.vis', vis_occ=None)
    ref_import._mod('fvcore')
    ref_import._mod('fvcore.nn')
    ref_import._mod('fvcore.nn.weight_init', c2_xavier_fill=lambda m: None)
    sys.modules['fvcore.nn'].weight_init = sys.modules['fvcore.nn.weight_init']
    ref_import._mod('detectron2')
    ref_import._mod('detectron2.layers', CNNBlockBase=torch.nn.Module,
                    Conv2d=torch.nn.Conv2d)
    pkg = 'refsem4'
    ref_import._mod(pkg)
    layers = ref_import.load('mmdet3d/models/semantic_net/layers.py', pkg + '.layers')
    ref_import._mod(pkg + '.side_adapter')
    ao = ref_import.load('mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py',
                         pkg + '.side_adapter.align_net_occ3d')
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    N = 2
    net = ao.AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=64, clip_outdim=24,
                           layer_lifting_map=['2->0->0'], fusion_type='cat_fusion',
                           layer_depth=2, num_temporal=1).eval()
    randomise(net, gen)
    for m in net.modules():
        if isinstance(m, layers.LayerNorm):
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    vt = raw.LSSViewTransformerRaw(grid_config=GRID, input_size=SIZE, downsample=16,
                                   out_channels=64, collapse_z=False, ds_feat=[2, 2, 2])
    net.lss_view_transformer = vt
    net.num_frame, net.num_camera = 1, N
    rig = synthetic.make_rig(1, N, SIZE)
    s2e, e2g, intr, pr, pt, bda = synthetic.rig_inputs(rig)
    hf, wf = SIZE[0] // 16, SIZE[1] // 16
    metric = torch.rand(1, N, hf * 8, wf * 8, generator=gen) * 14
    metric[metric < 1.0] = 0.0                      # holes, as a real depth map has
    clip2 = torch.randn(N, 32, 3, 7, generator=gen)
    clip1 = torch.randn(N, 32, 3, 7, generator=gen)
    supp = torch.randn(N, 16, 6, 14, generator=gen)
    sem_feat = torch.zeros(N, 8, hf, wf)
    metas = [s2e, e2g, intr, pr, pt, bda[None]]
    with torch.no_grad():
        cat_out = net.fusion_layers['layer_0'](supp, clip2, (hf, wf))
        out = net(sem_feat, {1: clip1, 2: clip2}, [supp], metric, metas)
        early = net.forward_early(sem_feat, {1: clip1, 2: clip2}, [supp], metric, metas)
    sd = {k: v for k, v in net.state_dict().items() if 'lss_view_transformer' not in k}
    res = {'metric': metric, 'clip1': clip1, 'clip2': clip2, 'supp': supp,
           's2e': s2e, 'e2g': e2g, 'intr': intr, 'pr': pr, 'pt': pt, 'bda': bda,
           'cat_fusion_out': cat_out, 'lifted': early, 'bin_occ': out['bin_occ'],
           'feat_occ': out['feat_occ']}
    res.update({'sd/' + k: v for k, v in sd.items()})
    path = os.path.join(ROOT, 'tests', 'golden', 'align_net_tiny.npz')
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in res.items()})
    print('wrote', path, {k: tuple(v.shape) for k, v in out.items()}, tuple(early.shape),
          float(early.abs().sum()))

    # ---- temporal decoder (num_temporal=2): forward_early of a past frame, then
    # forward(..., occ_feat_prevs) -- align_net_occ3d.py:252-262, 268-280 ----------
    net_t = ao.AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=32, clip_outdim=24,
                             layer_lifting_map=['2->0->0'], fusion_type='cat_fusion',
                             layer_depth=1, num_temporal=2).eval()
    randomise(net_t, gen)
    with torch.no_grad():
        net_t.temporal_fusion.deform_fusion_layer.t_deform.offset_conv[2].weight.mul_(6.0)
    vt_t = raw.LSSViewTransformerRaw(grid_config=GRID, input_size=SIZE, downsample=16,
                                     out_channels=32, collapse_z=False, ds_feat=[2, 2, 2])
    net_t.lss_view_transformer = vt_t
    net_t.num_frame, net_t.num_camera = 1, N
    metric_prev = torch.rand(1, N, hf * 8, wf * 8, generator=gen) * 14
    clip2_prev = torch.randn(N, 32, 3, 7, generator=gen)
    supp_prev = torch.randn(N, 16, 6, 14, generator=gen)
    with torch.no_grad():
        early_prev = net_t.forward_early(sem_feat, {1: clip1, 2: clip2_prev}, [supp_prev],
                                         metric_prev, metas)
        out_t = net_t(sem_feat, {1: clip1, 2: clip2}, [supp], metric, metas, [early_prev])
    res_t = {'metric_prev': metric_prev, 'clip2_prev': clip2_prev, 'supp_prev': supp_prev,
             'early_prev': early_prev, 'bin_occ': out_t['bin_occ'],
             'feat_occ': out_t['feat_occ']}
    res_t.update({'sd/' + k: v for k, v in net_t.state_dict().items()
                  if 'lss_view_transformer' not in k})
    path = os.path.join(ROOT, 'tests', 'golden', 'align_net_temporal_tiny.npz')
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in res_t.items()})
    print('wrote', path, {k: tuple(v.shape) for k, v in out_t.items()},
          float(early_prev.abs().sum()))


if __name__ == '__main__':
    main()

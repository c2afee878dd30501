"""Path-level golden vector: the 3-D occupancy path of SANInVeonTemporal.forward
(san_in_veon_temporal.py:118-123, 189-211, 257-259) composed from the REFERENCE'S
OWN modules on CPU, unmodified:

    FeatureExtractor            clip_utils/visual.py:23-91
    HighresSideAdaptorNetwork   side_adapter/highres_side_adaptor.py:109-300
    RecWithAttnbiasHead.update_remaining_clip_feats   clip_utils/visual.py:258-285
    AlignNetOcc3D (+ CatFusionLift, ResBlock3D, PredHead3D*)   side_adapter/align_net_occ3d.py
    LSSViewTransformerRaw       necks/view_transformer_raw.py
    trilinear upsampling + semantic_inference_3d   san_in_veon_temporal.py:196-211, 257-259

SANInVeonTemporal itself cannot be imported (open_clip, detectron2, timm absent),
so the chain is wired here line by line as its forward does.  Stand-ins, as in the
per-module generators: this repo's CLIP residual block for open_clip's (third
party, absent: parity of that block is unpinned), the ConvModule stand-in for
mmcv's, name-only stubs for detectron2 / fvcore, the torch index_add_ port for the
CUDA-only bev_pool_v2.  The depth branch is an INPUT (metric depth), the 2-D mask
branch (timm side adapter) only supplies a shape to the decoder.

    python oracle/tools/gen_golden_path.py -> tests/golden/path_tiny.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from gen_golden_body import ConvModuleStandIn, randomise  # noqa: E402
from gen_golden_hsa import _Registry  # noqa: E402
from oracle import lss_torch  # noqa: E402
from veon_amd import synthetic  # noqa: E402
from veon_amd.models.semantic_net import ClipVisualTrunk  # noqa: E402

GRID = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
        'depth': [1.0, 13.0, 1.0]}
SIZE, NCAM = (64, 176), 2
CFG = dict(clip_width=64, clip_layers=4, clip_heads=1, clip_first_tail=2, clip_proj_dim=24,
           embed_dim=64, hsa_dim=64, n_classes=5, occ_size=(4, 20, 20),
           hsa_fusion_map=('0->1->1', '1->2->2'))


def cpu_bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                    interval_starts, interval_lengths):
    return lss_torch.pool(depth.float(), feat.float(), ranks_depth, ranks_feat, ranks_bev,
                          bev_feat_shape)


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
    ref_import._mod('open_clip')
    ref_import._mod('open_clip.transformer', ResidualAttentionBlock=torch.nn.Module,
                    VisionTransformer=torch.nn.Module)
    ref_import._mod('detectron2')
    ref_import._mod('detectron2.config', configurable=lambda f: f)
    ref_import._mod('detectron2.utils')
    ref_import._mod('detectron2.utils.registry', Registry=_Registry)
    ref_import._mod('detectron2.layers', CNNBlockBase=torch.nn.Module, Conv2d=torch.nn.Conv2d,
                    ShapeSpec=object)
    pkg = 'refpath'
    ref_import._mod(pkg)
    ref_import.load('mmdet3d/models/semantic_net/attn_helper.py', pkg + '.attn_helper')
    ref_import._mod(pkg + '.clip_utils')
    vis = ref_import.load('mmdet3d/models/semantic_net/clip_utils/visual.py',
                          pkg + '.clip_utils.visual')
    layers = ref_import.load('mmdet3d/models/semantic_net/layers.py', pkg + '.layers')
    ref_import._mod(pkg + '.side_adapter')
    hsa = ref_import.load('mmdet3d/models/semantic_net/side_adapter/highres_side_adaptor.py',
                          pkg + '.side_adapter.highres_side_adaptor')
    ao = ref_import.load('mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py',
                         pkg + '.side_adapter.align_net_occ3d')

    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    W, Lr, K = CFG['clip_width'], CFG['clip_layers'], CFG['clip_first_tail']
    # ---- CLIP visual encoder (duck-typed open_clip VisionTransformer around this
    #      repo's blocks) -> the reference's FeatureExtractor and recognition head
    trunk = ClipVisualTrunk(image_size=64, patch_size=16, width=W, layers=Lr,
                            heads=CFG['clip_heads']).eval()
    ln_post = torch.nn.LayerNorm(W).eval()
    proj = torch.nn.Parameter(torch.randn(W, CFG['clip_proj_dim'], generator=gen) * W ** -0.5)
    enc = types.SimpleNamespace(
        output_tokens=False, image_size=(64, 64), patch_size=(16, 16),
        grid_size=trunk.grid_size, ln_pre=trunk.ln_pre, input_patchnorm=False,
        patchnorm_pre_ln=torch.nn.Identity(), conv1=trunk.conv1,
        class_embedding=trunk.class_embedding, positional_embedding=trunk.positional_embedding,
        patch_dropout=torch.nn.Identity(), output_dim=CFG['clip_proj_dim'],
        transformer=types.SimpleNamespace(resblocks=trunk.resblocks),
        global_average_pool=False, attn_pool=None, ln_post=ln_post, proj=proj)
    fe = vis.FeatureExtractor(enc, last_layer_idx=K, frozen_exclude=['all']).eval()
    head = vis.RecWithAttnbiasHead(enc, first_layer_idx=K, frozen_exclude=['all'],
                                   sos_token_format='cls_token', sos_token_num=3,
                                   cross_attn=True, downsample_method='bilinear').eval()
    # ---- HSA network
    cr_map = {int(i): (int(j), int(k)) for i, j, k in
              [x.split('->') for x in CFG['hsa_fusion_map']]}
    D = CFG['hsa_dim']
    pe = hsa.PatchEmbed(SIZE, (8, 8), embed_dim=D, norm_layer=False)
    body = torch.nn.ModuleList([
        hsa.HighresSideAdaptorBlock(dim=D, neck_dim=W, mlp_dim=D, pre_norm=(i == 0),
                                    use_add=cr_map[i][1] >= 0, use_checkpoint=False)
        for i in range(len(cr_map))])
    rear = hsa.AttnManipulateBlock(dim=D, mlp_dim=D, clip_dim=W, heads=CFG['clip_heads'],
                                   dim_head=32, attn_layers=max(Lr - K, 1), add_layers=2,
                                   supp_dim=D, pre_norm=False, use_checkpoint=False)
    hsa_net = hsa.HighresSideAdaptorNetwork(pe, body, rear, cr_map, use_checkpoint=False).eval()
    # ---- decoder + lift
    dec = ao.AlignNetOcc3D(clip_dim=W, hsa_dim=D, embed_dim=CFG['embed_dim'],
                           clip_outdim=CFG['clip_proj_dim'],
                           layer_lifting_map=['%d->0->0' % Lr], fusion_type='cat_fusion',
                           layer_depth=4, num_temporal=1).eval()
    randomise(dec, gen)
    with torch.no_grad():
        for net in (hsa_net, dec):
            for m in net.modules():
                if isinstance(m, (torch.nn.LayerNorm, layers.LayerNorm)):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
        ln_post.weight.uniform_(0.5, 1.5)
        ln_post.bias.normal_(0, 0.1)
    vt = raw.LSSViewTransformerRaw(grid_config=GRID, input_size=SIZE, downsample=16,
                                   out_channels=CFG['embed_dim'], collapse_z=False,
                                   ds_feat=[2, 2, 2])
    dec.lss_view_transformer = vt
    dec.num_frame, dec.num_camera = 1, NCAM
    ov = torch.randn(CFG['n_classes'], CFG['clip_proj_dim'], generator=gen)

    rig = synthetic.make_rig(1, NCAM, SIZE)
    s2e, e2g, intr, pr, pt, bda = synthetic.rig_inputs(rig)
    images = torch.randn(1, NCAM, 3, *SIZE, generator=gen)
    hf, wf = SIZE[0] // 16, SIZE[1] // 16
    metric = torch.rand(1, NCAM, SIZE[0] // 2, SIZE[1] // 2, generator=gen) * 14
    metric[metric < 1.0] = 0.0
    with torch.no_grad():
        # san_in_veon_temporal.py:116-124
        img = images.view(NCAM, 3, *SIZE)
        clip_input = F.interpolate(img, scale_factor=0.5, mode='bilinear')
        clip_feats = fe(clip_input)
        # :189-190
        offsets, attns, supp = hsa_net(img, clip_feats)
        clip_feats = head.update_remaining_clip_feats(clip_feats, offsets, attns)
        # :193-195 (sem_embed_ds only supplies a shape to the decoder)
        sem_embed_ds = torch.zeros(NCAM, 1, hf, wf)
        occ = dec(sem_embed_ds, clip_feats, [supp], metric, [s2e, e2g, intr, pr, pt, bda[None]],
                  [])
        # :196-211, 257-259
        feat_occ = F.interpolate(occ['feat_occ'], size=CFG['occ_size'], mode='trilinear',
                                 align_corners=False)
        bin_occ = F.interpolate(occ['bin_occ'], size=CFG['occ_size'], mode='trilinear',
                                align_corners=False)
        sem_occ = torch.einsum('qc,bczhw->bqzhw', ov, feat_occ)
    res = {'images': images, 'metric': metric, 'ov_classifier_weight': ov,
           's2e': s2e, 'e2g': e2g, 'intr': intr, 'pr': pr, 'pt': pt, 'bda': bda,
           'sem_occ': sem_occ, 'bin_occ': bin_occ, 'supp': supp,
           'clip_feat_proj': clip_feats['clip_feat_proj'], 'clip_last': clip_feats[Lr],
           'feat_occ_lowres': occ['feat_occ'], 'bin_occ_lowres': occ['bin_occ']}
    res.update({'trunk/' + k: v for k, v in trunk.state_dict().items()})
    res.update({'ln_post/' + k: v for k, v in ln_post.state_dict().items()})
    res['clip_proj'] = proj.detach()
    res.update({'hsa/' + k: v for k, v in hsa_net.state_dict().items()})
    res.update({'dec/' + k: v for k, v in dec.state_dict().items()
                if 'lss_view_transformer' not in k})
    path = os.path.join(ROOT, 'tests', 'golden', 'path_tiny.npz')
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in res.items()})
    print('wrote', path, tuple(sem_occ.shape), tuple(bin_occ.shape),
          'sem_occ rms %.4f' % float(sem_occ.pow(2).mean().sqrt()),
          os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()

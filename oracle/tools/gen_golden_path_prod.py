"""Path-level golden vector at PRODUCTION widths with the depth encoder in the loop
(VERDICT r2 item 6): the chain of gen_golden_path.py -- the reference's own
FeatureExtractor -> HighresSideAdaptorNetwork -> RecWithAttnbiasHead.
update_remaining_clip_feats -> AlignNetOcc3D + LSSViewTransformerRaw -> trilinear
upsampling -> classifier einsum (san_in_veon_temporal.py:118-123, 189-211, 257-259) --
with CLIP ViT-B/16 dimensions (width 768, 12 heads, 12 layers, K = 9, projection 512),
HSA width 384, embed_dim 256 (configs/san_config.py), and the metric depth produced
by the reference's own DepthAnythingV2Adaptor(vitb) and ``estimate_depth``
(veon_temporal.py:209-214, 244-253), two cameras at 64x176 on a small grid.

Weights are NOT stored (CLIP-B + DA-V2-B + HSA + decoder are ~0.8 GB in fp32): both
sides initialise every state-dict entry from its NAME (tests/helpers.named_init_), so
the fixture holds inputs and expected outputs only.  The CLIP residual block is this
repo's restatement on both sides (open_clip absent: unpinned), as in gen_golden_path.py.

    python oracle/tools/gen_golden_path_prod.py -> tests/golden/path_prod.npz
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
import gen_golden_vit  # noqa: E402
import ref_import  # noqa: E402
from gen_golden_body import ConvModuleStandIn  # noqa: E402
from gen_golden_hsa import _Registry  # noqa: E402
from gen_golden_path import cpu_bev_pool_v2  # noqa: E402
from tests.helpers import named_init_  # noqa: E402
from veon_amd import synthetic  # noqa: E402
from veon_amd.models.semantic_net import ClipVisualTrunk  # noqa: E402

GRID = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
        'depth': [1.0, 13.0, 1.0]}
SIZE, NCAM = (64, 176), 2
CFG = dict(clip_width=768, clip_layers=12, clip_heads=12, clip_first_tail=9,
           clip_proj_dim=512, embed_dim=256, hsa_dim=384, n_classes=17,
           occ_size=(4, 20, 20), hsa_fusion_map=('0->3->3', '1->6->6', '2->9->9'))
DAV2 = dict(encoder='vitb', features=128, out_channels=[96, 192, 384, 768], max_depth=80.0,
            use_lora=True, lora_r=16)
# the reference's depth is sigmoid * 80 m; the small test grid bins 1..13 m, so the
# fixture scales it into the grid's range before the lift (a fixed, stated factor), and
# the DPT head's last 1x1 conv is drawn 40x larger so that the sigmoid leaves its
# middle (depths of 20-60 m instead of 40 +- 1 m)
DEPTH_SCALE = 0.2
DEPTH_BOOST = {'depth_head.scratch.output_conv2.2.weight': 40.0}


def main():
    # ---- the reference's depth model first (its own package stubs), then the chain's
    dpt, _ = gen_golden_vit.load_reference()
    depth_model = dpt.DepthAnythingV2Adaptor(**DAV2)
    named_init_(depth_model, 'depth/', boost=DEPTH_BOOST)   # (train mode: LoRA unmerged)
    depth_model.eval()                                      # merges W += B A * alpha / r
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
    pkg = 'refpathprod'
    ref_import._mod(pkg)
    ref_import.load('mmdet3d/models/semantic_net/attn_helper.py', pkg + '.attn_helper')
    ref_import._mod(pkg + '.clip_utils')
    vis = ref_import.load('mmdet3d/models/semantic_net/clip_utils/visual.py',
                          pkg + '.clip_utils.visual')
    ref_import.load('mmdet3d/models/semantic_net/layers.py', pkg + '.layers')
    ref_import._mod(pkg + '.side_adapter')
    hsa = ref_import.load('mmdet3d/models/semantic_net/side_adapter/highres_side_adaptor.py',
                          pkg + '.side_adapter.highres_side_adaptor')
    ao = ref_import.load('mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py',
                         pkg + '.side_adapter.align_net_occ3d')

    W, Lr, K = CFG['clip_width'], CFG['clip_layers'], CFG['clip_first_tail']
    trunk = named_init_(ClipVisualTrunk(image_size=224, patch_size=16, width=W, layers=Lr,
                                        heads=CFG['clip_heads']).eval(), 'trunk/')
    ln_post = named_init_(torch.nn.LayerNorm(W).eval(), 'ln_post/')
    holder = torch.nn.ParameterDict({
        'clip_proj': torch.nn.Parameter(torch.zeros(W, CFG['clip_proj_dim'])),
        'ov_classifier_weight': torch.nn.Parameter(torch.zeros(CFG['n_classes'],
                                                               CFG['clip_proj_dim']))})
    named_init_(holder, 'top/')
    proj, ov = holder['clip_proj'], holder['ov_classifier_weight'].detach()
    enc = types.SimpleNamespace(
        output_tokens=False, image_size=(224, 224), patch_size=(16, 16),
        grid_size=trunk.grid_size, ln_pre=trunk.ln_pre, input_patchnorm=False,
        patchnorm_pre_ln=torch.nn.Identity(), conv1=trunk.conv1,
        class_embedding=trunk.class_embedding, positional_embedding=trunk.positional_embedding,
        patch_dropout=torch.nn.Identity(), output_dim=CFG['clip_proj_dim'],
        transformer=types.SimpleNamespace(resblocks=trunk.resblocks),
        global_average_pool=False, attn_pool=None, ln_post=ln_post, proj=proj)
    fe = vis.FeatureExtractor(enc, last_layer_idx=K, frozen_exclude=['all']).eval()
    head = vis.RecWithAttnbiasHead(enc, first_layer_idx=K, frozen_exclude=['all'],
                                   sos_token_format='cls_token', sos_token_num=100,
                                   cross_attn=True, downsample_method='bilinear').eval()
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
    hsa_net = named_init_(hsa.HighresSideAdaptorNetwork(pe, body, rear, cr_map,
                                                        use_checkpoint=False).eval(), 'hsa/')
    dec = ao.AlignNetOcc3D(clip_dim=W, hsa_dim=D, embed_dim=CFG['embed_dim'],
                           clip_outdim=CFG['clip_proj_dim'],
                           layer_lifting_map=['%d->0->0' % Lr], fusion_type='cat_fusion',
                           layer_depth=4, num_temporal=1).eval()
    named_init_(dec, 'dec/')
    vt = raw.LSSViewTransformerRaw(grid_config=GRID, input_size=SIZE, downsample=16,
                                   out_channels=CFG['embed_dim'], collapse_z=False,
                                   ds_feat=[2, 2, 2])
    dec.lss_view_transformer = vt
    dec.num_frame, dec.num_camera = 1, NCAM

    rig = synthetic.make_rig(1, NCAM, SIZE)
    s2e, e2g, intr, pr, pt, bda = synthetic.rig_inputs(rig)
    gen = torch.Generator().manual_seed(7)
    images = torch.randn(1, NCAM, 3, *SIZE, generator=gen)
    hf, wf = SIZE[0] // 16, SIZE[1] // 16
    with torch.no_grad():
        img = images.view(NCAM, 3, *SIZE)
        # veon_temporal.py:209-214, 244-253 (depth_img_inputs: the 252 x 700 image the data
        # pipeline prepares; here a bilinear resize of the same image, as the build does)
        din = F.interpolate(img, (252, 700), mode='bilinear', align_corners=False)
        abs_depth = depth_model(din)['metric_depth']
        abs_depth = F.interpolate(abs_depth[:, None], (SIZE[0] // 2, SIZE[1] // 2),
                                  mode='bilinear', align_corners=True)
        metric = abs_depth.view(1, NCAM, SIZE[0] // 2, SIZE[1] // 2)
        lift_depth = metric * DEPTH_SCALE
        # san_in_veon_temporal.py:116-124, 189-190
        clip_input = F.interpolate(img, scale_factor=0.5, mode='bilinear')
        clip_feats = fe(clip_input)
        offsets, attns, supp = hsa_net(img, clip_feats)
        clip_feats = head.update_remaining_clip_feats(clip_feats, offsets, attns)
        sem_embed_ds = torch.zeros(NCAM, 1, hf, wf)
        occ = dec(sem_embed_ds, clip_feats, [supp], lift_depth,
                  [s2e, e2g, intr, pr, pt, bda[None]], [])
        feat_occ = F.interpolate(occ['feat_occ'], size=CFG['occ_size'], mode='trilinear',
                                 align_corners=False)
        bin_occ = F.interpolate(occ['bin_occ'], size=CFG['occ_size'], mode='trilinear',
                                align_corners=False)
        sem_occ = torch.einsum('qc,bczhw->bqzhw', ov, feat_occ)
    res = {'images': images, 'metric': metric, 'depth_scale': torch.tensor(DEPTH_SCALE),
           's2e': s2e, 'e2g': e2g, 'intr': intr, 'pr': pr, 'pt': pt, 'bda': bda,
           'sem_occ': sem_occ, 'bin_occ': bin_occ, 'supp': supp[:, ::8].contiguous(),
           'clip_feat_proj': clip_feats['clip_feat_proj'][:, ::8].contiguous()}
    path = os.path.join(ROOT, 'tests', 'golden', 'path_prod.npz')
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in res.items()})
    print('wrote', path, tuple(sem_occ.shape), 'metric depth %.2f..%.2f m' %
          (float(metric.min()), float(metric.max())),
          'sem_occ rms %.4f range %.3f..%.3f' % (float(sem_occ.pow(2).mean().sqrt()),
                                                 float(sem_occ.min()), float(sem_occ.max())),
          'bin_occ rms %.4f' % float(bin_occ.pow(2).mean().sqrt()),
          os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()

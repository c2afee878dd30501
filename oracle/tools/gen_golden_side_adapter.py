"""Golden vector for SAN's 2-D mask branch.

TEST INFRASTRUCTURE (fixture generation, build container only).  Loaded from
/root/reference unmodified, with name-only stubs for detectron2 / fvcore / timm /
open_clip:

    RegionwiseSideAdapterNetwork, MLPMaskDecoder   side_adapter/side_adaptor_in_veon.py:30-263
    MLP, AddFusion, LayerNorm                      layers.py:9-101
    PatchEmbed                                     side_adapter/timm_wrapper.py:8-45
    FeatureExtractor, RecWithAttnbiasHead.forward  clip_utils/visual.py:23-216

and chained as SANInVeonTemporal.forward does (san_in_veon_temporal.py:123-139,
176-186); the four einsum lines of semantic_inference_2d(_w_embed) (:238-255) are
methods of a class that cannot be imported here and are restated below.

Stand-ins: timm is absent, so the ViT the network wraps is this repo's
``SideAdapterViT`` (timm-named restatement, parity UNPINNED for its blocks) with the
REFERENCE'S PatchEmbed swapped in; open_clip is absent, so the CLIP blocks are this
repo's mirrors (as in gen_golden_clip_head.py).  What this vector pins is everything
the reference owns: query / position tokens, bicubic position resize, fusion points
and AddFusion, the mask decoder, the attention-bias hand-over to the CLIP head and
the 2-D semantic inference.

    python oracle/tools/gen_golden_side_adapter.py -> tests/golden/side_adapter_tiny.npz
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
from gen_golden_body import randomise  # noqa: E402
from gen_golden_hsa import _Registry  # noqa: E402
from veon_amd.models.semantic_net import ClipVisualTrunk  # noqa: E402
from veon_amd.models.semantic_net.side_adapter import SideAdapterViT  # noqa: E402

CFG = dict(clip_width=64, clip_layers=4, clip_heads=2, clip_first_tail=3, clip_proj_dim=24,
           n_classes=5, width=48, depth=4, heads=3, queries=5, vit_image=64,
           fusion_map=('0->0', '1->1', '2->2', '3->3'), deep_supervision_idxs=(4,),
           embed_channels=16, mlp_channels=24, mlp_num_layers=3)
SIZE = (64, 96)          # full-resolution image; the CLIP branch sees half of it


class _BlockBase(torch.nn.Module):      # detectron2.layers.CNNBlockBase: names only
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride


def main():
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
    ref_import._mod('detectron2.utils.logger', log_first_n=lambda *a, **k: None)
    ref_import._mod('detectron2.layers', CNNBlockBase=_BlockBase, Conv2d=torch.nn.Conv2d,
                    ShapeSpec=object)
    ref_import._mod('timm', create_model=None)
    ref_import._mod('timm.models', register_model=lambda f: f)
    ref_import._mod('timm.models.vision_transformer', VisionTransformer=torch.nn.Module,
                    _create_vision_transformer=None)
    ref_import._mod('timm.models.layers', to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v))
    pkg = 'refside'
    ref_import._mod(pkg)
    ref_import.load('mmdet3d/models/semantic_net/attn_helper.py', pkg + '.attn_helper')
    ref_import._mod(pkg + '.clip_utils')
    vis = ref_import.load('mmdet3d/models/semantic_net/clip_utils/visual.py',
                          pkg + '.clip_utils.visual')
    layers = ref_import.load('mmdet3d/models/semantic_net/layers.py', pkg + '.layers')
    ref_import._mod(pkg + '.side_adapter')
    tw = ref_import.load('mmdet3d/models/semantic_net/side_adapter/timm_wrapper.py',
                         pkg + '.side_adapter.timm_wrapper')
    san = ref_import.load('mmdet3d/models/semantic_net/side_adapter/side_adaptor_in_veon.py',
                          pkg + '.side_adapter.side_adaptor_in_veon')

    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(3)
    W, Lr, K = CFG['clip_width'], CFG['clip_layers'], CFG['clip_first_tail']
    trunk = ClipVisualTrunk(image_size=32, patch_size=16, width=W, layers=Lr,
                            heads=CFG['clip_heads']).eval()
    ln_post = torch.nn.LayerNorm(W).eval()
    proj = torch.nn.Parameter(torch.randn(W, CFG['clip_proj_dim'], generator=gen) * W ** -0.5)
    enc = types.SimpleNamespace(
        output_tokens=False, image_size=(32, 32), patch_size=(16, 16),
        grid_size=trunk.grid_size, ln_pre=trunk.ln_pre, input_patchnorm=False,
        patchnorm_pre_ln=torch.nn.Identity(), conv1=trunk.conv1,
        class_embedding=trunk.class_embedding, positional_embedding=trunk.positional_embedding,
        patch_dropout=torch.nn.Identity(), output_dim=CFG['clip_proj_dim'],
        transformer=types.SimpleNamespace(resblocks=trunk.resblocks),
        global_average_pool=False, attn_pool=None, ln_post=ln_post, proj=proj)
    fe = vis.FeatureExtractor(enc, last_layer_idx=K, frozen_exclude=['all']).eval()
    head = vis.RecWithAttnbiasHead(enc, first_layer_idx=K, frozen_exclude=['all'],
                                   sos_token_format='cls_token', sos_token_num=CFG['queries'],
                                   cross_attn=True, downsample_method='bilinear').eval()

    # ---- the side adapter: reference network around the duck-typed ViT
    C = CFG['width']
    vit = SideAdapterViT(CFG['vit_image'], 16, C, CFG['depth'], CFG['heads'])
    vit.patch_embed = tw.PatchEmbed(img_size=CFG['vit_image'], patch_size=16, in_chans=3,
                                    embed_dim=C)
    x2side = {int(j): int(i) for i, j in [x.split('->') for x in CFG['fusion_map']]}
    fusion = torch.nn.ModuleDict({'layer_%d' % t: layers.AddFusion(W, C) for t in x2side})
    dec = san.MLPMaskDecoder(in_channels=C, total_heads=CFG['clip_heads'], total_layers=1,
                             embed_channels=CFG['embed_channels'],
                             mlp_channels=CFG['mlp_channels'],
                             mlp_num_layers=CFG['mlp_num_layers'], rescale_attn_bias=True)
    net = san.RegionwiseSideAdapterNetwork(
        vit_model=vit, fusion_layers=fusion, mask_decoder=dec, num_queries=CFG['queries'],
        fusion_map=x2side, deep_supervision_idxs=list(CFG['deep_supervision_idxs'])).eval()
    randomise(torch.nn.ModuleList([net, trunk, ln_post]), gen)
    with torch.no_grad():
        net.query_embed.normal_(0, 0.5, generator=gen)
        net.query_pos_embed.normal_(0, 0.5, generator=gen)
        vit.pos_embed.normal_(0, 0.5, generator=gen)
    ov = torch.randn(CFG['n_classes'], CFG['clip_proj_dim'], generator=gen)

    images = torch.randn(2, 3, *SIZE, generator=gen)
    with torch.no_grad():
        clip_in = F.interpolate(images, scale_factor=0.5, mode='bilinear')
        feats = fe(clip_in)
        mask_preds, attn_biases, san_feats = net(images, feats)
        mask_embs = [head(feats, ab, normalize=True) for ab in attn_biases]
        mask_logits = [torch.einsum('bqc,nc->bqn', e, ov) for e in mask_embs]
        # semantic_inference_2d_w_embed / semantic_inference_2d (:238-255), restated
        cls = F.softmax(mask_logits[-1], dim=-1)[..., :-1]
        mp = mask_preds[-1].sigmoid()
        sem_seg_ds = torch.einsum('bqc,bqhw->bchw', cls, mp)
        sem_embed_ds = torch.einsum('bqc,bqhw->bchw', mask_embs[-1], mp)
        up = F.interpolate(mask_preds[-1], size=SIZE, mode='bilinear', align_corners=False)
        sem_seg = torch.einsum('bqc,bqhw->bchw', cls, up.sigmoid())

    res = {'images': images, 'ov_classifier_weight': ov, 'clip_proj': proj.detach(),
           'mask_preds': mask_preds[-1], 'attn_bias': attn_biases[-1][0],
           'mask_embs': mask_embs[-1], 'mask_logits': mask_logits[-1],
           'sem_seg_ds': sem_seg_ds, 'sem_embed_ds': sem_embed_ds, 'sem_seg': sem_seg}
    for i, f in enumerate(san_feats):
        res['san_feat_%d' % i] = f
    for i in range(K + 1):
        res['clip_feat_%d' % i] = feats[i]
    for k, v in net.state_dict().items():
        res['net/' + k] = v
    for k, v in trunk.state_dict().items():
        res['trunk/' + k] = v
    for k, v in ln_post.state_dict().items():
        res['ln_post/' + k] = v
    path = os.path.join(ROOT, 'tests', 'golden', 'side_adapter_tiny.npz')
    np.savez_compressed(path, **{k: v.detach().float().numpy() for k, v in res.items()})
    print('wrote', path, os.path.getsize(path), 'bytes;', mask_preds[-1].shape,
          attn_biases[-1][0].shape, sem_embed_ds.shape)


if __name__ == '__main__':
    main()

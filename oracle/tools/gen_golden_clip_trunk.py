"""Golden vectors for the CLIP visual trunk wiring.

TEST INFRASTRUCTURE (fixture generation, build container only).  The reference's
own ``clip_utils/visual.py`` ``FeatureExtractor`` (:22-91) and
``attn_helper.resize_pos_embed2d`` are loaded unmodified (name-only stubs for
the open_clip / detectron2 type imports) and run around a duck-typed visual
encoder whose sub-modules are this repo's mirrors with seeded weights.  That
pins what the reference owns -- patchify, class token, position-embedding
resize, ln_pre, LND layout, per-block outputs -- while the transformer block
itself (open_clip, absent) stays a restatement.

    python oracle/tools/gen_golden_clip_trunk.py -> tests/golden/clip_trunk_tiny.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from veon_amd.models.semantic_net import ClipVisualTrunk  # noqa: E402


def main():
    ref_import._mod('open_clip')
    ref_import._mod('open_clip.transformer', ResidualAttentionBlock=torch.nn.Module,
                    VisionTransformer=torch.nn.Module)
    ref_import._mod('detectron2')
    ref_import._mod('detectron2.layers', ShapeSpec=object)
    pkg = 'refsem2'
    ref_import._mod(pkg)
    ref_import.load('mmdet3d/models/semantic_net/attn_helper.py', pkg + '.attn_helper')
    ref_import._mod(pkg + '.clip_utils')
    vis = ref_import.load('mmdet3d/models/semantic_net/clip_utils/visual.py',
                          pkg + '.clip_utils.visual')
    torch.manual_seed(0)
    trunk = ClipVisualTrunk(image_size=64, patch_size=16, width=64, layers=2, heads=1).eval()
    enc = types.SimpleNamespace(
        output_tokens=False, image_size=(64, 64), patch_size=(16, 16),
        grid_size=trunk.grid_size, ln_pre=trunk.ln_pre, input_patchnorm=False,
        patchnorm_pre_ln=torch.nn.Identity(), conv1=trunk.conv1,
        class_embedding=trunk.class_embedding,
        positional_embedding=trunk.positional_embedding,
        patch_dropout=torch.nn.Identity(),
        transformer=types.SimpleNamespace(resblocks=trunk.resblocks))
    fe = vis.FeatureExtractor(enc, last_layer_idx=-1, frozen_exclude=['all']).eval()
    x = torch.randn(2, 3, 32, 48)
    with torch.no_grad():
        out = fe(x)
    res = {'x': x.numpy()}
    for k, v in trunk.state_dict().items():
        res['sd/' + k] = v.numpy()
    for i in range(3):
        res['feat_%d' % i] = out[i].numpy()                  # (n, c, h, w)
        res['cls_%d' % i] = out['%d_cls_token' % i].numpy()  # (1, n, c)
    res['hw'] = np.array(out.spacial_shape)
    path = os.path.join(ROOT, 'tests', 'golden', 'clip_trunk_tiny.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, out.spacial_shape, out[2].shape)


if __name__ == '__main__':
    main()

"""Golden vectors for the DepthAnythingV2 / DINOv2 encoder, produced by running
the reference's own Python (mmdet3d/models/depth_anything/*) here on CPU with
tiny dimensions and seeded random weights (no pretrained weights exist in the
container).  Run from the repo root:  python oracle/tools/gen_golden_vit.py
Outputs tests/golden/dinov2_tiny.npz (state dict + input + expected outputs).
"""
import importlib
import os
import sys
import types
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get('VEON_REFERENCE', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _pkg(name, path=None, **attrs):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def load_reference():
    # third-party names the reference imports but never calls on this path
    _pkg('cv2', INTER_CUBIC=2, INTER_AREA=3, INTER_LINEAR=1, INTER_NEAREST=0)
    _pkg('torchvision')
    _pkg('torchvision.transforms', Compose=lambda x: x)
    _pkg('mmdet3d')
    _pkg('mmdet3d.models')
    _pkg('mmdet3d.models.builder', NECKS=_Registry())
    _pkg('mmdet3d.models.depth_anything',
         os.path.join(REF, 'mmdet3d/models/depth_anything'))
    dpt = importlib.import_module('mmdet3d.models.depth_anything.dpt')
    dino = importlib.import_module('mmdet3d.models.depth_anything.dinov2')
    return dpt, dino


def main():
    torch.manual_seed(0)
    dpt, dino = load_reference()
    d, depth, heads, lora_r = 64, 4, 1, 4
    enc = dino.DinoVisionTransformer(
        img_size=70, patch_size=14, embed_dim=d, depth=depth, num_heads=heads,
        mlp_ratio=4, init_values=1.0, ffn_layer='mlp', block_chunks=0,
        num_register_tokens=0, interpolate_antialias=False,
        interpolate_offset=0.1, lora_r=lora_r,
        block_fn=partial(dino.Block, attn_class=dino.MemEffAttention))
    head = dpt.DPTHead(d, features=8, use_bn=False, out_channels=[4, 8, 16, 16],
                       use_clstoken=False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in list(enc.named_parameters()) + list(head.named_parameters()):
            if n.endswith('lora_B'):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif n.endswith('gamma'):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith('bias'):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif 'norm' in n and n.endswith('weight'):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2 and 'lora_A' not in n:
                p.copy_(p.shape[-1] ** -0.5 * torch.randn(p.shape, generator=g))
    enc.train()
    head.train()
    sd = {('enc.' + k): v.clone() for k, v in enc.state_dict().items()}
    sd.update({('head.' + k): v.clone() for k, v in head.state_dict().items()})
    enc.eval()
    head.eval()
    x = torch.randn(2, 3, 28, 42, generator=g)
    taps = [0, 1, 2, 3]
    with torch.no_grad():
        feats = enc.get_intermediate_layers(x, taps, return_class_token=True)
        final = enc.forward_features(x)
        depth_map = head(feats, 2, 3) * 80.0
    out = {k: v.numpy() for k, v in sd.items()}
    out.update(
        cfg=np.array([d, depth, heads, lora_r]), taps=np.array(taps), x=x.numpy(),
        tap_patch=np.stack([f[0].numpy() for f in feats]),
        tap_cls=np.stack([f[1].numpy() for f in feats]),
        x_prenorm=final['x_prenorm'].numpy(),
        x_norm_clstoken=final['x_norm_clstoken'].numpy(),
        depth=depth_map.squeeze(1).numpy())
    np.savez_compressed(os.path.join(GOLD, 'dinov2_tiny.npz'), **out)
    print('saved', {k: v.shape for k, v in out.items() if not k.startswith(('enc.', 'head.'))})


if __name__ == '__main__':
    main()

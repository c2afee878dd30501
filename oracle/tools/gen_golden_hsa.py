"""Golden vectors for the high-resolution side adaptor network.

TEST INFRASTRUCTURE (fixture generation, build container only).  The reference's
own ``side_adapter/highres_side_adaptor.py`` is loaded unmodified (name-only
stubs for detectron2's ``configurable`` / ``Registry`` / ``ShapeSpec``) and its
``HighresSideAdaptorNetwork`` is built from the reference classes with tiny
dimensions and run on CPU.

    python oracle/tools/gen_golden_hsa.py -> tests/golden/hsa_tiny.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


class _Registry:
    def __init__(self, name):
        self.d = {}

    def register(self):
        def deco(cls):
            self.d[cls.__name__] = cls
            return cls
        return deco

    def get(self, name):
        return self.d[name]


def main():
    ref_import._mod('detectron2')
    ref_import._mod('detectron2.config', configurable=lambda f: f)
    ref_import._mod('detectron2.utils')
    ref_import._mod('detectron2.utils.registry', Registry=_Registry)
    ref_import._mod('detectron2.layers', ShapeSpec=object)
    hsa = ref_import.load(
        'mmdet3d/models/semantic_net/side_adapter/highres_side_adaptor.py', 'ref_hsa')
    torch.manual_seed(0)
    dim, clip_dim, mlp = 64, 32, 64
    fusion_map = ['0->1->1', '1->2->-1']
    cr_map = {int(i): (int(j), int(k)) for i, j, k in [x.split('->') for x in fusion_map]}
    pe = hsa.PatchEmbed((32, 48), (8, 8), embed_dim=dim, norm_layer=False)
    body = torch.nn.ModuleList([
        hsa.HighresSideAdaptorBlock(dim=dim, neck_dim=clip_dim, mlp_dim=mlp,
                                    pre_norm=(i == 0), use_add=cr_map[i][1] >= 0,
                                    use_checkpoint=False) for i in range(2)])
    rear = hsa.AttnManipulateBlock(dim=dim, mlp_dim=mlp, clip_dim=clip_dim, heads=2,
                                   dim_head=8, attn_layers=3, add_layers=2, supp_dim=16,
                                   pre_norm=False, use_checkpoint=False)
    net = hsa.HighresSideAdaptorNetwork(pe, body, rear, cr_map, use_checkpoint=False).eval()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    image = torch.randn(2, 3, 32, 48, generator=gen)
    clip = {1: torch.randn(2, clip_dim, 2, 3, generator=gen),
            2: torch.randn(2, clip_dim, 2, 3, generator=gen)}
    with torch.no_grad():
        _, attns, supp = net(image, clip)
        tok = torch.randn(2, 24, dim, generator=gen)
        cb = net.hsa_net_body[0].ff(tok, (4, 6))
    res = {'image': image, 'clip1': clip[1], 'clip2': clip[2], 'attns': attns,
           'supp': supp, 'tokens': tok, 'convblock_out': cb}
    res.update({'sd/' + k: v for k, v in net.state_dict().items()})
    path = os.path.join(ROOT, 'tests', 'golden', 'hsa_tiny.npz')
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in res.items()})
    print('wrote', path, tuple(attns.shape), tuple(supp.shape))


if __name__ == '__main__':
    main()

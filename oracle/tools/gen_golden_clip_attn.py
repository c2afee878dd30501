"""Golden vectors for SAN's query-token attention.

TEST INFRASTRUCTURE (fixture generation, build container only).  The reference's
own ``attn_helper.py`` (pure torch; its only third-party import,
``open_clip.transformer.ResidualAttentionBlock``, is a type annotation) is
loaded unmodified with a name-only stub for that import, and
``cross_attn_with_self_bias`` (attn_helper.py:10-300) is run on a seeded
``nn.MultiheadAttention``.  This pins the part of row a15 the reference owns;
open_clip's block itself stays unpinned (absent, version not stated).

    python oracle/tools/gen_golden_clip_attn.py -> tests/golden/clip_cross_attn.npz
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
    ref_import._mod('open_clip')
    ref_import._mod('open_clip.transformer', ResidualAttentionBlock=torch.nn.Module)
    ah = ref_import.load('mmdet3d/models/semantic_net/attn_helper.py', 'ref_attn_helper')
    torch.manual_seed(0)
    D, H, K, L, N = 64, 4, 5, 11, 2
    mha = torch.nn.MultiheadAttention(D, H).eval()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1)
        mha.out_proj.bias.normal_(0, 0.1)
    q = torch.randn(K, N, D)
    mem = torch.randn(L, N, D)
    bias = torch.randn(N * H, K, L)
    with torch.no_grad():
        out_bias = ah.cross_attn_with_self_bias(mha, q, mem, mem, attn_mask=bias)[0]
        out_none = ah.cross_attn_with_self_bias(mha, q, mem, mem, attn_mask=None)[0]
        bmask = torch.rand(N * H, K, L) < 0.3
        out_bool = ah.cross_attn_with_self_bias(mha, q, mem, mem, attn_mask=bmask)[0]
    out = {'q': q, 'mem': mem, 'bias': bias, 'bool_mask': bmask, 'out_bias': out_bias,
           'out_none': out_none, 'out_bool': out_bool}
    for k, v in mha.state_dict().items():
        out['mha/' + k] = v
    path = os.path.join(ROOT, 'tests', 'golden', 'clip_cross_attn.npz')
    np.savez_compressed(path, **{k: v.numpy() for k, v in out.items()})
    print('wrote', path, out_bias.shape)


if __name__ == '__main__':
    main()

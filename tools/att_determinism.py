#!/usr/bin/env python
"""Determinism stress of the attention kernel: it is deterministic by construction, so
any difference between two launches on the same input is a fault ("A wrong
result that came and went").  Counts launches and 16-query tiles that differ from the
first launch.

    N=300 CASES="((6,901,12,0,0.9),)" python tools/att_determinism.py
    VEON_HIP_LIB=/path/to/variant.so python tools/att_determinism.py     # A/B a build
CASES entries: (B, T, H, with_bias, input scale); scale 0.9 makes the reference maximum
move for ~20 % of the queries."""
import ast
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import vit_ops  # noqa: E402

N = int(os.environ.get('N', '300'))
CASES = ast.literal_eval(os.environ.get(
    'CASES', '((6, 901, 12, 0, 0.9), (6, 901, 16, 0, 0.9), (6, 705, 12, 1, 0.7))'))
LOG2Q = os.environ.get('VEON_ATT_RAW_Q') != '1'


def main():
    torch.manual_seed(1)
    for B, T, H, wb, scale in CASES:
        qkv = (torch.randn(B, T, 3 * H * 64, device='cuda') * scale).bfloat16()
        bias = torch.randn(B, H, T, T, device='cuda') * 2 if wb else None
        out = torch.empty(B, T, H * 64, device='cuda', dtype=qkv.dtype)
        ref = vit_ops.attention(qkv, H, bias, q_log2=LOG2Q).clone()
        events = launches_bad = 0
        for _ in range(N):
            vit_ops.attention(qkv, H, bias, out=out, q_log2=LOG2Q)
            diff = (out != ref).view(B, T, H, 64).any(-1)
            if bool(diff.any()):
                launches_bad += 1
                events += len({(b, t // 16, h) for b, t, h in diff.nonzero().tolist()})
        print('%s B%d T%d H%d bias%d scale %.1f: %d launches, %d differ from the first, '
              '%d differing 16-query tiles'
              % (os.path.basename(os.environ.get('VEON_HIP_LIB', 'libveon_hip.so')), B, T, H,
                 wb, scale, N, launches_bad, events), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""HSA network (highres_side_adaptor.py) at VEON shapes: PyTorch fp32 / bf16
autocast vs the ConvBlocks on the MFMA conv kernel.  6 cameras."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd.models.semantic_net.hsa_network import HighresSideAdaptorNetwork  # noqa: E402
from tools.hotpath_bench import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    only_hip = os.environ.get('ONLY_HIP') == '1'   # for a rocprofv3 trace of the MFMA path
    for size in ((256, 704), (512, 1408))[1 if only_hip else 0:]:
        net = HighresSideAdaptorNetwork.build(input_size=size).to(dev).eval()
        img = torch.randn(6, 3, *size, device=dev)
        h, w = size[0] // 32, size[1] // 32          # CLIP sees the image at 1/2, patch 16
        clip = {i: torch.randn(6, 768, h, w, device=dev) for i in (1, 3, 6, 9)}
        toks = (size[0] // 8) * (size[1] // 8)
        fl = 8 * 2.0 * 6 * toks * 384 * 384 * 9
        with torch.no_grad():
            t32 = t16 = float('nan')
            if not only_hip:
                t32 = timeit(lambda: net(img, clip), 5)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    t16 = timeit(lambda: net(img, clip), 5)
            net.set_conv_dtype(torch.bfloat16)
            tm = timeit(lambda: net(img, clip), 20)
            blk = net.hsa_net_body[1].ff
            x = torch.randn(6, toks, 384, device=dev)
            tb = timeit(lambda: blk(x, (size[0] // 8, size[1] // 8)), 10)
            net.set_conv_dtype(None)
            tb32 = float('nan') if only_hip else timeit(
                lambda: blk(x, (size[0] // 8, size[1] // 8)), 5)
        print('%dx%d (%d tokens/cam): torch fp32 %.2f ms | torch bf16 autocast %.2f ms | '
              'MFMA ConvBlocks %.2f ms   [one ConvBlock: %.3f ms = %.0f TF/s on its two '
              'convs; torch fp32 %.3f ms]' % (size[0], size[1], toks, t32, t16, tm, tb,
                                             fl / 4 / (tb * 1e-3) / 1e12, tb32), flush=True)


if __name__ == '__main__':
    main()

import sys, torch
sys.path.insert(0, '/root/repo')
from veon_amd import _lib, vit_ops
from tools.gemm_bench import timeit
L = _lib.lib()
dev = 'cuda:0'
for name, M, N, K in (('B qkv', 5406, 2304, 768), ('B fc2', 5406, 768, 3072)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    b = torch.randn(N, device=dev)
    for cfg in (1, 2, 7, 4):
        row = []
        for abl, what in ((0, 'full'), (1, 'no-store'), (9, 'dma+barrier only'), (11, 'barrier only')):
            L.veon_gemm_ring_set(cfg | (abl << 8))
            row.append('%s %.1f' % (what, timeit(lambda: vit_ops.linear(a, w, b))))
        print(name, 'cfg', cfg, ' | '.join(row), flush=True)
L.veon_gemm_ring_set(-1)

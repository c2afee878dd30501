#!/usr/bin/env python
"""Residual-epilogue GEMM (proj / fc2 of the ViT blocks) by tile configuration
(veon_gemm_ring_set): the dispatcher keeps it on the small-tile kernel.  Not a test."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import _lib, vit_ops
from tools.gemm_bench import timeit
dev='cuda:0'
L=_lib.lib()
g=torch.Generator(device=dev).manual_seed(0)
for name,M,N,K in (('B fc2',5406,768,3072),('L fc2',5406,1024,4096),('L proj',5406,1024,1024),('B proj',5406,768,768)):
    a=(torch.rand(M,K,device=dev,generator=g)*2-1).bfloat16()
    w=((torch.rand(N,K,device=dev,generator=g)*2-1)*K**-0.5).bfloat16()
    b=torch.randn(N,device=dev,generator=g)
    gam=torch.rand(N,device=dev,generator=g)
    r=torch.zeros(M,N,device=dev)
    line=name
    for cfg in (0,7,4,3,1):
        L.veon_gemm_ring_set(cfg)
        t=min(timeit(lambda: vit_ops.linear_residual_(r,a,w,b,gam)) for _ in range(3))
        line+='  cfg%d %.1f'%(cfg,t)
    L.veon_gemm_ring_set(-1)
    print(line, flush=True)

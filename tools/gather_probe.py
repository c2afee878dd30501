#!/usr/bin/env python
"""Probe: how much of the pool kernel's time is gather-miss latency?  Same
structure, but every point reads feat row 0 / depth 0 (all gathers L1-hot)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._inputs import lift_case
from veon_amd import _lib, synthetic
from veon_amd.ops.bev_pool_v2 import bev_pool as bp

def timeit(fn, iters=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

grid, size, cams, C = synthetic.GRID_S2, (256, 704), 6, 80
dev = 'cuda:0'
case = lift_case(grid, size, cams, C, dev)
depth, feat = case['depth'], case['feat_nhwc']
rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
bp.mark_sorted(st, 0, 640000 - 1)
bp.build_plan(rb, st, 1, 640000)
shape = (1, 16, 200, 200, C)
z = torch.zeros_like(rf)
for name, rff, rdd in (('real gathers', rf, rd), ('all rows = 0', z, z),
                       ('rows mod 64', rf % 64, rd % 64), ('rows mod 1024', rf % 1024, rd % 1024)):
    t = timeit(lambda: bp._fused_forward(depth, feat, rdd, rff, rb, st, ln, shape, _lib.LAYOUT_BCZYX))
    print('%-16s %.2f us' % (name, t))

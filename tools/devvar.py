#!/usr/bin/env python
"""Which device is this, and how fast are fill / pool on it?  (run-to-run study)"""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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

p = torch.cuda.get_device_properties(0)
try:
    uid = subprocess.run(['rocm-smi', '--showuniqueid'], capture_output=True, text=True).stdout.strip().splitlines()
    uid = [l for l in uid if 'Unique' in l][:1]
except Exception as e:
    uid = [repr(e)]
print('device', p.name, 'cus', p.multi_processor_count, uid, 'host', os.uname().nodename)
grid, size, cams, C = synthetic.GRID_S2, (256, 704), 6, 80
dev = 'cuda:0'
case = lift_case(grid, size, cams, C, dev)
depth, feat = case['depth'], case['feat_nhwc']
rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
bp.mark_sorted(st, 0, 640000 - 1)
bp.build_plan(rb, st, 1, 640000)
shape = (1, 16, 200, 200, C)
out = torch.empty(1, C, 16, 200, 200, device=dev)
for rep in range(3):
    tz = timeit(lambda: out.zero_())
    tp = timeit(lambda: bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX))
    tl = timeit(lambda: bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BZYXC))
    print('rep %d: zero_fill %.2f us | pool cf %.2f us | pool cl %.2f us' % (rep, tz, tp, tl))

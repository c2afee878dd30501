#!/usr/bin/env python
"""Kernel microbench on the GPU box: product pool kernels vs experimental
variants (tools/ubench/pool_variants.hip), interleaved rounds in one process,
HIP-event timed.  Not part of the product or the tests.

    python tools/kbench.py [S2|SV] [rounds]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._inputs import lift_case  # noqa: E402
from veon_amd import _lib, synthetic  # noqa: E402
from veon_amd.ops.bev_pool_v2 import bev_pool as bp  # noqa: E402

UB = os.path.join(ROOT, 'tools', 'ubench')


def build_variants():
    so = os.path.join(UB, 'libpoolvar.so')
    src = os.path.join(UB, 'pool_variants.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                               '-ffp-contract=off', '-fPIC', '-shared', src, '-o', so])
    return ctypes.CDLL(so)


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'S2'
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cfg = {'S2': (synthetic.GRID_S2, (256, 704), 6, 80),
           'SV': (synthetic.GRID_VEON, (512, 1408), 6, 256)}[tag]
    grid, size, cams, C = cfg
    dev = 'cuda:0'
    case = lift_case(grid, size, cams, C, dev)
    depth, feat = case['depth'], case['feat_nhwc']
    rb, rd, rf, st, ln = (case[k] for k in ('rb', 'rd', 'rf', 'st', 'ln'))
    D = case['D']
    hf, wf = size[0] // 16, size[1] // 16
    X, Y, Z = case['gsize']
    vpb = Z * Y * X
    shape = (1, Z, Y, X, C)
    alg = 4 * (cams * hf * wf * C + cams * D * hf * wf + 3 * rb.numel() +
               2 * st.numel() + vpb * C)
    print('%s: P_kept=%d I=%d out=%.1f MB alg=%.1f MB' %
          (tag, rb.numel(), st.numel(), vpb * C * 4 / 1e6, alg / 1e6))
    bp.mark_sorted(st, int(rb[0]), int(rb[-1]))

    L = _lib.lib()
    V = build_variants()
    n_tiles = (vpb + 63) // 64
    tf = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    tp = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    s = _lib.stream_ptr(torch.device(dev))
    assert V.poolvar_plan(st.numel(), rb.numel(), 1, ctypes.c_int64(vpb), _lib.ptr(rb),
                          _lib.ptr(st), _lib.ptr(tf), _lib.ptr(tp), s) == 0
    torch.cuda.synchronize()
    tf2, tp2 = torch.full_like(tf, -7), torch.full_like(tp, -7)
    assert V.poolvar_plan2(st.numel(), rb.numel(), 1, ctypes.c_int64(vpb), _lib.ptr(rb),
                           _lib.ptr(st), _lib.ptr(tf2), _lib.ptr(tp2), s) == 0
    torch.cuda.synchronize()
    print('plan2 == plan:', torch.equal(tf, tf2), torch.equal(tp, tp2))
    print('plan  us', timeit(lambda: V.poolvar_plan(st.numel(), rb.numel(), 1, ctypes.c_int64(vpb), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(tf2), _lib.ptr(tp2), s)))
    print('plan2 us', timeit(lambda: V.poolvar_plan2(st.numel(), rb.numel(), 1, ctypes.c_int64(vpb), _lib.ptr(rb), _lib.ptr(st), _lib.ptr(tf2), _lib.ptr(tp2), s)))
    plan4_full = torch.zeros((n_tiles + 1) * 4, dtype=torch.int32, device=dev)
    plan4 = plan4_full[4:]
    assert V.poolvar_plan4(1, ctypes.c_int64(vpb), _lib.ptr(tf), _lib.ptr(tp), _lib.ptr(plan4), s) == 0
    torch.cuda.synchronize()
    tfc, tpc = tf.cpu().numpy().astype(np.int64), tp.cpu().numpy().astype(np.int64)
    cnts, nps = np.diff(tfc), np.diff(tpc)
    order = np.argsort(-nps, kind='stable') if not os.environ.get('KB_NOSORT') else np.arange(n_tiles)
    p8 = np.stack([order, tfc[:-1][order], tpc[:-1][order], (cnts[order] << 24) | nps[order]], 1).astype(np.int32)
    plan8 = torch.from_numpy(p8).to(dev).contiguous()
    V.poolvar_set_plan8(ctypes.c_void_p(plan8.data_ptr()))
    # 256-B tile records (cached plan v2) built on the host for the experiment
    rbc, rfc, rdc, stc = (x.cpu().numpy().astype(np.int64) for x in (rb, rf, rd, st))
    recs = np.zeros((n_tiles, 64), np.int32)
    r8 = recs.view(np.uint8).reshape(n_tiles, 256)
    nbig = 0
    for t in range(n_tiles):
        c_, n_ = int(cnts[t]), int(nps[t])
        if c_ == 0:
            continue
        if c_ > 24 or n_ > 24:
            recs[t, 0] = 255
            nbig += 1
            continue
        i0_, p0_ = int(tfc[t]), int(tpc[t])
        recs[t, 0] = c_ | (n_ << 8)
        starts = stc[i0_:i0_ + c_]
        r8[t, 4:4 + c_] = (rbc[starts] - t * 64).astype(np.uint8)
        r8[t, 28:28 + c_] = (starts - p0_).astype(np.uint8)
        r8[t, 28 + c_] = n_
        recs[t, 16:16 + n_] = rfc[p0_:p0_ + n_]
        recs[t, 40:40 + n_] = rdc[p0_:p0_ + n_]
    print('records: %d tiles on the general path' % nbig)
    recs_d = torch.from_numpy(recs).to(dev).contiguous()
    V.poolvar_set_recs(ctypes.c_void_p(recs_d.data_ptr()))
    empty = int((tf[1:] == tf[:-1]).sum())
    print('tiles=%d empty=%d (%.0f%%)' % (n_tiles, empty, 100.0 * empty / n_tiles))

    ref = bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)
    out = torch.empty_like(ref)

    def product_cf_search():  # plan rebuilt per call
        if hasattr(st, '_veon_plan'):
            del st._veon_plan
        return bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)

    def product_cf_table():  # plan cached
        if not hasattr(st, '_veon_plan'):
            bp.build_plan(rb, st, 1, vpb)
        return bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)

    def product_cl():
        if not hasattr(st, '_veon_plan'):
            bp.build_plan(rb, st, 1, vpb)
        return bp._fused_forward(depth, feat, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BZYXC)

    def zero_fill():
        out.zero_()

    feat_h = feat.half()
    feat_b = feat.bfloat16()

    def product_cf_f16():
        return bp._fused_forward(depth, feat_h, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)

    def product_cf_bf16():
        return bp._fused_forward(depth, feat_b, rd, rf, rb, st, ln, shape, _lib.LAYOUT_BCZYX)

    bp.build_row_table(rb, st, 1, vpb, X)

    def product_maxpool():
        return bp.bev_pool_v2_maxpool(depth, feat, rd, rf, rb, shape, st, ln, (2, 2, 2))

    def product_maxpool_f16():
        return bp.bev_pool_v2_maxpool(depth, feat_h, rd, rf, rb, shape, st, ln, (2, 2, 2))

    def var(v, cs):
        def f():
            r = V.poolvar_run(v, C, cs, st.numel(), 1, ctypes.c_int64(vpb), _lib.ptr(depth),
                              _lib.ptr(feat), _lib.ptr(rd), _lib.ptr(rf), _lib.ptr(rb),
                              _lib.ptr(st), _lib.ptr(ln), _lib.ptr(tf), _lib.ptr(tp),
                              _lib.ptr(out), s)
            assert r == 0, r
        return f

    if os.environ.get('KB_STAMPS'):
        cs = C if C <= 128 else 64
        slabs = (C + cs - 1) // cs
        stamps = torch.zeros(n_tiles * slabs * 8, dtype=torch.int64, device=dev)
        V.poolvar_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
        plan4_full[:2] = torch.tensor([stamps.data_ptr() & 0xffffffff, stamps.data_ptr() >> 32], dtype=torch.int64).to(torch.int32).to(dev)
        f = var(int(os.environ['KB_STAMPS']), cs)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        stamps.zero_()
        f()
        torch.cuda.synchronize()
        sp8 = stamps.cpu().numpy().reshape(-1, 8).astype(np.int64)
        occ = sp8[:, 3] > 0
        if occ.any():
            ph = np.diff(sp8[occ][:, :6], axis=1) * 0.01
            print('phases (occupied tiles) mean us: plan %.2f  prologue(meta+stage) %.2f  gather %.2f  store-issue %.2f  store-drain %.2f' % tuple(ph.mean(0)))
            print('phases p90: ' + ' '.join('%.2f' % x for x in np.percentile(ph, 90, axis=0)))
        sp = sp8[:, [0, 5]]
        t0 = sp[:, 0].min()
        start = (sp[:, 0] - t0) * 0.01  # us (100 MHz wall clock)
        end = (sp[:, 1] - t0) * 0.01
        dur = end - start
        tfc = tf.cpu().numpy()
        tpc = tp.cpu().numpy()
        cnt = np.repeat(np.diff(tfc), slabs)
        pts = np.repeat(np.diff(tpc), slabs)
        print('kernel span %.1f us; WGs %d' % (end.max(), len(dur)))
        print('dur us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f' % (
            dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90),
            np.percentile(dur, 99), dur.max()))
        for lo, hi in ((0, 1), (1, 8), (8, 32), (32, 128), (128, 512), (512, 100000)):
            m = (pts >= lo) & (pts < hi)
            if m.any():
                print('  pts [%d,%d): n=%d dur mean %.2f max %.2f' % (lo, hi, m.sum(), dur[m].mean(), dur[m].max()))
        order = np.argsort(-end)[:10]
        for i in order:
            print('  late WG %d: start %.1f end %.1f dur %.1f cnt %d pts %d' % (i, start[i], end[i], dur[i], cnt[i], pts[i]))
        # concurrency over time
        for tt in np.linspace(0, end.max(), 12):
            print('  t=%.1f running=%d done=%d' % (tt, ((start <= tt) & (end > tt)).sum(), (end <= tt).sum()))
        return
    cs_opts = [C] if C <= 128 else [64, 128]
    cases = [('zero_fill(torch)', zero_fill), ('product_cf_percall_plan', product_cf_search),
             ('product_cl', product_cl), ('product_cf_cached_plan', product_cf_table),
             ('product_cf_f16', product_cf_f16), ('product_cf_bf16', product_cf_bf16),
             ('product_maxpool', product_maxpool), ('product_maxpool_f16', product_maxpool_f16)]
    queue = torch.zeros(4, dtype=torch.int32, device=dev)
    def var5(v, cs, wpc):
        f = var(v, cs)
        def g():
            V.poolvar_set_queue(ctypes.c_void_p(0), wpc)
            f()
        return g
    for v in []:
        for cs in cs_opts:
            cases.append(('var%d cs=%d' % (v, cs), var(v, cs)))
    for v in []:
        for cs in cs_opts:
            for wpc in (4, 5, 6, 7, 8):
                cases.append(('var%d cs=%d wpc=%d' % (v, cs, wpc), var5(v, cs, wpc)))
    # correctness of variants
    for name, fn in cases:
        if name.startswith('var'):
            out.fill_(float('nan'))
            fn()
            torch.cuda.synchronize()
            ok = torch.equal(out, ref)
            print('%-26s %s' % (name, 'bit-exact' if ok else 'MISMATCH(expected for ablations)'))
    res = {n: [] for n, _ in cases}
    for _ in range(rounds):
        for n, fn in cases:
            res[n].append(timeit(fn))
    for n, _ in cases:
        t = np.array(res[n])
        print('%-26s min %7.2f us  med %7.2f us   %6.0f GB/s (alg, med)' %
              (n, t.min(), np.median(t), alg / np.median(t) / 1e3))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Whole-forward hipGraph of the occupancy path (BASELINE configs[2] / [4]:
"hipGraph-captured forward"): capture VeonOccupancyPath.forward once, replay it,
check the replays against eager outputs, then capture a SECOND graph (the S2 lift
step of bench.py) and replay the first again -- the situation in which a lift
graph faulted in round 1 -- and time eager vs replay.

    python tools/graph_path.py [vitb|vitl] [--veon-res] [--sparse]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import synthetic  # noqa: E402
from veon_amd.models import build_neck  # noqa: E402
from veon_amd.models.veon_occ import VeonOccupancyPath  # noqa: E402


def wall(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    enc = 'vitl' if 'vitl' in sys.argv else 'vitb'
    size = (512, 1408) if '--veon-res' in sys.argv else (256, 704)
    dev = 'cuda:0'
    torch.manual_seed(0)
    kw = dict(VeonOccupancyPath.VEON_L) if enc == 'vitl' else dict(encoder='vitb')
    if '--dense' in sys.argv:       # the reference's dense two-hot tensor into the lift
        kw['sparse_lift_eps'] = None  # (default: two-hot lift by construction, eps 1e-6)
    net = VeonOccupancyPath(input_size=size, **kw).to(dev).eval()
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    images = torch.randn(1, 6, 3, *size, device=dev)
    with torch.no_grad():
        for _ in range(3):
            ref = net(images, geom)
        torch.cuda.synchronize()
        ref = {k: v.clone() for k, v in ref.items()}
        print('eager ok', {k: tuple(v.shape) for k, v in ref.items()}, flush=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(images, geom)
        torch.cuda.current_stream().wait_stream(side)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=side):   # the stream the warm-up ran on
            out = net(images, geom)
        print('captured the whole forward', flush=True)
        for i in range(3):
            g1.replay()
            torch.cuda.synchronize()
            same = {k: bool(torch.equal(out[k], ref[k])) for k in ref}
            close = {k: float((out[k].float() - ref[k].float()).abs().max()) for k in ref}
            print('replay %d: equal %s  max|diff| %s' % (i, same, close), flush=True)
        # new inputs through the static input tensors
        images2 = torch.randn_like(images)
        ref2 = {k: v.clone() for k, v in net(images2, geom).items()}
        images.copy_(images2)
        g1.replay()
        torch.cuda.synchronize()
        print('new input: equal', {k: bool(torch.equal(out[k], ref2[k])) for k in ref2},
              flush=True)

        # ---- a second graph (another module's lift), then the first again
        vt = build_neck(dict(type='LSSViewTransformer', grid_config=synthetic.GRID_S2,
                             input_size=(256, 704), downsample=16, in_channels=8,
                             out_channels=80, accelerate=False, collapse_z=False)).to(dev).eval()
        vt.sync_free = True
        d5, f5 = synthetic.make_depth_feat(1, 6, vt.D, 80, 16, 44, seed=0, device=dev)
        g2geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, (256, 704)))]

        def lift():
            return vt.view_transform([f5] + g2geom, d5.view(6, vt.D, 16, 44),
                                     f5.view(6, 80, 16, 44))[0]
        lift()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            lift()
        torch.cuda.current_stream().wait_stream(side)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            o2 = lift()
        print('captured a second graph', flush=True)
        g2.replay()
        torch.cuda.synchronize()
        print('second graph replayed, sum %.4f' % float(o2.sum()), flush=True)
        g1.replay()
        torch.cuda.synchronize()
        print('first graph replayed AFTER the second: equal',
              {k: bool(torch.equal(out[k], ref2[k])) for k in ref2}, flush=True)
        g2.replay()
        g1.replay()
        torch.cuda.synchronize()
        t_e = wall(lambda: net(images, geom), 20)
        t_g = wall(g1.replay, 20)
        net.two_streams = False
        t_e1 = wall(lambda: net(images, geom), 20)
        print('%s %dx%d: eager two streams %.3f ms | eager one stream %.3f ms | hipGraph replay '
              '%.3f ms -> %.1f 6-cam samples/s' % (enc, size[0], size[1], t_e, t_e1, t_g,
                                                   1e3 / t_g), flush=True)


if __name__ == '__main__':
    main()

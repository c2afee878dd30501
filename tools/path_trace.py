#!/usr/bin/env python
"""N forwards of the whole occupancy path (VeonOccupancyPath, one stream) and
nothing else -- meant to run under `rocprofv3 --kernel-trace --stats`, so that
(sum of kernel durations) / N is the GPU-busy time of one step and can be held
against the wall time printed here.  Not a test."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.models.veon_occ import VeonOccupancyPath  # noqa: E402


def main():
    n = int(os.environ.get('N', 20))
    dev, size = 'cuda:0', (256, 704)
    torch.manual_seed(0)
    temporal = os.environ.get('TEMPORAL') == '1'
    kw = (dict(VeonOccupancyPath.VEON_L) if os.environ.get('ENC') == 'vitl'
          else dict(encoder='vitb'))
    net = VeonOccupancyPath(input_size=size, num_temporal=2 if temporal else 1,
                            **kw).to(dev).eval()
    net.two_streams = os.environ.get('TWO_STREAMS', '0') == '1'
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    images = torch.randn(1, 6, 3, *size, device=dev)
    if temporal:   # streaming form: kept past volume, warp + fusion every step
        eye = torch.eye(4, device=dev)[None, None]
        move = eye.clone()
        move[0, 0, :3, 3] = torch.tensor([1.5, 0.2, 0.01], device=dev)
        with torch.no_grad():
            kept = net.lift_frame(torch.randn_like(images), geom)
        plain = net.forward
        net.forward = lambda im, gm: plain(im, gm, [net.align(kept, [eye, move])])
    with torch.no_grad():
        for _ in range(3):
            net(images, geom)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(images, geom)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        if os.environ.get('CAPTURE') == '1':
            # Stream capture records the launches without running them: the time to
            # capture one forward is the pure host cost of issuing it (the graph is
            # dropped, never replayed).
            net.two_streams = False
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                net(images, geom)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            t1 = time.perf_counter()
            with torch.cuda.graph(g):
                net(images, geom)
            t_cap = time.perf_counter() - t1
            del g
            print('host cost of issuing one forward (stream capture, nothing executes): '
                  '%.2f ms' % (t_cap * 1e3))
    print('%d forwards (+3 warm-up): wall %.2f ms per step, host issue time %.2f ms per step'
          % (n, t_all / n * 1e3, t_issue / n * 1e3))


if __name__ == '__main__':
    main()

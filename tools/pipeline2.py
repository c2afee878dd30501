#!/usr/bin/env python
"""Two samples in flight: two independent instances of the occupancy path (own weights
copy, own static buffers, own hipGraph) replayed alternately on two streams, against
the sequential replay of one graph.  Same work per sample, nothing skipped; a
THROUGHPUT figure (latency per sample gets worse), reported beside the sequential one.

    python tools/pipeline2.py [vitb|vitl] [steps]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.graphs import GraphedCallable  # noqa: E402
from veon_amd.models.veon_occ import VeonOccupancyPath  # noqa: E402


def build(enc, size, dev, seed):
    torch.manual_seed(seed)
    kw = dict(VeonOccupancyPath.VEON_L) if enc == 'vitl' else dict(encoder=enc)
    net = VeonOccupancyPath(input_size=size, **kw).to(dev).eval()
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    return net, geom


def run(enc='vitb', size=(256, 704), steps=40, dev='cuda:0', n_flight=2):
    """-> dict(sequential_ms, pipelined_ms per sample, outputs_equal)."""
    nets = [build(enc, size, dev, 0) for _ in range(n_flight)]
    images = torch.randn(1, 6, 3, *size, device=dev)
    with torch.no_grad():
        graphs = [GraphedCallable(lambda im, n=n, g=g: n(im, g), (images,)) for n, g in nets]
    torch.cuda.synchronize()

    def timed(fn):
        fn(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def sequential(k):
        for _ in range(k):
            graphs[0].graph.replay()

    streams = [torch.cuda.Stream() for _ in range(n_flight)]

    def pipelined(k):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i % n_flight]):
                graphs[i % n_flight].graph.replay()
        for s in streams:
            cur.wait_stream(s)

    seq = timed(sequential)
    ref = {k: v.clone() for k, v in graphs[0].static_out.items()}
    pip = timed(pipelined)
    same = all(torch.equal(graphs[j].static_out[k], ref[k])
               for j in range(n_flight) for k in ('sem_occ', 'bin_occ'))
    return {'sequential_ms': seq, 'pipelined_ms': pip, 'in_flight': n_flight,
            'outputs_equal': bool(same)}


if __name__ == '__main__':
    enc = sys.argv[1] if len(sys.argv) > 1 else 'vitb'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    if '--json' in sys.argv:   # what bench.py's child process prints
        import json
        print('PIPELINE2 ' + json.dumps(run(enc, steps=steps, n_flight=2)), flush=True)
        sys.exit(0)
    for nf in (2, 3):
        r = run(enc, steps=steps, n_flight=nf)
        print('%s 256x704: sequential %.3f ms/sample (%.1f samples/s) | %d in flight %.3f '
              'ms/sample (%.1f samples/s) | outputs equal: %s'
              % (enc, r['sequential_ms'], 1e3 / r['sequential_ms'], nf, r['pipelined_ms'],
                 1e3 / r['pipelined_ms'], r['outputs_equal']), flush=True)

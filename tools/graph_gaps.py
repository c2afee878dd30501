#!/usr/bin/env python
"""How much of a replayed forward is launch gap?  Run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/graph_gaps.py run`,
then `python3 tools/graph_gaps.py report DIR`: for the last replays, wall time, the
union of the kernel intervals (time with at least one kernel running), the sum of
kernel durations and the number of kernels."""
import csv
import glob
import os
import sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from veon_amd import synthetic
    from veon_amd.graphs import GraphedCallable
    from veon_amd.models.veon_occ import VeonOccupancyPath
    dev, size = 'cuda:0', (256, 704)
    torch.manual_seed(0)
    net = VeonOccupancyPath(input_size=size, encoder='vitb').to(dev).eval()
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    images = torch.randn(1, 6, 3, *size, device=dev)
    with torch.no_grad():
        g = GraphedCallable(lambda im: net(im, geom), (images,))
    for _ in range(5):
        g.graph.replay()
    torch.cuda.synchronize()
    # marker: a distinctive kernel between replays
    mark = torch.zeros(977, device=dev)
    for _ in range(6):
        mark.cos_()
        g.graph.replay()
        torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'cos' in r['Kernel_Name'].lower()]
    for a, b in list(zip(marks, marks[1:]))[-4:]:
        ks = rows[a + 1:b]
        if not ks:
            continue
        iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in ks)
        t0, t1 = iv[0][0], max(e for _, e in iv)
        union, cs, ce = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > ce:
                union += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        total = sum(e - s for s, e in iv)
        print('replay: %d kernels, wall %.3f ms, some kernel running %.3f ms (idle %.3f ms = '
              '%.1f %%), sum of kernel durations %.3f ms' % (
                  len(ks), (t1 - t0) / 1e6, union / 1e6, (t1 - t0 - union) / 1e6,
                  100.0 * (t1 - t0 - union) / (t1 - t0), total / 1e6))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        report(sys.argv[2])

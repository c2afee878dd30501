#!/usr/bin/env python
"""Per-kernel table of a rocprofv3 --kernel-trace output directory:
    python tools/trace_table.py gpurun_out/<tag>/trace N_FORWARDS [top]
ms per forward, launches per forward, grouped by (shortened) kernel name."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<[^(]{0,80})?)', name)
    return (m.group(1) if m else name)[:110]


def main():
    src, n = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            tot[k][0] += 1
            tot[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    rows = sorted(tot.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for v in tot.values())
    print('total %.3f ms per forward over %.1f launches' % (total / n, sum(v[0] for v in tot.values()) / n))
    own = sum(v[1] for k, v in tot.items() if k.startswith('k_'))
    print('own kernels %.3f ms, others (torch / MIOpen / rocBLAS) %.3f ms' % (own / n, (total - own) / n))
    for k, (c, ms) in rows[:top]:
        print('%8.3f ms %7.1f x  %s' % (ms / n, c / n, k))


if __name__ == '__main__':
    main()

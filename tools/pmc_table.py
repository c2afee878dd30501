#!/usr/bin/env python
"""Per-kernel means of every counter found under a tools/pmc_run.sh output
directory, plus the kernel-trace average durations.

    python tools/pmc_table.py gpurun_out/<tag> [kernel-name filter]
"""
import collections
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, 'pmc*', '**', '*counter_collection.csv'),
                       recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if flt in k:
                vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if flt in k:
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(set(vals) | set(dur)):
        print('==', k[:150])
        if dur.get(k):
            d = sorted(dur[k])
            print('   %-28s n=%d mean %.2f us  min %.2f  med %.2f' %
                  ('duration(kernel-trace)', len(d), sum(d) / len(d), d[0], d[len(d) // 2]))
        for c in sorted(vals.get(k, {})):
            v = vals[k][c]
            print('   %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()

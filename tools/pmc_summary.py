#!/usr/bin/env python
"""Summarise the two separate rocprofv3 --pmc passes of tools/gpu_profile.sh
(FETCH_SIZE, WRITE_SIZE; KiB per launch) into profiles/<tag>_pmc_hbm_bytes.json.

    python tools/pmc_summary.py gpurun_out/<dir> profiles/r01_pmc_hbm_bytes.json
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                out[r['Kernel_Name']].append(float(r['Counter_Value']))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    res = collections.defaultdict(dict)
    for sub, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        for k, v in collect(os.path.join(src, sub), counter).items():
            if 'veon' in k or 'k_' in k:
                res[k][counter] = {'launches': len(v), 'mean_KiB': sum(v) / len(v),
                                   'min_KiB': min(v), 'max_KiB': max(v)}
    json.dump(res, open(dst, 'w'), indent=1)
    for k, v in res.items():
        if 'k_pool' in k:
            print(k[:70], {c: round(m['mean_KiB'], 1) for c, m in v.items()})


if __name__ == '__main__':
    main()

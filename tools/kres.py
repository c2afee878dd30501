#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one .hip file (compile only,
no GPU): `python tools/kres.py veon_amd/csrc/bev_pool_rows.hip [filter]`."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
           '--offload-arch=gfx950', '-I', os.path.join(ROOT, 'include'), '-c', src,
           '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r'remark:\s+([\w][^:]*?): (.+?) \[-Rpass', line)
        if not m:
            m = re.search(r'(Function Name|Name): (\S+)', line)
            if m:
                cur = {'name': m.group(2)}
                rows.append(cur)
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ('Function Name', 'Name'):
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = subprocess.run(['c++filt', r['name']],
                              capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = name.split('(')[0]
        if flt and flt not in name:
            continue
        print('%-58s vgpr %4s agpr %3s sgpr %3s scratch %4s lds %6s occ %s' % (
            name[:58], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'),
            r.get('ScratchSize [bytes/lane]', '?'), r.get('LDS Size [bytes/block]', '?'),
            r.get('Occupancy [waves/SIMD]', '?')))


if __name__ == '__main__':
    main()

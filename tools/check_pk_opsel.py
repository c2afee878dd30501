#!/usr/bin/env python
"""Build-time guard for DESIGN 4b's packed-FP32 finding: compile every HIP source to ISA
and fail if a kernel that issues MFMAs also contains a v_pk_*_f32 whose op_sel selects the
HIGH dword for the LOW half (op_sel:[..1..]) -- the form that read its operand as 0.0 in
lanes 48-63 beside the wave's own MFMAs (tools/ubench/pk_opsel_repro.hip).
    python tools/check_pk_opsel.py        (needs hipcc; no GPU)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def offenders(flags=(), only=None):
    """{(source file, kernel): count}; ``only``: restrict to these source file names."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(ROOT, 'veon_amd', 'csrc', '*.hip'))):
            if only is not None and os.path.basename(src) not in only:
                continue
            asm = os.path.join(tmp, os.path.basename(src) + '.s')
            subprocess.check_call(
                ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                 '-I', os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', asm, src]
                + list(flags), stderr=subprocess.DEVNULL)
            cur, mfma, bad = None, set(), {}
            for line in open(asm):
                m = re.match(r'^(_Z\w+):', line)
                if m:
                    cur = m.group(1)
                if 'v_mfma' in line:
                    mfma.add(cur)
                if 'v_pk_' in line and '_f32' in line:
                    sel = re.search(r'op_sel:\[([01,]+)\]', line)
                    if sel and '1' in sel.group(1):
                        bad[cur] = bad.get(cur, 0) + 1
            for k, n in bad.items():
                if k in mfma:
                    out[(os.path.basename(src), k)] = n
    return out


if __name__ == '__main__':
    found = {}
    for flags in ((), ('-DVEON_HALF_FP16',)):
        found.update(offenders(flags))
    for (src, kern), n in found.items():
        print('%s: %s has %d packed f32 ops with a low-half op_sel beside MFMAs' % (src, kern, n))
    print('%d offending kernels' % len(found))
    sys.exit(1 if found else 0)

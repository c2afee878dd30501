"""Build libveon_hip.so (gfx950) in-tree with hipcc.

    python -m veon_amd.build [--force]

hipcc cross-compiles without a GPU.  The library lands beside this file
(veon_amd/libveon_hip.so) so it travels with the source snapshot and shows up
as an in-tree native library in the loaded-objects audit.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libveon_hip.so')
ARCH = 'gfx950'

# -ffp-contract=off: the only fused multiply-adds are the explicit fmaf() calls,
# so the kernels' arithmetic matches the C oracle operation for operation.
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '--offload-arch=' + ARCH, '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(INCLUDE, '*.h'))


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in _deps())


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand)
                     or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    cmd = [hipcc()] + FLAGS + ['-I', INCLUDE, '-o', LIB] + sources()
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

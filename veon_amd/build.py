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
# the same sources with -DVEON_HALF_FP16: IEEE fp16 operands on the MFMA kernels
# instead of bf16 (csrc/mfma_common.h); selected by veon_amd.half.set_half_dtype
LIB_F16 = os.path.join(HERE, 'libveon_hip_f16.so')
FLAVOURS = {'bf16': (LIB, '_build', []), 'fp16': (LIB_F16, '_build_f16', ['-DVEON_HALF_FP16'])}
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


def up_to_date(flavour=None):
    for fl in ([flavour] if flavour else list(FLAVOURS)):
        lib = FLAVOURS[fl][0]
        if not os.path.exists(lib):
            return False
        t = os.path.getmtime(lib)
        if not all(os.path.getmtime(d) <= t for d in _deps()):
            return False
    return True


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand)
                     or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _obj(src, flavour='bf16'):
    return os.path.join(HERE, FLAVOURS[flavour][1], os.path.basename(src)[:-4] + '.o')


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(INCLUDE, '*.h'))
    return any(os.path.getmtime(d) > t for d in [src] + headers)


def build(force=False, verbose=False, jobs=None):
    """One object per .hip and flavour (compiled in parallel, only the stale ones),
    then one link per flavour: an edit of one kernel file rebuilds in seconds.
    Returns the path of the default (bf16) library."""
    if not force and up_to_date():
        return LIB
    cflags = [f for f in FLAGS if f != '-shared']
    todo = []
    for fl, (lib, objdir, defs) in FLAVOURS.items():
        os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
        todo += [(s, fl) for s in sources() if force or _stale(_obj(s, fl), s)]
    jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2) // 2)) or 1
    procs, failed = [], False
    pending = list(todo)
    while pending or procs:
        while pending and len(procs) < jobs:
            src, fl = pending.pop(0)
            cmd = [hipcc()] + cflags + FLAVOURS[fl][2] + ['-I', INCLUDE, '-c', src, '-o',
                                                          _obj(src, fl)]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
        src, pr = procs.pop(0)
        if pr.wait() != 0:
            failed = True
            print('hipcc failed on', src, file=sys.stderr)
    if failed:
        raise subprocess.CalledProcessError(1, 'hipcc')
    for fl, (lib, objdir, defs) in FLAVOURS.items():
        cmd = [hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', lib] + \
            [_obj(s, fl) for s in sources()]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

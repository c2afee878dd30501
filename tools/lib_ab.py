#!/usr/bin/env python
"""Same-box A/B of several BUILDS of libveon_hip.so (boxes differ by +-3 %, more than
some kernel changes are worth): every library is timed in its own child process,
rounds interleaved, on the Conv3d body and two encoder GEMMs.

    python tools/lib_ab.py [rounds] name=path.so name=path.so ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    sys.path.insert(0, ROOT)
    from veon_amd import conv3d_ops, vit_ops
    from veon_amd.models.semantic_net import AlignBody3D
    from tools.vit_bench import timeit
    dev = 'cuda:0'
    torch.manual_seed(0)
    C, Z, Y, X = 256, 8, 100, 100
    x = torch.randn(1, C, Z, Y, X, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * (27 * C) ** -0.5
    vol, wp = conv3d_ops.pack(x), conv3d_ops.pack_weight(w)
    out, sc = vol.like(), torch.ones(C, device=dev)
    f = lambda: conv3d_ops.conv3d_k3(vol, wp, sc, sc, relu=True, out=out)  # noqa: E731
    timeit(f, iters=30)
    conv = min(timeit(f, iters=40) for _ in range(3))
    body = AlignBody3D(C, 4).to(dev).eval()
    with torch.no_grad():
        g = lambda: body(x)  # noqa: E731  (pack + 8 convs + unpack, as tools/body_bench.py)
        timeit(g, iters=5)
        bt = min(timeit(g, iters=10) for _ in range(3))
    res = []
    for M, N, K, epi in ((5406, 3072, 768, 1), (5406, 2304, 768, 0), (5406, 768, 3072, 0)):
        a = torch.randn(M, K, device=dev).bfloat16()
        ww = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev)
        h = lambda: vit_ops.linear(a, ww, bias, epi, out=o)  # noqa: E731
        timeit(h, iters=20)
        res.append(min(timeit(h, iters=40) for _ in range(3)))
    print('conv %.1f us | body %.1f us | fc1 %.1f qkv %.1f fc2 %.1f us'
          % (conv, bt, res[0], res[1], res[2]), flush=True)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    libs = [a.split('=', 1) for a in sys.argv[1:] if '=' in a]
    for r in range(rounds):
        for name, path in libs:
            env = dict(os.environ, VEON_HIP_LIB=os.path.join(ROOT, path), LIB_AB_CHILD='1')
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith('conv ')]
            print('round %d  %-10s %s' % (r, name, line[0] if line else out.stdout[-300:]),
                  flush=True)


if __name__ == '__main__':
    if os.environ.get('LIB_AB_CHILD') == '1':
        child()
    else:
        main()

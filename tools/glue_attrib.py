#!/usr/bin/env python
"""Which modules launch the kernels that are NOT ours?  One-stream forwards of
VeonOccupancyPath under torch.profiler: every sub-module (to DEPTH levels) runs inside
a record_function range, every device kernel launched through an aten op is charged to
the innermost such range.  Prints us / forward per (module, aten op), largest first.
    python tools/glue_attrib.py [vitb|vitl] [n_forwards]
Not a test."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veon_amd import synthetic  # noqa: E402
from veon_amd.models.veon_occ import VeonOccupancyPath  # noqa: E402


def main():
    enc = sys.argv[1] if len(sys.argv) > 1 else 'vitb'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev, size = 'cuda:0', (256, 704)
    torch.manual_seed(0)
    kw = dict(VeonOccupancyPath.VEON_L) if enc == 'vitl' else dict(encoder='vitb')
    net = VeonOccupancyPath(input_size=size, two_streams=False, **kw).to(dev).eval()
    geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
    images = torch.randn(1, 6, 3, *size, device=dev)
    depth = int(os.environ.get('DEPTH', 3))

    def wrap(mod, label):
        inner = mod.forward

        def fwd(*a, **k):
            with record_function(label):
                return inner(*a, **k)
        mod.forward = fwd
    for name, mod in net.named_modules():
        if name and name.count('.') < depth:
            wrap(mod, 'M:' + name)

    def wrap_method(obj, attr, label):
        inner = getattr(obj, attr)

        def fn(*a, **k):
            with record_function(label):
                return inner(*a, **k)
        setattr(obj, attr, fn)
    # stages that are plain methods, not sub-module calls
    for obj, attr in ((net, 'estimate_depth'), (net, 'clip_features'), (net, '_tail'),
                      (net, '_classify'), (net.occ_decoder, 'prepare_depth'),
                      (net.occ_decoder, 'prepare_meta'), (net.occ_decoder, 'fuse'),
                      (net.depth_model.pretrained, 'intermediate_rows'),
                      (net.depth_model.depth_head, 'hip_forward'),
                      (net.clip_rec_head, 'update_remaining_clip_feats'),
                      (net.view_transformer, '_lift_maxpool')):
        if hasattr(obj, attr):
            wrap_method(obj, attr, 'M:%s()' % attr)
    with torch.no_grad():
        for _ in range(3):
            net(images, geom)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(n):
                net(images, geom)
            torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0.0, 0])
    kern = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        ks = getattr(ev, 'kernels', None)
        if not ks:
            continue
        where, par = '(path)', ev.cpu_parent
        while par is not None:
            if par.name.startswith('M:'):
                where = par.name[2:]
                break
            par = par.cpu_parent
        for k in ks:
            if '::k_' in k.name or k.name.startswith('k_'):
                continue   # ours
            op = ev.name if ev.name.startswith('aten::') else k.name[:44]
            acc[(where, op)][0] += k.duration
            acc[(where, op)][1] += 1
            kern[k.name[:70]][0] += k.duration
            kern[k.name[:70]][1] += 1
    total = sum(v[0] for v in acc.values())
    print('device kernels launched through aten ops: %.1f us per forward' % (total / n))
    for (where, op), (us, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:70]:
        print('%8.1f us %5.1f x  %-44s %s' % (us / n, c / n, op, where))
    print('--- by kernel')
    for name, (us, c) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:40]:
        print('%8.1f us %5.1f x  %s' % (us / n, c / n, name))


if __name__ == '__main__':
    main()

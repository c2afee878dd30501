#!/usr/bin/env python
"""bench.py -- 6-camera lift (LSS view transform + bev_pool_v2) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic 6-camera sample:
``view_transformer.view_transform(input, depth, tran_feat)`` -- the region the
reference's own benchmark times (tools/analysis_tools/benchmark_view_transformer.py:
120-138), with pre-computed ranks (``accelerate=True``, that tool's default).
Workload = BASELINE.json configs[1] (S2): 6 cams, 256x704, D=59, C=80, 200x200x16
voxels.  Inputs are resident in HBM before the timed region.  N>1: one process
per GPU (torchrun), independent samples per rank, no data-path collective
("weak" scaling); the wall time is the max over ranks.

`--gpus N` with N > 1 and no launcher in the environment (WORLD_SIZE unset) starts
the N ranks itself (`python -m torch.distributed.run --nproc-per-node N ...` as a
child process, before anything touches the GPU) and exits with the child's code; a
launcher whose world size differs from --gpus is an error.

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel of the step
(k_pool_fused_cf, HBM bound): algorithmic bytes / mean launch duration.  The
duration in `frac` is the `rocprofv3 --kernel-trace --stats` average of a child
process of THIS run (tools/pool_case.py ROOFLINE: the kernel round-robin into 8
output allocations held at once, so the placement-dependent speed of one
allocation -- DESIGN 4 -- averages out; `roofline.placement` says how many of the
8 were in the fast mode); the stats CSV of that child is kept under
gpurun_out/bench_evidence/ (copy it to profiles/ to commit it).  The same launch
timed by HIP events on the launch stream in this process is the secondary key
`kernel_ms_hip_events`.  `roofline.traffic` comes from two rocprofv3 --pmc passes
run as child processes of this script (FETCH_SIZE, WRITE_SIZE; N=1 only).  At N=1
the line also carries
  `sv`    the VEON-shaped lift (6 cams 512x1408, D=88, C=256): the fused forward
          (691.7 MB algorithmic) and the fused pool + 2x2x2 max-pool (118 MB fp32 /
          the Conv3d body's bf16 input), each with its own roofline numbers;
  `veonb` BASELINE configs[2]: the whole 3-D occupancy path (DA-V2 ViT-B + CLIP
          ViT-B/16 + HSA + lift + Conv3d body + heads), ms per 6-camera sample and
          the MFMA roofline of its conv body;
  `veonl_fp16` BASELINE configs[4] on one GPU: the VEON-L path with fp16 operands
          (libveon_hip_f16.so), whole forward from one hipGraph;
  `cpu_baseline` the pure-PyTorch index_add_ port of the step on the host cores,
          at the best of several thread counts.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, same guide

WORKLOADS = {
    # tag: (grid key, input_size, n_cams, C, neck type)
    'S2': ('GRID_S2', (256, 704), 6, 80, 'LSSViewTransformer'),
    'SV': ('GRID_VEON', (512, 1408), 6, 256, 'LSSViewTransformerRaw'),
    'S1': ('GRID_BEVDET', (256, 704), 1, 64, 'LSSViewTransformer'),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=2000)
    p.add_argument('--warmup', type=int, default=50)
    p.add_argument('--workload', default='S2', choices=sorted(WORKLOADS) + ['VEONB', 'VEONL'],
                   help='S2 (default, BASELINE configs[1]) / SV / S1: the lift; VEONB / '
                        'VEONL: the chained occupancy path of BASELINE configs[2] / [3]')
    p.add_argument('--no-graph', action='store_true',
                   help='eager launches instead of a captured hipGraph')
    p.add_argument('--shard', default='replicas', choices=['replicas', 'cameras'],
                   help='replicas: one sample per GPU, no collective (default); '
                        'cameras: the six cameras of ONE sample split over the '
                        'GPUs + RCCL all-reduce of the voxel volume')
    p.add_argument('--reduce', default='allreduce', choices=['allreduce', 'scatter'],
                   help='--shard cameras with VEONB / VEONL: one all-reduce of the un-pooled '
                        'volume, or reduce-scatter over channel slices + sharded max-pool + '
                        'all-gather of the pooled slices (1.78x fewer link bytes)')
    p.add_argument('--placement', action='store_true',
                   help='also report the kernel into a placement-tuned persistent '
                        'output volume (veon_amd/placement.py) as roofline.placed')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=10.0)
    p.add_argument('--no-sv', action='store_true', help='skip the `sv` sub-object')
    p.add_argument('--no-veonb', action='store_true', help='skip the `veonb` sub-object')
    p.add_argument('--no-veonl', action='store_true', help='skip the `veonl_fp16` sub-object')
    p.add_argument('--half', default=os.environ.get('VEON_HALF', 'bf16'),
                   choices=['bf16', 'fp16'],
                   help='16-bit operand type of the MFMA path for --workload VEONB / VEONL '
                        '(veon_amd/half.py: libveon_hip.so / libveon_hip_f16.so); BASELINE '
                        'configs[2] names bf16, configs[4] fp16')
    p.add_argument('--no-pmc', action='store_true',
                   help='skip the rocprofv3 --pmc child passes (roofline.traffic = null)')
    p.add_argument('--no-rocprof', action='store_true',
                   help='skip the rocprofv3 --kernel-trace child (roofline from HIP events)')
    p.add_argument('--evidence-dir', default=os.path.join(ROOT, 'gpurun_out', 'bench_evidence'),
                   help='where the rocprofv3 kernel stats of the roofline child are kept')
    p.add_argument('--dry-run', action='store_true',
                   help='launch / rendezvous / timing protocol only, on the CPU with gloo '
                        'and a sleep as the step: what the CPU tests drive (no GPU needed)')
    return p.parse_args()


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks (one process per GPU) as a
    child `torch.distributed.run` and exit with its code.  Runs before anything
    initialises the GPU in this process (importing torch does not), so no GPU
    process is ever replaced or forked (tools/dist_test.sh:11-22 is the reference's
    launcher of the same shape)."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args, rank, world):
    """The N-rank protocol without a GPU: gloo rendezvous, W warm-up + K timed steps
    of a 1 ms sleep between barriers, MAX over ranks, one JSON line from rank 0."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit('world size %d != --gpus %d' % (dist.get_world_size(), args.gpus))
    elapsed = timed_steps(lambda: time.sleep(1e-3), args.steps, args.warmup, dist, 'cpu',
                          rehearse=True, gpu=False)
    if rank == 0:
        print(json.dumps({
            'metric': '6cam_lift_samples_per_sec', 'value': None, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dry_run': True,
            'config': {'workload': 'dry run: 1 ms sleep per step, no GPU work',
                       'parallelism': 'replicas x%d' % world}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def solo_rank(world, rank):
    return world == 1 and rank == 0


def algorithmic_bytes(n_cams, hf, wf, C, D, p_kept, n_intervals, n_vox, batch=1,
                      feat_bytes=4, out_bytes=4, out_div=1):
    """feat once + depth once + 3 rank arrays + 2 interval arrays
    + the output volume written once (zeros included)."""
    return (batch * n_cams * hf * wf * C * feat_bytes + 4 * batch * n_cams * D * hf * wf +
            4 * (3 * p_kept + 2 * n_intervals) + batch * n_vox * C * out_bytes // out_div)


def event_ms(fn, iters, warm=5):
    """Mean time of `fn` over `iters` back-to-back calls, HIP events on the stream
    the kernels are launched on (torch's current stream)."""
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def timed_steps(run, steps, warmup, dist, dev, rehearse=False, gpu=True):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both
    sides; max over ranks."""
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    for _ in range(warmup):
        run()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cpu' if (rehearse or dist.get_backend() == 'gloo') else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


# ---------------------------------------------------------------------------
# PMC traffic: rocprofv3 child processes (counters cannot be read in-process)
# ---------------------------------------------------------------------------
def pmc_traffic():
    """{kernel-name substring: (FETCH_SIZE KiB, WRITE_SIZE KiB) mean per launch}
    from two separate `rocprofv3 --pmc` passes of tools/pool_case.py ALL, run as
    children of this process; {} when rocprofv3 is unavailable or fails."""
    exe = shutil.which('rocprofv3')
    if exe is None:
        return {}, 'rocprofv3 not on PATH'
    out = {}
    tmp = tempfile.mkdtemp(prefix='veon_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '--',
                   sys.executable, os.path.join(ROOT, 'tools', 'pool_case.py'), 'ALL']
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=240)
            if r.returncode != 0:
                return {}, 'rocprofv3 --pmc %s exited %d' % (counter, r.returncode)
            vals = {}
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'),
                               recursive=True):
                for row in csv.DictReader(open(f)):
                    if row['Counter_Name'] == counter:
                        vals.setdefault(row['Kernel_Name'], []).append(
                            float(row['Counter_Value']))
            for k, v in vals.items():
                out.setdefault(k, {})[counter] = sum(v) / len(v)
    except (subprocess.TimeoutExpired, OSError) as e:
        return {}, 'rocprofv3 child failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, 'two rocprofv3 --pmc child passes of tools/pool_case.py in this run'


def kernel_trace_stats(evidence_dir):
    """`rocprofv3 --kernel-trace --stats -- python tools/pool_case.py ROOFLINE` as a
    child of this run -> ({kernel-name: {'avg_ns', 'calls', 'per_buffer_us'}}, note).
    The S2 kernel is launched round-robin into ROOFLINE_BUFFERS allocations, so launch
    i (in start order) wrote buffer i % ROOFLINE_BUFFERS: `per_buffer_us` is the median
    per allocation.  The child's *_kernel_stats.csv is copied to `evidence_dir`."""
    exe = shutil.which('rocprofv3')
    if exe is None:
        return {}, 'rocprofv3 not on PATH'
    from tools.pool_case import ROOFLINE_BUFFERS
    tmp = tempfile.mkdtemp(prefix='veon_kt_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    out = {}
    try:
        cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '--',
               sys.executable, os.path.join(ROOT, 'tools', 'pool_case.py'), 'ROOFLINE']
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=300)
        if r.returncode != 0:
            return {}, 'rocprofv3 --kernel-trace exited %d' % r.returncode
        stats = glob.glob(os.path.join(tmp, '**', '*kernel_stats.csv'), recursive=True)
        trace = glob.glob(os.path.join(tmp, '**', '*kernel_trace.csv'), recursive=True)
        if not stats or not trace:
            return {}, 'rocprofv3 wrote no kernel stats'
        for row in csv.DictReader(open(stats[0])):
            out[row['Name']] = {'avg_ns': float(row['AverageNs']), 'calls': int(row['Calls'])}
        per = {}
        for row in csv.DictReader(open(trace[0])):
            per.setdefault(row['Kernel_Name'], []).append(
                (int(row['Start_Timestamp']), int(row['End_Timestamp'])))
        for name, spans in per.items():
            if 'k_pool_fused_cf' not in name or name not in out:
                continue
            spans.sort()
            med = []
            for b in range(ROOFLINE_BUFFERS):
                d = sorted(e - st for st, e in spans[b::ROOFLINE_BUFFERS])
                med.append(round(d[len(d) // 2] / 1e3, 2))
            out[name]['per_buffer_us'] = med
        try:
            os.makedirs(evidence_dir, exist_ok=True)
            shutil.copy(stats[0], os.path.join(evidence_dir, 'roofline_kernel_stats.csv'))
        except OSError:
            pass
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return {}, 'rocprofv3 child failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, ('rocprofv3 --kernel-trace --stats child of this run '
                 '(python tools/pool_case.py ROOFLINE)')


def rocprof_of(kt, needle):
    """(average ms, calls, entry) of the first traced kernel whose name contains
    `needle`, or (None, 0, None)."""
    for k, c in kt.items():
        if needle in k:
            return c['avg_ns'] * 1e-6, c['calls'], c
    return None, 0, None


# the two placement modes of the S2 kernel: ~34 us and ~41 us
PLACEMENT_THRESHOLD_US = 37.5


def traffic_of(pmc, needle):
    """WRITE_SIZE + 2 x FETCH_SIZE in bytes for the first kernel whose name
    contains `needle` (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads by
    2x on gfx950, WRITE_SIZE is exact; for gathers the x2 is an upper bound)."""
    for k, c in pmc.items():
        if needle in k and 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            return int((c['WRITE_SIZE'] + 2.0 * c['FETCH_SIZE']) * 1024.0)
    return None


# ---------------------------------------------------------------------------
# sub-objects
# ---------------------------------------------------------------------------
def sv_subobject(dev, pmc, kt):
    """The VEON-shaped lift (SV): fused forward and fused pool + max-pool kernels,
    each with its own algorithmic bytes (never mixed)."""
    from tools._inputs import lift_case
    from veon_amd import conv3d_ops, synthetic
    from veon_amd.ops.bev_pool_v2 import bev_pool as bp
    grid, size, cams, C = synthetic.GRID_VEON, (512, 1408), 6, 256
    cs = lift_case(grid, size, cams, C, str(dev))
    depth, feat = cs['depth'], cs['feat_nhwc']
    rb, rd, rf, st = cs['rb'], cs['rd'], cs['rf'], cs['st']
    X, Y, Z = cs['gsize']
    D, hf, wf = cs['D'], size[0] // 16, size[1] // 16
    vpb = Z * Y * X
    shape = (1, Z, Y, X, C)
    vs = bp.build_voxel_table(rb, st, 1, vpb, attach=False)
    fb = feat.bfloat16()
    out = torch.empty((1, C, Z, Y, X), dtype=torch.float32, device=dev)
    vol = conv3d_ops.PaddedVolume(1, C, Z // 2, Y // 2, X // 2, dev)
    p, i = rb.numel(), st.numel()
    cases = {
        'fused': ('k_rows_fused_cf', 'bev_pool_v2 -> (B,C,Z,Y,X) fp32',
                  algorithmic_bytes(cams, hf, wf, C, D, p, i, vpb),
                  lambda: bp.rows_forward(depth, feat, rd, rf, vs, shape, out=out)),
        'maxpool': ('k_rows_maxpool<0', 'bev_pool_v2 + 2x2x2 max-pool -> (B,C,Z/2,Y/2,X/2) fp32',
                    algorithmic_bytes(cams, hf, wf, C, D, p, i, vpb, out_div=8),
                    lambda: bp.rows_maxpool(depth, feat, rd, rf, vs, shape, (2, 2, 2))),
        'maxpool_bf16': ('k_rows_maxpool<2', 'the same from bf16 rows into the Conv3d body\'s '
                         'padded bf16 input (what the VEON path runs)',
                         algorithmic_bytes(cams, hf, wf, C, D, p, i, vpb, feat_bytes=2,
                                           out_bytes=2, out_div=8),
                         lambda: bp.rows_maxpool(depth, fb, rd, rf, vs, shape, (2, 2, 2),
                                                 out_volume=vol)),
    }
    res = {'workload': 'SV: 6-cam 512x1408, D=%d, C=%d, %dx%dx%d voxels (configs/veon), '
                       'cached ranks' % (D, C, X, Y, Z),
           'points_kept': p, 'intervals': i}
    for key, (kname, what, alg, fn) in cases.items():
        ev_ms = event_ms(fn, 50)
        rp_ms, calls, _ = rocprof_of(kt, kname)
        ms = rp_ms if rp_ms is not None else ev_ms
        gbs = alg / (ms * 1e-3) / 1e9
        res[key] = {'kernel': kname.split('<')[0], 'what': what, 'bound': 'hbm',
                    'algorithmic_bytes': alg, 'kernel_ms': round(ms, 5),
                    'kernel_ms_source': 'rocprofv3' if rp_ms is not None else 'hip_events',
                    'kernel_ms_rocprof': None if rp_ms is None else round(rp_ms, 5),
                    'rocprof_calls': calls, 'kernel_ms_hip_events': round(ev_ms, 5),
                    'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 4),
                    'traffic': traffic_of(pmc, kname)}
    return res


def veon_path(args, dev, encoder, size, steps, warmup, dist, world):
    """One chained occupancy-path workload -> (ms per step, stage dict, body
    TFLOP/s, launch mode).  The timed step is the whole forward replayed from ONE
    hipGraph (BASELINE configs[4]: "hipGraph-captured forward") unless capture
    fails or --no-graph is given."""
    from tools import hotpath_bench
    from veon_amd.graphs import GraphedCallable
    r = hotpath_bench.run_full(encoder, size, dev=str(dev), iters=5, verbose=False)
    step, launch = r['step'], 'eager, encoder branches on two streams'
    if not args.no_graph:
        try:
            net, images, geom = r['net'], r['images'], r['geom']
            graphed = GraphedCallable(lambda im: net(im, geom), (images,))
            step = lambda: graphed.graph.replay()  # noqa: E731  (static input in place)
            launch = 'one hipGraph of the whole forward'
        except Exception as e:  # report, do not hide
            print('whole-forward capture failed (%r); eager' % (e,), file=sys.stderr)
    with torch.no_grad():
        el = timed_steps(step, steps, warmup, dist, dev)
    ms = el / steps * 1e3
    body_flops = 8 * 2.0 * 8 * 100 * 100 * 256 * 256 * 27
    tf = body_flops / (r['body_ms'] * 1e-3) / 1e12
    stages = {k: round(v, 3) for k, v in r.items() if k.endswith('_ms')}
    return ms, stages, tf, launch


VEON_WHAT = ('the 3-D occupancy path of VeonTemporal.simple_test '
             '(veon_amd/models/veon_occ.py): DA-V2 %s + DPT head -> depth; CLIP %s first '
             'blocks -> HSA network -> CLIP tail with attention biases; CatFusionLift -> '
             'two-hot lift by construction (depth map -> per-pixel windows, eps 1e-6; D=88, '
             'C=256, 200x200x16, sync-free prepare, fused 2x2x2 max-pool) -> 4x '
             'ResBlock3D -> occ/sem heads -> open-vocab classifier -> upsample -> arg-max; '
             '%s operands on MFMA (fp32 accumulation), random weights; timm side-adapter ViT '
             '/ mask decoder / text encoder not included')


def bench_hotpath(args, rank, world, dev, dist):
    """--workload VEONB / VEONL: the whole path as the timed step.  --shard replicas
    (default): one sample per GPU, no collective (BASELINE configs[4]); --shard
    cameras: the six cameras of ONE sample over the ranks, encoders per shard, one
    RCCL all-reduce of the voxel feature volume (configs[3])."""
    from veon_amd import half
    half.set_half_dtype(args.half)   # before any module packs its weights
    enc = 'vitb' if args.workload == 'VEONB' else 'vitl'
    clip = 'ViT-B/16' if enc == 'vitb' else 'ViT-L/14-336'
    if args.shard == 'cameras':
        from tools import hotpath_bench
        r = hotpath_bench.run_full(enc, (256, 704), dev=str(dev), iters=3, verbose=False)
        net, images, geom = r['net'], r['images'], r['geom']
        rd = torch.bfloat16 if os.environ.get('VEON_REDUCE_DTYPE', 'bf16') == 'bf16' else None

        launch = 'eager'
        step = None
        if not args.no_graph:
            try:
                from veon_amd.models.veon_occ import CameraShardedStep
                with torch.no_grad():
                    sharded = CameraShardedStep(net, images, geom, reduce_dtype=rd,
                                                reduce=args.reduce)
                step = lambda: sharded(images)  # noqa: E731
                launch = sharded.launch
            except Exception as e:  # report, do not hide
                print('sharded graph segments failed (%r); eager' % (e,), file=sys.stderr)
        if step is None:
            def step():
                return net.forward_camera_sharded(images, geom, reduce_dtype=rd,
                                                  reduce=args.reduce)
        with torch.no_grad():
            el = timed_steps(step, args.steps, args.warmup, dist, dev)
        ms = el / args.steps * 1e3
        stages = {k: round(v, 3) for k, v in r.items() if k.endswith('_ms')}
        tf = 8 * 2.0 * 8 * 100 * 100 * 256 * 256 * 27 / (r['body_ms'] * 1e-3) / 1e12
        launch = '%s; %s of the un-pooled volume in %s' % (
            launch, 'all-reduce' if args.reduce == 'allreduce' else
            'reduce-scatter (channel slices) + sharded max-pool + all-gather',
            'bf16' if rd is not None else 'fp32')
        value, scaling = 1e3 / ms, 'strong'
        par = 'cameras of one sample sharded over %d GPUs + RCCL all-reduce' % world
    else:
        ms, stages, tf, launch = veon_path(args, dev, enc, (256, 704), args.steps,
                                           args.warmup, dist, world)
        value, scaling, par = world * 1e3 / ms, 'weak', 'replicas x%d' % world
    if rank != 0:
        return
    print(json.dumps({
        'metric': '6cam_hotpath_samples_per_sec', 'value': round(value, 2),
        'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': scaling,
        'vs_baseline': None, 'dtype': half.name(), 'data': 'synthetic',
        'config': {'workload': '%s, 6-cam 256x704: ' % args.workload +
                               VEON_WHAT % ('ViT-B' if enc == 'vitb' else 'ViT-L', clip,
                                            half.name()),
                   'parallelism': par, 'launch': launch, 'stages_ms': stages},
        'roofline': {'kernel': 'k_conv3d_k3 (8 launches, AlignNetOcc3D body)', 'bound': 'mfma',
                     'achieved': round(tf, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(tf / MFMA_PEAK_TFLOPS, 4), 'traffic': None},
        'cpu_baseline': None}), flush=True)


def main():
    args = parse()
    launch_ranks(args)   # --gpus N > 1 without a launcher: never returns
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('bench.py: launcher world size %d != --gpus %d' % (world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (no CPU fallback)')
    # rehearsal knobs for a ONE-GPU box (never set by the driver): all ranks on
    # device 0 and the gloo backend, to exercise the N>1 control flow
    rehearse = os.environ.get('VEON_BENCH_REHEARSAL') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit('bench.py: process group of %d ranks != --gpus %d'
                             % (dist.get_world_size(), args.gpus))

    from veon_amd import _lib, synthetic
    from veon_amd.models import build_neck
    from veon_amd.ops.bev_pool_v2 import bev_pool as bp
    _lib.lib()  # fail loudly if the HIP library is missing

    if args.workload in ('VEONB', 'VEONL'):
        bench_hotpath(args, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    grid_key, input_size, n_cams, C, neck_type = WORKLOADS[args.workload]
    grid = getattr(synthetic, grid_key)
    cfg = dict(type=neck_type, grid_config=grid, input_size=input_size,
               downsample=16, out_channels=C, accelerate=True, collapse_z=False)
    if neck_type == 'LSSViewTransformer':
        cfg['in_channels'] = 8  # depth_net is not on the timed path
    else:
        cfg['ds_feat'] = [1, 1, 1]
    vt = build_neck(cfg).to(dev).eval()
    hf, wf = input_size[0] // 16, input_size[1] // 16
    D = vt.D
    rig = synthetic.make_rig(1, n_cams, input_size)
    geom = [t.to(dev) for t in synthetic.rig_inputs(rig)]
    depth5, feat5 = synthetic.make_depth_feat(1, n_cams, D, C, hf, wf,
                                              seed=rank, device=dev)
    depth = depth5.view(n_cams, D, hf, wf)
    tran_feat = feat5.view(n_cams, C, hf, wf)
    inp = [feat5] + geom
    core_returns_tuple = vt._core_returns_depth

    def step():
        out = vt.view_transform(inp, depth, tran_feat)
        return out[0] if core_returns_tuple else out

    if args.shard == 'cameras':
        # strong scaling of ONE sample: each rank lifts its cameras with cached
        # ranks of its own sub-rig, then all-reduce (RCCL) of the volume
        from veon_amd import sharding
        sharded = sharding.CameraShardedLift(vt)

        def step():  # noqa: F811
            return sharded(inp, depth5)

    with torch.no_grad():
        out = step()  # pre_compute + first launch
        torch.cuda.synchronize()
        Z, Y, X = (int(vt.grid_size[i]) for i in (2, 1, 0))
        # the accelerate path squeezes a unit Z (view_transformer.py:281)
        assert tuple(out.shape) in ((1, C, Z, Y, X), (1, C, Y, X)), out.shape
        p_kept = vt.ranks_bev.numel()
        n_int = vt.interval_starts.numel()

        graph = None
        if not args.no_graph:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):   # the warm-up's stream
                    step()
            except Exception as e:  # report, do not hide
                print('hipGraph capture failed (%s); running eager' % e,
                      file=sys.stderr)
                graph = None
        run = graph.replay if graph is not None else step
        launch_probe = None
        if graph is not None:
            # A replay of this two-kernel graph costs more on the GPU front end than
            # the two plain launches do on the host on some boxes (and less on
            # others): measure both briefly, run the timed steps with the faster.
            # (the probe runs as many steps as the timed region: plain launches win a
            # 40-step probe by 4 us but pay a fixed ~25 ms runtime stall somewhere
            # between 500 and 2000 steps -- 48.3 us/step at K = 500, 61 at K = 2000 --
            # which a replayed graph does not)
            def probe(fn, n=max(40, min(args.steps, 4000))):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n * 1e6
            # the first pass absorbs a one-time stall of the HIP runtime (25-60 ms,
            # somewhere in the first ~2500 plain launches of a process)
            first = probe(step, max(3000, min(args.steps, 4000)))
            launch_probe = {'hipGraph_us': round(probe(graph.replay), 2),
                            'eager_us': round(probe(step), 2),
                            'eager_first_pass_us': round(first, 2)}
            if launch_probe['eager_us'] < launch_probe['hipGraph_us']:
                run, graph = step, None

        elapsed = timed_steps(run, args.steps, args.warmup, dist, dev, rehearse)

        # ---- kernel leg: the fused pool kernel alone, HIP events on the stream it
        # is launched on, into the buffer the allocator hands out (as the step)
        feat_nhwc = feat5.permute(0, 1, 3, 4, 2).contiguous()
        shape = (1, Z, Y, X, C)
        kiters = max(args.steps, 200)

        def kernel_only(out=None):
            return bp._fused_forward(depth5, feat_nhwc, vt.ranks_depth,
                                     vt.ranks_feat, vt.ranks_bev,
                                     vt.interval_starts, vt.interval_lengths,
                                     shape, _lib.LAYOUT_BCZYX, out=out)
        # round-robin into 8 allocations held at once, like the rocprofv3 child: the
        # speed of this kernel depends on where its output volume lies
        from tools.pool_case import ROOFLINE_BUFFERS
        bufs = [torch.empty((1, C, Z, Y, X), dtype=torch.float32, device=dev)
                for _ in range(ROOFLINE_BUFFERS)] if not bp._rows_ok(C) else [None]
        turn = [0]

        def kernel_rr():
            turn[0] += 1
            return kernel_only(bufs[turn[0] % len(bufs)])
        kernel_ms = event_ms(kernel_rr, kiters, warm=10)
        del bufs
        placed = None
        if args.placement and args.shard == 'replicas':
            from veon_amd import placement
            buf, info = placement.best_placed(lambda o: kernel_only(o), (1, C, Z, Y, X),
                                              torch.float32, dev)
            placed_ms = event_ms(lambda: kernel_only(buf), kiters, warm=10)
            placed = {'kernel_ms': round(placed_ms, 5), 'candidates': info['candidates'],
                      'what': 'the same launch into the fastest of several candidate '
                              'allocations (veon_amd/placement.py); not the headline: it '
                              'depends on what the allocator can offer on the box'}

        # ---- per-call leg (what every VEON config runs: accelerate=False):
        # geometry + prepare + pool each step, no host sync, hipGraph-captured
        percall = None
        try:
            cfg2 = dict(cfg, accelerate=False)
            vt2 = build_neck(cfg2).to(dev).eval()
            vt2.sync_free = True

            def step2():
                o = vt2.view_transform(inp, depth, tran_feat)
                return o[0] if core_returns_tuple else o
            ref2 = step2()
            torch.cuda.synchronize()
            ref2 = ref2.reshape(out.shape)  # the cached-rank path squeezes a unit Z
            assert torch.equal(ref2, out) or (ref2 != out).float().mean() < 1e-3
            s2 = torch.cuda.Stream()
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                step2()
            torch.cuda.current_stream().wait_stream(s2)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=s2):
                step2()
            for _ in range(args.warmup):
                g2.replay()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                g2.replay()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t2
            percall = {'ms_per_step': round(el2 / args.steps * 1e3, 5),
                       'samples_per_s': round(args.steps / el2, 2),
                       'what': 'view_transform(accelerate=False, sync_free): geometry + '
                               'counting-sort prepare (5 launches, static workspace) + '
                               'pool per step, one hipGraph'}
        except Exception as e:  # report, do not hide
            print('per-call leg failed: %r' % (e,), file=sys.stderr)

    # ---- backward leg (row a10; training only): the op's backward on a channels-last
    # out_grad (what the reference's op receives, bev_pool.py:43-83) and the drop-in's
    # whole backward from a (B,C,Z,Y,X) out_grad (+ its layout copy)
    backward = None
    if solo_rank(world, rank) and args.workload == 'S2' and args.shard == 'replicas':
        try:
            og_cl = torch.randn((1, Z, Y, X, C), device=dev)
            og_cf = og_cl.permute(0, 4, 1, 2, 3).contiguous()
            args_b = (vt.ranks_bev, depth5, feat_nhwc, vt.ranks_feat, vt.ranks_depth)
            t_k = event_ms(lambda: bp._backward_impl(og_cl, *args_b), 50, warm=5)
            t_f = event_ms(lambda: bp._backward_impl(
                og_cf.permute(0, 2, 3, 4, 1).contiguous(), *args_b), 50, warm=5)
            alg_b = (4 * n_int * C + 4 * n_cams * hf * wf * C + 4 * n_cams * D * hf * wf +
                     4 * 3 * p_kept + 8 * n_int +
                     4 * n_cams * hf * wf * C + 4 * n_cams * D * hf * wf)
            backward = {
                'what': 'bev_pool_v2 backward at S2 (depth.grad + feat.grad): QuickCumsumCuda.'
                        'backward as the reference structures it (stable sort by ranks_feat + '
                        'run lengths in torch, then the HIP grad kernel), HIP events',
                'op_ms': round(t_k, 5), 'with_layout_copy_ms': round(t_f, 5),
                'algorithmic_bytes': alg_b,
                'achieved_GBps': round(alg_b / (t_k * 1e-3) / 1e9, 1),
                'note': 'out_grad rows of the occupied voxels + depth + feat + ranks read, '
                        'both gradients written; the (B,C,Z,Y,X) -> channels-last copy of '
                        'out_grad (205 MB read + written, as the reference does) dominates '
                        'the drop-in\'s backward'}
        except Exception as e:  # report, do not hide
            print('backward leg failed: %r' % (e,), file=sys.stderr)

    alg = algorithmic_bytes(n_cams, hf, wf, C, D, p_kept, n_int, Z * Y * X)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    value = (world if args.shard == 'replicas' else 1) * args.steps / elapsed

    solo = rank == 0 and world == 1
    pmc, pmc_src = ({}, 'not collected')
    kt, kt_src = ({}, 'not collected')
    if solo and not args.no_pmc:
        pmc, pmc_src = pmc_traffic()
    if solo and not args.no_rocprof and args.workload == 'S2':
        torch.cuda.empty_cache()
        kt, kt_src = kernel_trace_stats(args.evidence_dir)
    kname = 'k_pool_fused_cf' if not bp._rows_ok(C) else 'k_rows_fused_cf'
    rp_ms, rp_calls, rp_entry = rocprof_of(kt, kname)
    events_ms = kernel_ms
    if rp_ms is not None:   # the reproducible number: frac is computed from it
        kernel_ms = rp_ms
        achieved = alg / (kernel_ms * 1e-3) / 1e9
    placement = None
    if rp_entry is not None and rp_entry.get('per_buffer_us'):
        per = rp_entry['per_buffer_us']
        fast = sum(1 for v in per if v < PLACEMENT_THRESHOLD_US)
        placement = {'buffers': len(per), 'fast': fast, 'median_us_per_buffer': per,
                     'threshold_us': PLACEMENT_THRESHOLD_US,
                     'mode': 'fast' if fast == len(per) else 'slow' if fast == 0 else 'mixed'}
    result = {
        'metric': '6cam_lift_samples_per_sec',
        'value': round(value, 2),
        'unit': 'samples/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 5),
        'higher_is_better': True,
        'scaling': 'weak' if args.shard == 'replicas' else 'strong',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': '%s: %d-cam %dx%d, D=%d, C=%d, %dx%dx%d voxels, '
                        'view_transform(accelerate=True) -> (B,C,Z,Y,X)'
                        % (args.workload, n_cams, input_size[0], input_size[1],
                           D, C, X, Y, Z),
            'points_kept': p_kept, 'intervals': n_int,
            'launch': 'hipGraph' if graph is not None else 'eager',
            'launch_probe_us_per_step': launch_probe,
            'output_volume': 'allocator-placed (torch.empty per step)',
            'parallelism': ('replicas x%d (one sample per GPU, no collective)' % world
                            if args.shard == 'replicas' else
                            'cameras sharded over %d GPUs + all-reduce of the volume' % world),
        },
        'roofline': {
            'kernel': '%s (bev_pool_v2 fused zero-fill+pool+layout)' % kname,
            'bound': 'hbm',
            'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic_of(pmc, kname) if args.workload == 'S2' else None,
            'traffic_source': pmc_src,
            'algorithmic_bytes': alg,
            'kernel_ms': round(kernel_ms, 5),
            'kernel_ms_source': 'rocprofv3' if rp_ms is not None else 'hip_events',
            'kernel_ms_rocprof': None if rp_ms is None else round(rp_ms, 5),
            'rocprof_calls': rp_calls, 'rocprof_source': kt_src,
            'kernel_ms_hip_events': round(events_ms, 5),
            'frac_hip_events': round(alg / (events_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'kernel_launches_timed': kiters,
            'placement_mode': placement['mode'] if placement else None,
            'placement': placement,
        },
    }
    if placed is not None:
        pg = alg / (placed['kernel_ms'] * 1e-3) / 1e9
        placed.update(achieved=round(pg, 1), frac=round(pg / HBM_PEAK_GBS, 4))
        result['roofline']['placed'] = placed
    if percall is not None:
        result['percall_prepare'] = percall
    if backward is not None:
        result['backward'] = backward
    if solo and not args.no_sv and args.workload == 'S2':
        try:
            with torch.no_grad():
                result['sv'] = sv_subobject(dev, pmc, kt)
        except Exception as e:  # report, do not hide
            print('sv sub-object failed: %r' % (e,), file=sys.stderr)
    if solo and not args.no_veonb and args.workload == 'S2':
        try:
            ms, stages, tf, launch = veon_path(args, dev, 'vitb', (256, 704), 20, 3, None, 1)
            result['veonb'] = {
                'workload': 'VEONB (BASELINE configs[2]), 6-cam 256x704: ' +
                            VEON_WHAT % ('ViT-B', 'ViT-B/16', 'bf16'),
                'ms_per_step': round(ms, 4), 'samples_per_s': round(1e3 / ms, 2),
                'steps': 20, 'dtype': 'bf16', 'launch': launch, 'stages_ms': stages,
                'roofline': {'kernel': 'k_conv3d_k3 (8 launches, AlignNetOcc3D body)',
                             'bound': 'mfma', 'achieved': round(tf, 1),
                             'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': round(tf / MFMA_PEAK_TFLOPS, 4)}}
            # throughput with TWO samples in flight (two independent instances, each
            # its own hipGraph, replayed alternately on two streams): the same work per
            # sample, a serving-style figure beside the sequential one above
            # (a child process: in THIS process, with the S2 / per-call / VEON-B graphs
            # and their streams alive, the same two instances overlap far less -- 6.5 vs
            # 5.9 ms per sample -- for a reason not yet found)
            torch.cuda.empty_cache()
            cp = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pipeline2.py'),
                                 'vitb', '40', '--json'], stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT, text=True, timeout=300)
            line = [ln for ln in cp.stdout.splitlines() if ln.startswith('PIPELINE2 ')]
            if not line:
                raise RuntimeError('tools/pipeline2.py: ' + cp.stdout[-300:])
            r2 = json.loads(line[0][len('PIPELINE2 '):])
            result['veonb']['two_in_flight'] = {
                'ms_per_sample': round(r2['pipelined_ms'], 4),
                'samples_per_s': round(1e3 / r2['pipelined_ms'], 2),
                'sequential_ms_same_instances': round(r2['sequential_ms'], 4),
                'outputs_equal_to_sequential': r2['outputs_equal'],
                'measured_in': 'child process (python tools/pipeline2.py vitb 40 --json)',
                'what': 'two independent path instances (own buffers, own hipGraph) '
                        'replayed alternately on two streams; per-sample latency is '
                        'higher, throughput is what is reported'}
        except Exception as e:  # report, do not hide
            print('veonb sub-object failed: %r' % (e,), file=sys.stderr)
    if solo and not args.no_veonb and args.workload == 'S2':
        # the same forward WITH the 2-D mask branch the reference always runs beside it
        # (SAN side-adapter ViT on the full-resolution image -> mask proposals ->
        # attention biases -> CLIP recognition head -> class logits -> 2-D semantic
        # maps; san_in_veon_temporal.py:123-139, 176-186): SURVEY 8 row f3
        try:
            from veon_amd import synthetic as syn
            from veon_amd.graphs import GraphedCallable
            from veon_amd.models.veon_occ import VeonOccupancyPath
            torch.cuda.empty_cache()
            torch.manual_seed(0)
            net2 = VeonOccupancyPath(input_size=(256, 704), encoder='vitb',
                                     side_adapter=True).to(dev).eval()
            geom2 = [t.to(dev) for t in syn.rig_inputs(syn.make_rig(1, 6, (256, 704)))]
            im2 = torch.randn(1, 6, 3, 256, 704, device=dev)
            with torch.no_grad():
                launch2 = 'one hipGraph of the whole forward'
                try:
                    g2d = GraphedCallable(lambda im: net2(im, geom2, with_2d=True), (im2,))
                    step2d = g2d.graph.replay
                except Exception as e:  # report, do not hide
                    print('with-2d capture failed (%r); eager' % (e,), file=sys.stderr)
                    launch2 = 'eager'
                    step2d = lambda: net2(im2, geom2, with_2d=True)  # noqa: E731
                g3d = GraphedCallable(lambda im: net2(im, geom2), (im2,))
                ms2d = timed_steps(step2d, 20, 3, None, dev) / 20 * 1e3
                ms3d = timed_steps(g3d.graph.replay, 20, 3, None, dev) / 20 * 1e3
            result['veonb_with_2d'] = {
                'what': 'VEONB forward + the 2-D open-vocabulary mask branch on the same CLIP '
                        'features: side-adapter ViT (width 240, 6 heads of 40, 8 blocks, 100 '
                        'queries; blocks on the MFMA kernels, width padded to 256 / heads to '
                        '64) + mask decoder + recognition head + 2-D semantic inference',
                'ms_per_step': round(ms2d, 4), 'samples_per_s': round(1e3 / ms2d, 2),
                'ms_per_step_3d_only_same_instance': round(ms3d, 4),
                'branch_2d_ms': round(ms2d - ms3d, 4), 'steps': 20, 'dtype': 'bf16',
                'launch': launch2}
            del net2, g3d
            torch.cuda.empty_cache()
        except Exception as e:  # report, do not hide
            print('veonb_with_2d sub-object failed: %r' % (e,), file=sys.stderr)
    if solo and not args.no_veonl and args.workload == 'S2':
        # BASELINE configs[4] on one GPU: VEON-L, fp16 operands, hipGraph-captured
        # forward (8 x data-parallel = 8 such replicas, no collective)
        try:
            from veon_amd import half
            torch.cuda.empty_cache()
            with half.use('fp16'):
                ms, stages, tf, launch = veon_path(args, dev, 'vitl', (256, 704), 10, 2, None, 1)
            result['veonl_fp16'] = {
                'workload': 'VEONL (BASELINE configs[4], one replica), 6-cam 256x704: ' +
                            VEON_WHAT % ('ViT-L', 'ViT-L/14-336', 'fp16'),
                'ms_per_step': round(ms, 4), 'samples_per_s': round(1e3 / ms, 2),
                'steps': 10, 'dtype': 'fp16', 'launch': launch, 'stages_ms': stages,
                'roofline': {'kernel': 'k_conv3d_k3 (8 launches, AlignNetOcc3D body)',
                             'bound': 'mfma', 'achieved': round(tf, 1),
                             'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': round(tf / MFMA_PEAK_TFLOPS, 4)}}
        except Exception as e:  # report, do not hide
            print('veonl_fp16 sub-object failed: %r' % (e,), file=sys.stderr)
    if solo and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args, grid, input_size, n_cams, C,
                                              rig, depth5, feat5)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, grid, input_size, n_cams, C, rig, depth5, feat5):
    """The same step (pool with cached ranks -> (B,C,Z,Y,X)) by the pure-PyTorch
    index_add_ port on the host cores, at the best of several torch thread
    counts (a short probe each, then the sample at the winner).  The oracle is
    only the thing timed here, never part of the product path."""
    from oracle import lss_torch
    lower, interval, gsize = lss_torch.grid_infos(grid)
    fr = lss_torch.make_frustum(grid['depth'], input_size, 16)
    cams = (rig['sensor2ego'], rig['intrins'], rig['post_rots'],
            rig['post_trans'], rig['bda'])
    coor = lss_torch.lidar_coor(fr, *cams)
    ranks = lss_torch.voxel_prepare(coor, lower, interval, gsize)
    d, f = depth5.cpu(), feat5.cpu()

    def once():
        lss_torch.lift(fr, (lower, interval, gsize), cams, d, f, ranks=ranks)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    probes = {}
    for n in sorted({t for t in (8, 16, 32, 64, 128, default_threads) if t <= max(ncpu, 8)}):
        torch.set_num_threads(n)
        once()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.5:
            once()
            k += 1
        probes[n] = k / (time.perf_counter() - t0)
    best = max(probes, key=probes.get)
    torch.set_num_threads(best)
    n, t0 = 0, time.perf_counter()
    while True:
        once()
        n += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or n >= 400:
            break
    # the per-call step VEON runs (accelerate=False): geometry + prepare + pool each
    # call, same thread count, a bounded sample
    m, t1 = 0, time.perf_counter()
    while True:
        lss_torch.lift(fr, (lower, interval, gsize), cams, d, f)
        m += 1
        el2 = time.perf_counter() - t1
        if el2 > max(args.cpu_seconds * 0.5, 2.0) or m >= 200:
            break
    torch.set_num_threads(default_threads)
    return {
        'value': round(n / el, 3), 'unit': 'samples/s', 'cores': best,
        'host_cores': ncpu, 'cpu_model': cpu_model(),
        'kind': 'port',
        'sample': '%d iterations (%.1f s) of oracle.lss_torch.lift with cached '
                  'ranks on the %s workload, torch %d threads (best of %s by a 1.5 s '
                  'probe each: %s samples/s), fp32'
                  % (n, el, args.workload, best, sorted(probes),
                     {k: round(v, 2) for k, v in sorted(probes.items())}),
        'percall': {'value': round(m / el2, 3), 'unit': 'samples/s', 'cores': best,
                    'sample': '%d iterations (%.1f s) of oracle.lss_torch.lift WITHOUT cached '
                              'ranks (get_lidar_coor + voxel_pooling_prepare_v2 + pool per '
                              'call: the counterpart of percall_prepare)' % (m, el2)},
    }


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


if __name__ == '__main__':
    main()

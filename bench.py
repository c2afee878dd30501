#!/usr/bin/env python
"""bench.py -- 6-camera lift (LSS view transform + bev_pool_v2) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic 6-camera sample:
``view_transformer.view_transform(input, depth, tran_feat)`` -- the region the
reference's own benchmark times (tools/analysis_tools/benchmark_view_transformer.py:
120-138), with pre-computed ranks (``accelerate=True``, that tool's default).
Workload = BASELINE.json configs[1]: 6 cams, 256x704, D=59, C=80, 200x200x16
voxels.  Inputs are resident in HBM before the timed region.  N>1: one process
per GPU (torchrun), independent samples per rank, no data-path collective
("weak" scaling); the wall time is the max over ranks.

Rank 0 prints ONE JSON line with `roofline` (bev_pool_v2 fused kernel, HBM
bound, algorithmic bytes / mean launch time from HIP events) and `cpu_baseline`
(the pure-PyTorch index_add_ port of the same step on the host cores).
"""
import argparse
import json
import os
import sys
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
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--workload', default='S2', choices=sorted(WORKLOADS) + ['VEONB'],
                   help='S2 (default, BASELINE configs[1]) / SV / S1: the lift; VEONB: the '
                        'chained hot path of BASELINE configs[2] (tools/hotpath_bench.py)')
    p.add_argument('--no-graph', action='store_true',
                   help='eager launches instead of a captured hipGraph')
    p.add_argument('--shard', default='replicas', choices=['replicas', 'cameras'],
                   help='replicas: one sample per GPU, no collective (default); '
                        'cameras: the six cameras of ONE sample split over the '
                        'GPUs + RCCL all-reduce of the voxel volume')
    p.add_argument('--no-placement', action='store_true',
                   help='do not keep / tune a persistent output volume '
                        '(veon_amd/placement.py); the allocator places it')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=12.0)
    p.add_argument('--pmc-traffic', type=float, default=None,
                   help='HBM bytes per launch from a separate rocprofv3 --pmc '
                        'run (profiles/), copied into roofline.traffic')
    return p.parse_args()


def committed_pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r01_pmc_hbm_bytes.json: two separate `rocprofv3 --pmc` runs of this
    script, FETCH_SIZE and WRITE_SIZE in KiB).  WRITE_SIZE is exact for these
    stores; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
    (an upper bound here: the guide calibrates the x2 for wide streaming reads,
    these are gathers).  None when the file or the kernel is missing."""
    if workload != 'S2':
        return None, None
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_bytes.json')
    try:
        with open(path) as f:
            data = json.load(f)
        for name, c in data.items():
            if 'k_pool_fused_cf<4, 32' in name:
                b = (c['WRITE_SIZE']['mean_KiB'] + 2.0 * c['FETCH_SIZE']['mean_KiB']) * 1024.0
                return round(b), 'profiles/r01_pmc_hbm_bytes.json (WRITE_SIZE + 2 x FETCH_SIZE)'
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def algorithmic_bytes(n_cams, hf, wf, C, D, p_kept, n_intervals, n_vox, batch=1):
    """feat once + depth once + 3 rank arrays + 2 interval arrays
    + the output volume written once (zeros included), fp32/int32."""
    return 4 * (batch * n_cams * hf * wf * C + batch * n_cams * D * hf * wf +
                3 * p_kept + 2 * n_intervals + batch * n_vox * C)


def bench_hotpath(args, rank, world, dev, dist):
    """BASELINE configs[2] shape: DA-V2 ViT-B + CLIP ViT-B/16 + lift + Conv3d body
    + heads, bf16, one 6-camera 256x704 sample per step (replicas for N > 1).
    The roofline object is the dominant MFMA kernel (the 3x3x3 conv body)."""
    from tools import hotpath_bench
    r = hotpath_bench.run_full('vitb', (256, 704), dev=str(dev), iters=5, verbose=False)
    step = r['step']
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64,
                         device='cpu' if dist.get_backend() == 'gloo' else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return
    ms = el / args.steps * 1e3
    body_flops = 8 * 2.0 * 8 * 100 * 100 * 256 * 256 * 27
    tf = body_flops / (r['body_ms'] * 1e-3) / 1e12
    print(json.dumps({
        'metric': '6cam_hotpath_samples_per_sec', 'value': round(world * 1e3 / ms, 2),
        'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'VEONB: the 3-D occupancy path of VeonTemporal.simple_test '
                               '(veon_amd/models/veon_occ.py), 6-cam 256x704: DA-V2 ViT-B + DPT '
                               'head -> depth; CLIP ViT-B/16 first 9 blocks -> HSA network -> '
                               'CLIP tail with attention biases; CatFusionLift -> sync-free lift '
                               '(D=88, C=256, 200x200x16, fused 2x2x2 max-pool) -> 4x ResBlock3D '
                               '-> occ/sem heads -> open-vocab classifier -> upsample -> arg-max; '
                               'bf16 on MFMA, random weights; timm side-adapter ViT / mask '
                               'decoder / text encoder not included',
                   'parallelism': 'replicas x%d' % world,
                   'stages_ms': {k: round(v, 3) for k, v in r.items() if k.endswith('_ms')}},
        'roofline': {'kernel': 'k_conv3d_k3 (8 launches, AlignNetOcc3D body)', 'bound': 'mfma',
                     'achieved': round(tf, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(tf / MFMA_PEAK_TFLOPS, 4), 'traffic': None},
        'cpu_baseline': None}))


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
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

    from veon_amd import _lib, synthetic
    from veon_amd.models import build_neck
    from veon_amd.ops.bev_pool_v2 import bev_pool as bp
    _lib.lib()  # fail loudly if the HIP library is missing

    if args.workload == 'VEONB':
        return bench_hotpath(args, rank, world, dev, dist)

    grid_key, input_size, n_cams, C, neck_type = WORKLOADS[args.workload]
    grid = getattr(synthetic, grid_key)
    cfg = dict(type=neck_type, grid_config=grid, input_size=input_size,
               downsample=16, out_channels=C, accelerate=True, collapse_z=False)
    if neck_type == 'LSSViewTransformer':
        cfg['in_channels'] = 8  # depth_net is not on the timed path
    else:
        cfg['ds_feat'] = [1, 1, 1]
    vt = build_neck(cfg).to(dev).eval()
    # one output volume kept across steps (a graph replay does that anyway),
    # picked among a few allocations by timing the kernel on each
    vt.persistent_output = not args.no_placement and args.shard == 'replicas'
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
                with torch.cuda.graph(graph):
                    gout = step()
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
            def probe(fn, n=40):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n * 1e6
            launch_probe = {'hipGraph_us': round(probe(graph.replay), 2),
                            'eager_us': round(probe(step), 2)}
            if launch_probe['eager_us'] < launch_probe['hipGraph_us']:
                run, graph = step, None

        for _ in range(args.warmup):
            run()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64,
                                device='cpu' if rehearse else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())

        # ---- kernel leg: the fused pool kernel alone, HIP events on the
        # stream it is launched on (torch's current stream)
        feat_nhwc = feat5.permute(0, 1, 3, 4, 2).contiguous()
        shape = (1, Z, Y, X, C)

        def kernel_only():
            return bp._fused_forward(depth5, feat_nhwc, vt.ranks_depth,
                                     vt.ranks_feat, vt.ranks_bev,
                                     vt.interval_starts, vt.interval_lengths,
                                     shape, _lib.LAYOUT_BCZYX, out=vt._out_buf)
        for _ in range(10):
            kernel_only()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            kernel_only()
        e1.record()
        torch.cuda.synchronize()
        kernel_ms = e0.elapsed_time(e1) / args.steps

        # ---- per-call leg (what every VEON config runs: accelerate=False):
        # geometry + prepare + pool each step, no host sync, hipGraph-captured
        percall = None
        try:
            cfg2 = dict(cfg, accelerate=False)
            vt2 = build_neck(cfg2).to(dev).eval()
            vt2.sync_free = True
            vt2.persistent_output = vt.persistent_output

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
            with torch.cuda.graph(g2):
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
                       'what': 'view_transform(accelerate=False, sync_free): camera '
                               'matrices + fused geometry/counting-sort prepare + '
                               'plan + pool per step, hipGraph'}
        except Exception as e:  # report, do not hide
            print('per-call leg failed: %r' % (e,), file=sys.stderr)

    alg = algorithmic_bytes(n_cams, hf, wf, C, D, p_kept, n_int, Z * Y * X)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    value = (world if args.shard == 'replicas' else 1) * args.steps / elapsed

    traffic, traffic_source = args.pmc_traffic, 'command line'
    if traffic is None:
        traffic, traffic_source = committed_pmc_traffic(args.workload)
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
            'output_volume': ('persistent, best-placed of %d allocations (kernel %.1f us vs '
                              'median %.1f us; veon_amd/placement.py)' % (
                                  vt.placement_info['candidates'],
                                  vt.placement_info['best_ms'] * 1e3,
                                  vt.placement_info['median_ms'] * 1e3)
                              if vt.placement_info else 'allocator-placed, fresh per step'),
            'parallelism': ('replicas x%d (one sample per GPU, no collective)' % world
                            if args.shard == 'replicas' else
                            'cameras sharded over %d GPUs + all-reduce of the volume' % world),
        },
        'roofline': {
            'kernel': 'k_pool_fused_cf (bev_pool_v2 fused zero-fill+pool+layout)',
            'bound': 'hbm',
            'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic,
            'traffic_source': traffic_source,
            'algorithmic_bytes': alg,
            'kernel_ms': round(kernel_ms, 5),
        },
    }

    if percall is not None:
        result['percall_prepare'] = percall
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args, grid, input_size, n_cams, C,
                                              rig, depth5, feat5)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, grid, input_size, n_cams, C, rig, depth5, feat5):
    """The same step (pool with cached ranks -> (B,C,Z,Y,X)) by the pure-PyTorch
    index_add_ port on the host cores.  The oracle is only the thing timed
    here, never part of the product path."""
    from oracle import lss_torch
    threads = torch.get_num_threads()
    lower, interval, gsize = lss_torch.grid_infos(grid)
    fr = lss_torch.make_frustum(grid['depth'], input_size, 16)
    cams = (rig['sensor2ego'], rig['intrins'], rig['post_rots'],
            rig['post_trans'], rig['bda'])
    coor = lss_torch.lidar_coor(fr, *cams)
    ranks = lss_torch.voxel_prepare(coor, lower, interval, gsize)
    d, f = depth5.cpu(), feat5.cpu()
    for _ in range(2):
        lss_torch.lift(fr, (lower, interval, gsize), cams, d, f, ranks=ranks)
    n, t0 = 0, time.perf_counter()
    while True:
        lss_torch.lift(fr, (lower, interval, gsize), cams, d, f, ranks=ranks)
        n += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or n >= 200:
            break
    return {
        'value': round(n / el, 3), 'unit': 'samples/s', 'cores': threads,
        'kind': 'port',
        'sample': '%d iterations (%.1f s) of oracle.lss_torch.lift with cached '
                  'ranks on the %s workload, torch %d threads, fp32'
                  % (n, el, args.workload, threads),
    }


if __name__ == '__main__':
    main()

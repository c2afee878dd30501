"""bench.py's own launch path: `python bench.py --gpus N` without
a launcher must start N ranks itself (the reference's protocol is a launcher that
spawns one process per GPU: tools/dist_test.sh:11-22, tools/test.py:178), rank 0
prints ONE line with n_gpus = N, and a launcher whose world size differs from --gpus
is an error.

CPU: `--dry-run` (gloo, a sleep as the step) drives launch + rendezvous + the
barrier / MAX-over-ranks timing protocol without a GPU.
GPU: the real S2 step and the camera-sharded VEON-L step at N = 2 on ONE GPU under
VEON_BENCH_REHEARSAL=1 (both ranks on device 0, gloo instead of RCCL)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                        'TORCHELASTIC_RUN_ID')}
    env.update(extra)
    return env


def _json_lines(stdout):
    out = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith('{') and line.endswith('}'):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_gpus_n_launches_n_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run', '--steps', '3',
                        '--warmup', '1'], env=_env(), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout           # rank 0 only
    line = lines[0]
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['dry_run'] is True and line['ms_per_step'] >= 1.0   # 1 ms sleep per step


def test_single_rank_dry_run_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, '--dry-run', '--steps', '2', '--warmup', '0'],
                       env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _json_lines(r.stdout)
    assert line['n_gpus'] == 1


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'],
                       env=_env(WORLD_SIZE='3', RANK='0'), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0
    assert 'world size 3 != --gpus 2' in r.stderr + r.stdout


def test_driver_launch_shape_is_accepted():
    """The driver's own command shape (torch.distributed.run ... bench.py --gpus N)."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                        '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29653', BENCH, '--gpus', '2', '--dry-run',
                        '--steps', '2', '--warmup', '1'], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _json_lines(r.stdout)
    assert line['n_gpus'] == 2


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_lift_rehearsal():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '20', '--warmup', '3',
                        '--no-pmc', '--no-rocprof', '--no-cpu-baseline'],
                       env=_env(VEON_BENCH_REHEARSAL='1'), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = _json_lines(r.stdout)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert 'sv' not in line and 'veonb' not in line     # N = 1 extras only


@pytest.mark.gpu
@pytest.mark.parametrize('reduce', ['scatter', 'allreduce'])
def test_two_ranks_camera_sharded_veonl_rehearsal(reduce):
    """BASELINE configs[3] control flow at full size: VEON-L, 6 cameras 256x704 split
    3 + 3 over two ranks (on one GPU, gloo), hipGraph segments around the collectives."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', 'VEONL', '--shard',
                        'cameras', '--reduce', reduce, '--steps', '3', '--warmup', '1'],
                       env=_env(VEON_BENCH_REHEARSAL='1'), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = _json_lines(r.stdout)
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    assert 'hipGraph segments' in line['config']['launch'], line['config']['launch']

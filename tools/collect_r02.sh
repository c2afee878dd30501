#!/bin/bash
# Copy the evidence of tools/gpu_profile_r02.sh (gpurun_out/<tag>/) into profiles/ under
# the round's names.  Usage: bash tools/collect_r02.sh [tag]
TAG=${1:-r02_final}
S=gpurun_out/$TAG
P=profiles
cp $S/pytest_gpu.log $P/r02_pytest_gpu_final.log
cp $S/bench_default.json $P/r02_bench_default_final.json
cp $S/bench_profiled.json $P/r02_bench_profiled.json
cp $S/kernel_stats_bench.csv $P/r02_kernel_stats_bench.csv
cp $S/bench_veonl.json $P/r02_bench_veonl_final.json
cp $S/path_table.txt $P/r02_path_trace_final.txt
grep -v amdgpu.ids $S/graph_path.txt > $P/r02_graph_path_final.txt
python3 - "$S" <<'PY' > $P/r02_pmc_pool_kernels.txt
import sys
src = sys.argv[1] + '/pmc_pool/summary.txt'
keep, on = [], False
for line in open(src):
    if line.startswith('== '):
        on = any(k in line for k in ('k_pool_fused_cf', 'k_rows_fused_cf', 'k_rows_maxpool'))
    if on:
        keep.append(line.rstrip()[:200])
print('# rocprofv3 --pmc passes (one counter set per pass, no trace domains) and a separate')
print('# --kernel-trace pass of `python3 tools/pool_case.py ALL` (tools/pmc_run.sh): per-launch')
print('# means for the dominant pool kernels of S2 (k_pool_fused_cf) and SV (k_rows_*).')
print('# FETCH_SIZE / WRITE_SIZE are KiB per launch; traffic = WRITE_SIZE + 2 x FETCH_SIZE (gfx950).')
print('\n'.join(keep))
PY
ls -la $P | grep r02_ | awk '{print $5, $9}'

#!/bin/bash
# End-of-round evidence run on the GPU box (one gpurun call):
#   pytest -m gpu, smoke, the DEFAULT bench line, a rocprofv3 kernel trace of the same
#   bench command (PMC children and CPU baseline off: they launch no kernels of the
#   timed region), the PMC passes of the dominant pool kernels, VEON-L line, path trace.
# Usage: bash tools/gpu_profile_r02.sh [tag]   -> gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r02_final}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -n 3 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?"; tail -n 1 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench exit $?"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- \
  python3 $R/bench.py --no-pmc --no-cpu-baseline --no-sv --no-veonb --no-veonl \
  > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
echo "bench trace exit $?"
cd $R
cp $(find $OUT/bench_trace -name "*kernel_stats.csv" | head -n 1) $OUT/kernel_stats_bench.csv
grep -i "k_pool_fused_cf" $OUT/kernel_stats_bench.csv | cut -c1-200
bash tools/pmc_run.sh $TAG/pmc_pool "tools/pool_case.py ALL" k_ > $OUT/pmc_pool.log 2>&1
echo "pmc exit $?"
timeout -k 10 400 python bench.py --workload VEONL --no-pmc > $OUT/bench_veonl.json 2> $OUT/bench_veonl.err
echo "veonl exit $?"
bash tools/gpu_profile_path.sh $TAG/path > $OUT/path.log 2>&1
python3 tools/trace_table.py $OUT/path/trace 23 70 > $OUT/path_table.txt
head -n 3 $OUT/path_table.txt
for a in "vitb" "vitl" "vitb --veon-res"; do timeout -k 10 300 python tools/graph_path.py $a 2>&1 | tail -n 1; done | tee $OUT/graph_path.txt

#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.
# Usage (from the repo root, via gpurun): bash tools/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"
cat $OUT/bench.json
tail -3 $OUT/bench.err

#!/bin/bash
# rocprofv3 kernel trace + PMC passes of bench.py on the GPU box.
# Usage: bash tools/gpu_profile.sh <tag>    (outputs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph --no-placement"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace exit $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch exit $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
echo "pmc write exit $?"
find $OUT -name "*.csv" | head -20

#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench.py command (placement-tuned output
# volume, launch mode chosen by the probe).  Usage: bash tools/gpu_profile_final.sh <tag>
set -o pipefail
TAG=${1:-prof_final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.log
echo "trace exit $?"
cut -c1-900 $OUT/bench.json
find $OUT -name "*kernel_stats.csv" -o -name "*kernel_trace.csv" | head

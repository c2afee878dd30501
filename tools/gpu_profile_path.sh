#!/bin/bash
# rocprofv3 kernel trace of the whole occupancy path.  Usage: bash tools/gpu_profile_path.sh <tag>
set -o pipefail
TAG=${1:-path}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/path_trace.py > $OUT/plain.log 2>&1
echo "plain exit $?"
grep forwards $OUT/plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/path_trace.py > $OUT/trace.log 2>&1
echo "trace exit $?"
grep forwards $OUT/trace.log
find $OUT -name "*kernel_stats.csv" | head -3

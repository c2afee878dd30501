#!/bin/bash
# rocprofv3 kernel trace of the temporal fusion (MFMA path only) on the GPU box.
# Usage: bash tools/gpu_profile_temporal.sh <tag>
set -o pipefail
TAG=${1:-temporal}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export ONLY_HIP=1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/temporal_bench.py > $OUT/trace.log 2>&1
echo "temporal trace exit $?"
grep -v "MIOpen\|amdgpu.ids" $OUT/trace.log | tail -5
find $OUT -name "*kernel_stats.csv" | head -3

#!/bin/bash
# rocprofv3 kernel trace + MFMA-busy PMC pass of the MFMA kernels (conv body,
# ViT GEMM / attention) on the GPU box.  Usage: bash tools/gpu_profile_mfma.sh <tag>
set -o pipefail
TAG=${1:-mfma}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B=$GRAFT_REPO_ROOT/tools/body_bench.py
A=$GRAFT_REPO_ROOT/tools/att_bench.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/body_trace -- python3 $B > $OUT/body_trace.log 2>&1
echo "body trace exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/body_pmc -- python3 $B > $OUT/body_pmc.log 2>&1
echo "body pmc exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/att_trace -- python3 $A > $OUT/att_trace.log 2>&1
echo "att trace exit $?"
find $OUT -name "*.csv" | head -20
V=$GRAFT_REPO_ROOT/tools/vit_bench.py
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vit_trace -- python3 $V > $OUT/vit_trace.log 2>&1
echo "vit trace exit $?"

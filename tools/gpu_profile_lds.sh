#!/bin/bash
# rocprofv3 counter passes (MFMA busy, LDS activity / bank conflicts, wait reasons)
# of the two MFMA main loops on the GPU box: the Conv3d body (tools/body_bench.py)
# and the encoder GEMMs (tools/vit_bench.py).  One --pmc set per pass, no tracing
# beside them.  Per-kernel means -> gpurun_out/<tag>/{body,vit}_summary.txt
#   bash tools/gpu_profile_lds.sh <tag>
set -o pipefail
TAG=${1:-lds}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
PASSES=(
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"
 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
)
for W in body vit; do
  S=$ROOT/tools/${W}_bench.py
  mkdir -p $OUT/$W
  i=0
  for P in "${PASSES[@]}"; do
    timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/$W/pmc$i -- python3 $S > $OUT/$W/pmc$i.log 2>&1
    echo "$W pmc pass $i ($P) exit $?"
    i=$((i+1))
  done
  python3 $ROOT/tools/pmc_table.py $OUT/$W "k_" > $OUT/${W}_summary.txt 2>&1
  tail -60 $OUT/${W}_summary.txt
done

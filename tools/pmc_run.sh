#!/bin/bash
# rocprofv3 counter passes (one --pmc set per pass, no tracing beside them) and a
# kernel trace of ONE command on the GPU box; per-kernel means go to
# gpurun_out/<tag>/summary.txt.
#   bash tools/pmc_run.sh <tag> "<python args>" [kernel-name filter]
set -o pipefail
TAG=$1; ARGS=$2; FILT=${3:-k_}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $ROOT/$ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($P) exit $?"
  i=$((i+1))
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/$ARGS > $OUT/trace.log 2>&1
echo "trace exit $?"
python3 $ROOT/tools/pmc_table.py $OUT "$FILT" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
# SQ counters of the attention kernel (one counter per pass): where a wave's cycles go.
export TMPDIR=/tmp
cd /tmp
for ctr in SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
           SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
           SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
           SQ_INSTS_SALU SQ_IFETCH SQ_WAVES_EQ_64 GRBM_GUI_ACTIVE SQ_LEVEL_WAVES; do
  rm -rf /tmp/pa
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/pa -- python3 $GRAFT_REPO_ROOT/tools/att_case.py "$@" > /tmp/pa.log 2>&1
  f=$(find /tmp/pa -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "$ctr: not collected ($(grep -i -m1 "error\|invalid\|not" /tmp/pa.log | cut -c1-100))"; continue; fi
  python3 - "$f" $ctr <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r['Counter_Name']==sys.argv[2] and 'attention' in r['Kernel_Name']]
v=[float(r['Counter_Value']) for r in rows]
print('%-26s n=%d mean %.5g' % (sys.argv[2], len(v), sum(v)/max(len(v),1)))
PY
done

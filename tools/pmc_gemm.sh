#!/bin/bash
# L2 hit / miss and fabric fetch of GEMM configurations (one counter per pass).
export TMPDIR=/tmp
cd /tmp
run() {  # label M N K cfg mode
  for ctr in TCC_HIT_sum TCC_MISS_sum FETCH_SIZE; do
    rm -rf /tmp/pg
    rocprofv3 --pmc $ctr --output-format csv -d /tmp/pg -- python3 $GRAFT_REPO_ROOT/tools/gemm_case.py $2 $3 $4 $5 $6 10 > /dev/null 2>&1
    f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$1" $ctr <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r['Counter_Name']==sys.argv[3] and ('gemm' in r['Kernel_Name'] or 'Cijk' in r['Kernel_Name'])]
by={}
for r in rows: by.setdefault(r['Kernel_Name'][:60],[]).append(float(r['Counter_Value']))
for k,v in by.items(): print('%-14s %-12s %-60s n=%d mean %.4g'%(sys.argv[2],sys.argv[3],k,len(v),sum(v)/len(v)))
PY
  done
}
run "Bfc2-small"  5406 768 3072 0 resid
run "Bfc2-256x128" 5406 768 3072 7 resid
run "Bqkv-256x256" 5406 2304 768 1 bf16
run "Bqkv-small"  5406 2304 768 0 bf16
run "Bqkv-torch"  5406 2304 768 0 torch

#!/bin/bash
# rocprofv3 kernel trace of the per-call lift (tools/percall_case.py): average duration per kernel
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-percall}
mkdir -p $OUT
cd /tmp
rm -rf /tmp/pc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python3 $GRAFT_REPO_ROOT/tools/percall_case.py S2 50 > $OUT/run.log 2>&1
f=$(find /tmp/pc -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%7d calls  avg %8.2f us  min %8.2f  max %8.2f  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Name'][:80]))
PY
cat $OUT/kernels.txt

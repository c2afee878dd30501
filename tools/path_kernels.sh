#!/bin/bash
# Per-kernel time of one forward of the occupancy path (rocprofv3 --kernel-trace --stats):
#   bash tools/path_kernels.sh <tag> [vitl]      -> gpurun_out/<tag>/kernels.txt
set -o pipefail
TAG=${1:-pathk}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ "$2" = "vitl" ] && export ENC=vitl
export N=10
cd /tmp
rm -rf /tmp/pk
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/tools/path_trace.py > $OUT/trace.log 2>&1
echo "trace exit $?"
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = 13.0   # 3 warm-up + 10 timed forwards
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('GPU-busy per forward: %.3f ms (%d kernel names)' % (tot / n / 1e6, len(rows)))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print('%8.1f us/fwd  %6.1f calls/fwd  avg %7.1f us  %s' % (
        float(r['TotalDurationNs']) / n / 1e3, float(r['Calls']) / n,
        float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
cat $OUT/kernels.txt | head -50

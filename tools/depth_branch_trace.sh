#!/bin/bash
# rocprofv3 kernel stats of the depth branch alone (estimate_depth: DA-V2 encoder + DPT head
# + resizes), 20 iterations.  Usage: bash tools/depth_branch_trace.sh <tag> [vitb|vitl]
set -o pipefail
TAG=${1:-depth}
ENC=${2:-vitb}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
cat > /tmp/depth_branch.py <<PY
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from veon_amd.models.veon_occ import VeonOccupancyPath
torch.manual_seed(0)
kw = dict(VeonOccupancyPath.VEON_L) if '$ENC' == 'vitl' else dict(encoder='vitb')
net = VeonOccupancyPath(input_size=(256, 704), **kw).to('cuda:0').eval()
img = torch.randn(6, 3, 256, 704, device='cuda:0')
with torch.no_grad():
    for _ in range(3):
        net.estimate_depth(img)
    torch.cuda.synchronize()
    for _ in range(20):
        net.estimate_depth(img)
    torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/depth_branch.py > $OUT/trace.log 2>&1
echo "trace exit $?"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/depth_branch_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 23 / 1e6
print('depth branch: %.3f ms of kernel time per call' % tot)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    print('  %7.3f ms  %6.1f x  avg %7.1f us  %s' % (float(r['TotalDurationNs']) / 23 / 1e6,
          int(r['Calls']) / 23, float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY

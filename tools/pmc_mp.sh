set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g/pmc
mkdir -p $OUT
cd /tmp
for ord in azimuth none; do
  for ctr in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
    VEON_COLD_ORDER=$ord rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_${ord}_$ctr -- python3 $GRAFT_REPO_ROOT/tools/pool_case.py SV rows_mp_bf16 0 10 > /dev/null 2>&1
    f=$(find /tmp/pmc_${ord}_$ctr -name "*counter_collection.csv" | head -1)
    python3 - "$f" $ord $ctr <<'PY'
import csv,sys
v=[float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'k_rows_maxpool' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[3]]
print(sys.argv[2], sys.argv[3], 'launches', len(v), 'mean', sum(v)/max(len(v),1))
PY
  done
done

#!/bin/bash
# Kernel trace of the decode ALONE, launched eagerly (no graph): tools/lab/dec_only_trace.sh [precision] [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
PREC=${1:-bf16}; TAG=${2:-dec_only}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/lab/dec_only.py 10 $PREC | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/tools/lab/dec_only.py 5 $PREC nograph > $OUT/${TAG}.log 2>&1
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv; rm -rf $OUT/${TAG}_trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/${TAG}_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY

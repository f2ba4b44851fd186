#!/bin/bash
# Kernel-trace only (one rocprofv3 run, ~1 min): gpurun_out/TAG_kernel_stats.csv + the top of it on stdout.
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/bench.py --steps 10 --cpu-clips 0 --parity-clips 0 $QT_ARGS > $OUT/${TAG}_trace.log 2>&1
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/${TAG}_kernel_stats.csv')))
for r in rows[:${2:-24}]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
tail -1 $OUT/${TAG}_trace.log | cut -c1-160

#!/bin/bash
# Kernel trace of the encoder ALONE (nothing on other streams): compare with tools/quick_trace.sh (the pipeline) on one box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/lab/enc_only.py 30 | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_only_trace -o run -- python3 $ROOT/tools/lab/enc_only.py 20 > $OUT/enc_only.log 2>&1
cp $(find $OUT/enc_only_trace -name "*kernel_stats.csv" | head -1) $OUT/enc_only_kernel_stats.csv; rm -rf $OUT/enc_only_trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/enc_only_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY

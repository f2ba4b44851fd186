#!/bin/bash
# Launch gaps of the encoder ALONE: kernel trace of 20 encodes, busy time against wall time: tools/lab/enc_gaps.sh [precision] [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
PREC=${1:-bf16}; TAG=${2:-enc_gaps}; BATCH=${3:-64}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/lab/enc_only.py 30 24 $PREC $BATCH | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/tools/lab/enc_only.py 20 24 $PREC $BATCH > $OUT/${TAG}.log 2>&1
python3 - <<PY | tee $OUT/${TAG}.txt
import csv, glob
f = glob.glob('$OUT/${TAG}_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the last 20 encodes: find the log-mel launches
starts = [i for i, r in enumerate(rows) if 'cn_logmel' in r[2]]
first = starts[-20]
seg = rows[first:]
busy = sum(e - s for s, e, _ in seg) / 1e3
wall = (seg[-1][1] - seg[0][0]) / 1e3
gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
print("kernels %d (%.1f per encode)  busy %.1f us  wall %.1f us  gaps %.1f us = %.2f %% of wall; per encode: busy %.1f wall %.1f" % (
    len(seg), len(seg) / 20, busy, wall, sum(gaps), 100 * sum(gaps) / wall, busy / 20, wall / 20))
import collections
by = collections.defaultdict(list)
for i, g in enumerate(gaps): by[seg[i][2][:60] + ' -> ' + seg[i + 1][2][:40]].append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("%8.2f us avg x %4d  %s" % (sum(v) / len(v), len(v), k))
PY
rm -rf $OUT/${TAG}_trace

#!/bin/bash
# Kernel trace of the encoder ALONE (nothing on other streams), per (kernel, grid size): tools/lab/enc_only_trace.sh [precision] [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
PREC=${1:-bf16}; TAG=${2:-enc_only}; BATCH=${3:-64}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/lab/enc_only.py 30 24 $PREC $BATCH | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/tools/lab/enc_only.py 20 24 $PREC $BATCH > $OUT/${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/${TAG}_trace/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:90], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in acc.values())
with open('$OUT/${TAG}_by_grid.csv', 'w') as o:
    o.write('kernel,grid,calls,avg_us,share\n')
    for (n, g), (c, t) in rows:
        o.write(f'"{n}",{g},{c},{t / c:.1f},{t / tot:.4f}\n')
for (n, g), (c, t) in rows[:30]:
    print(f"{n[:84]:84s} grid {g:8d} calls {c:5d} avg {t / c:8.1f} us {100 * t / tot:5.1f}%")
PY
rm -rf $OUT/${TAG}_trace

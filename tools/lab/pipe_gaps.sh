#!/bin/bash
# Where the encoder stream idles inside the pipeline: kernel trace of a short bench run, gaps between consecutive ENCODER kernels.
#   bash tools/lab/pipe_gaps.sh TAG [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-pipe_gaps}
cd /tmp && export TMPDIR=/tmp
export CN_SCHED_TRIAL=0
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/bench.py --steps 24 --warmup 8 --cpu-clips 0 --parity-clips 0 "${@:2}" > $OUT/${TAG}.log 2>&1
tail -1 $OUT/${TAG}.log | cut -c1-200
python3 - <<PY | tee $OUT/${TAG}.txt
import csv, glob, collections
f = glob.glob('$OUT/${TAG}_trace/**/*kernel_trace.csv', recursive=True)[0]
rd = list(csv.DictReader(open(f)))
print(list(rd[0].keys()))
ENC = ('cn_logmel', 'cn_stem', 'cn_dwconv', 'cn_mlp_', 'cn_down_', 'cn_ln_patchify', 'cn_frame_mean', 'Li224E', '<224')
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in rd)
enc = [r for r in rows if any(k in r[2] for k in ENC)]
starts = [i for i, r in enumerate(enc) if 'cn_logmel' in r[2]]
seg = enc[starts[-21]:starts[-1]]   # the last 20 whole encodes
busy = sum(e - s for s, e, *_ in seg) / 1e3
wall = (seg[-1][1] - seg[0][0]) / 1e3
gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
print("encoder kernels %d  busy %.1f us  wall %.1f us  idle %.1f us = %.2f %% of wall; per encode: busy %.1f wall %.1f idle %.1f" % (
    len(seg), busy, wall, sum(gaps), 100 * sum(gaps) / wall, busy / 20, wall / 20, sum(gaps) / 20))
print("queues / streams of the encoder kernels:", collections.Counter((r[3], r[4]) for r in seg))
by = collections.defaultdict(list)
for i, g in enumerate(gaps): by[seg[i][2][:50] + ' -> ' + seg[i + 1][2][:40]].append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("%8.2f us avg (max %7.1f) x %4d  %s" % (sum(v) / len(v), max(v), len(v), k))
print("gap before each encode (us):", [round((seg[i + 1][0] - seg[i][1]) / 1e3) for i in range(len(seg) - 1) if 'cn_logmel' in seg[i + 1][2]])
# decode activity: runs of non-encoder kernels on the other queues
dq = [r for r in rows if not any(k in r[2] for k in ENC) and r[3] != seg[0][3] and r[0] >= seg[0][0] and r[1] <= seg[-1][1]]
runs = []
for r in dq:
    if runs and r[0] - runs[-1][1] < 200e3 and r[3] == runs[-1][3]: runs[-1][1] = r[1]; runs[-1][2] += 1; runs[-1][4] += r[1] - r[0]
    else: runs.append([r[0], r[1], 1, r[3], r[1] - r[0]])
t0 = seg[0][0]
print("decode runs (queue, start ms, span ms, kernels, busy ms):", [(q, round((a - t0) / 1e6, 2), round((b - a) / 1e6, 2), n, round(bz / 1e6, 2)) for a, b, n, q, bz in runs])
print("encode starts (ms):", [round((r[0] - t0) / 1e6, 2) for r in seg if 'cn_logmel' in r[2]])
other = [r for r in rows if not any(k in r[2] for k in ENC) and r[0] >= seg[0][0] and r[1] <= seg[-1][1]]
print("other kernels in the window: %d, busy %.1f us per encode" % (len(other), sum(e - s for s, e, *_ in other) / 1e3 / 20))
PY
rm -rf $OUT/${TAG}_trace

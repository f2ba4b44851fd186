#!/bin/bash
# The first N > 1 lease in one command: weak-scaling lines at 1 / 2 / 4 / 8 GPUs and BASELINE configs[3]
# (2048 clips sharded over 8 GPUs) with what proves that RCCL joined the ranks, next to the single-GPU caption hash of the same clips.
#   bash tools/scale_round.sh [TAG]   -> gpurun_out/TAG_scale_{n1,n2,n4,n8,g2048_n8,g2048_n1}.json + a summary table
# bench.py --gpus N starts its own ranks (a child torch.distributed.run on 127.0.0.1) before it touches a GPU.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "[scale_round] $NGPU GPUs visible"
line() { name=$1; shift; timeout 1200 python3 bench.py "$@" --cpu-clips 0 --parity-clips 0 --also "" 2> $OUT/${TAG}_scale_${name}.err | tail -1 > $OUT/${TAG}_scale_${name}.json; }
for n in 1 2 4 8; do
  if [ $n -le $NGPU ]; then line n$n --gpus $n --steps 100 --warmup 5; fi
done
# (CN_BENCH_ROTATE=1: one input batch instead of four rotating ones -- 2048 synthetic clips take minutes to generate on the host; both lines alike)
export CN_BENCH_ROTATE=1
if [ 8 -le $NGPU ]; then line g2048_n8 --gpus 8 --global-batch 2048 --steps 8 --warmup 2; fi
# the same 2048 clips on ONE GPU: the reference hash of the sharded job (captions_sha256 = the trimmed ids of the last step, clip order)
line g2048_n1 --gpus 1 --global-batch 2048 --steps 8 --warmup 2
unset CN_BENCH_ROTATE
python3 - <<PY
import glob, json, os
rows = {}
for f in sorted(glob.glob("$OUT/${TAG}_scale_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
        continue
    name = os.path.basename(f)[len("${TAG}_scale_"):-5]
    rows[name] = d
    c = d["config"]
    print(f"{name:10s} n_gpus {d['n_gpus']} value {d['value']:>10} clips/s  scaling {d['scaling']:6s} world_size_observed {c['world_size_observed']} "
          f"allreduce_of_ones {c['allreduce_of_ones']} collective_backend {c['collective_backend']} per-rank {d.get('rank_clips_per_sec')} "
          f"gather {d.get('gather')} sha {d['captions_sha256'][:16]}")
if "n1" in rows:
    for n in (2, 4, 8):
        if f"n{n}" in rows:
            print(f"weak-scaling efficiency at {n} GPUs: {rows[f'n{n}']['value'] / (n * rows['n1']['value']):.3f}")
if "g2048_n8" in rows and "g2048_n1" in rows:
    same = rows["g2048_n8"]["captions_sha256"] == rows["g2048_n1"]["captions_sha256"]
    print("2048 clips over 8 GPUs return the single-GPU captions:", same)
PY

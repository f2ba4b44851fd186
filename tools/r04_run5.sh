cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_f}
python -m pytest tests/test_gpu_bench_n2.py tests/test_gpu_bucketing.py tests/test_gpu_exact.py -x -q -m gpu 2>&1 | tail -6
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "topk_decisions or greedy_search" 2>&1 | grep -E "top-k at reference|greedy greedy" > gpurun_out/${T}_floors.txt
wc -l gpurun_out/${T}_floors.txt
run() { name=$1; shift; env "$@" python bench.py --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" $EXTRA 2>gpurun_out/${T}_$name.err | tail -1 > gpurun_out/${T}_$name.json; }
EXTRA="--steps 20" run bf16_a X=1
EXTRA="--steps 20" run bf16_cu64 CN_DEC_CUS=64
EXTRA="--steps 20" run bf16_b X=1
EXTRA="--steps 20" run bf16_cu64b CN_DEC_CUS=64
EXTRA="--steps 60 --batch 16" run b16_d2 CN_DEC_STREAMS=2
EXTRA="--steps 60 --batch 16" run b16_d3 CN_DEC_STREAMS=3
EXTRA="--steps 60 --batch 16" run b16_d3_cu64 CN_DEC_STREAMS=3 CN_DEC_CUS=64
EXTRA="--steps 10 --workload mixed" run mix64_old CN_BUCKET_SECONDS=960
EXTRA="--steps 10 --workload mixed" run mix64_f50 CN_BUCKET_FIXED=50
EXTRA="--steps 10 --workload mixed" run mix64_f150 CN_BUCKET_FIXED=150
EXTRA="--steps 10 --workload mixed" run mix64_f250 CN_BUCKET_FIXED=250
EXTRA="--steps 10 --workload mixed --batch 128" run mix128_f100 CN_BUCKET_FIXED=100
EXTRA="--steps 10 --workload mixed --batch 128" run mix128_f150 CN_BUCKET_FIXED=150
EXTRA="--steps 10 --workload mixed --batch 128" run mix128_old CN_BUCKET_SECONDS=960
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("audio_seconds_per_sec"), d.get("encode_ms"), d.get("decode_ms"))
    except Exception as e: print(f, "ERR", e)
PY

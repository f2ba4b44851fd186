cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_k}
env X=1 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_a.err | tail -1 > gpurun_out/${T}_bf16_a.json
env CN_DEC_CUS=32 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu32.err | tail -1 > gpurun_out/${T}_bf16_cu32.json
env CN_DEC_CUS=48 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu48.err | tail -1 > gpurun_out/${T}_bf16_cu48.json
env CN_DEC_CUS=64 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu64.err | tail -1 > gpurun_out/${T}_bf16_cu64.json
env CN_DEC_CUS=96 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu96.err | tail -1 > gpurun_out/${T}_bf16_cu96.json
env X=1 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_b.err | tail -1 > gpurun_out/${T}_bf16_b.json
env CN_DEC_CUS=32 CN_ENC_RESERVE=32 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu32_r32.err | tail -1 > gpurun_out/${T}_bf16_cu32_r32.json
env CN_DEC_CUS=32 CN_DEC_STREAMS=3 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu32_d3.err | tail -1 > gpurun_out/${T}_bf16_cu32_d3.json
env X=1 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_a.err | tail -1 > gpurun_out/${T}_mixed16_a.json
env CN_DEC_CUS=32 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_cu32.err | tail -1 > gpurun_out/${T}_mixed16_cu32.json
env CN_DEC_CUS=64 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_cu64.err | tail -1 > gpurun_out/${T}_mixed16_cu64.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"))
    except Exception as e: print(f, "ERR", e)
PY

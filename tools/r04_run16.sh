cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_s}
env CN_ENC_RESERVE=24 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r24.err | tail -1 > gpurun_out/${T}_bf16_r24.json
env CN_ENC_RESERVE=0 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r0.err | tail -1 > gpurun_out/${T}_bf16_r0.json
env CN_ENC_RESERVE=8 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r8.err | tail -1 > gpurun_out/${T}_bf16_r8.json
env CN_ENC_RESERVE=16 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r16.err | tail -1 > gpurun_out/${T}_bf16_r16.json
env CN_ENC_RESERVE=20 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r20.err | tail -1 > gpurun_out/${T}_bf16_r20.json
env CN_ENC_RESERVE=24 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r24_b.err | tail -1 > gpurun_out/${T}_bf16_r24_b.json
env CN_ENC_RESERVE=0 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_r0_b.err | tail -1 > gpurun_out/${T}_bf16_r0_b.json
env CN_ENC_RESERVE=24 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_r24.err | tail -1 > gpurun_out/${T}_mixed16_r24.json
env CN_ENC_RESERVE=0 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_r0.err | tail -1 > gpurun_out/${T}_mixed16_r0.json
env CN_ENC_RESERVE=16 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_r16.err | tail -1 > gpurun_out/${T}_mixed16_r16.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["pipeline_consistent"], d["windows"]["clips_per_sec"])
    except Exception as e: print(f, "ERR", e)
PY

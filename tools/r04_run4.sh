cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_e}
 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_a.err | tail -1 > gpurun_out/${T}_bf16_a.json
CN_DEC_CUS=32 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu32.err | tail -1 > gpurun_out/${T}_bf16_cu32.json
CN_DEC_CUS=40 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu40.err | tail -1 > gpurun_out/${T}_bf16_cu40.json
CN_DEC_CUS=64 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu64.err | tail -1 > gpurun_out/${T}_bf16_cu64.json
CN_DEC_CUS=40 CN_ENC_RESERVE=40 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_cu40_r40.err | tail -1 > gpurun_out/${T}_bf16_cu40_r40.json
 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_b.err | tail -1 > gpurun_out/${T}_bf16_b.json
 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_a.err | tail -1 > gpurun_out/${T}_mixed16_a.json
CN_DEC_CUS=40 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_cu40.err | tail -1 > gpurun_out/${T}_mixed16_cu40.json
CN_DEC_CUS=64 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_cu64.err | tail -1 > gpurun_out/${T}_mixed16_cu64.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T","r04_c")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"))
    except Exception as e: print(f, "ERR", e)
PY

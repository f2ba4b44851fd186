cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_o}
env CN_DEC_GROUP=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g1.err | tail -1 > gpurun_out/${T}_bf16_g1.json
env CN_DEC_GROUP=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g2.err | tail -1 > gpurun_out/${T}_bf16_g2.json
env CN_DEC_GROUP=3 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g3.err | tail -1 > gpurun_out/${T}_bf16_g3.json
env CN_DEC_GROUP=4 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g4.err | tail -1 > gpurun_out/${T}_bf16_g4.json
env CN_DEC_GROUP=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g1b.err | tail -1 > gpurun_out/${T}_bf16_g1b.json
env CN_DEC_GROUP=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g2b.err | tail -1 > gpurun_out/${T}_bf16_g2b.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g2_d1.err | tail -1 > gpurun_out/${T}_bf16_g2_d1.json
env CN_DEC_GROUP=4 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g4_d1.err | tail -1 > gpurun_out/${T}_bf16_g4_d1.json
env CN_DEC_GROUP=1 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g1.err | tail -1 > gpurun_out/${T}_mixed16_g1.json
env CN_DEC_GROUP=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g2.err | tail -1 > gpurun_out/${T}_mixed16_g2.json
env CN_DEC_GROUP=3 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g3.err | tail -1 > gpurun_out/${T}_mixed16_g3.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g2_d2.err | tail -1 > gpurun_out/${T}_mixed16_g2_d2.json
env CN_DEC_GROUP=1 python bench.py --precision exact --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_exact_g1.err | tail -1 > gpurun_out/${T}_exact_g1.json
env CN_DEC_GROUP=2 python bench.py --precision exact --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_exact_g2.err | tail -1 > gpurun_out/${T}_exact_g2.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"), d["pipeline_consistent"], d["pipeline_steps_checked"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY

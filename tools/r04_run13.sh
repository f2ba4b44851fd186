cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_p}
env CN_DEC_GROUP=1 CN_DEC_STREAMS=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g1_d2.err | tail -1 > gpurun_out/${T}_bf16_g1_d2.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g2_d1.err | tail -1 > gpurun_out/${T}_bf16_g2_d1.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g2_d2.err | tail -1 > gpurun_out/${T}_bf16_g2_d2.json
env CN_DEC_GROUP=3 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g3_d1.err | tail -1 > gpurun_out/${T}_bf16_g3_d1.json
env CN_DEC_GROUP=3 CN_DEC_STREAMS=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g3_d2.err | tail -1 > gpurun_out/${T}_bf16_g3_d2.json
env CN_DEC_GROUP=4 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g4_d1.err | tail -1 > gpurun_out/${T}_bf16_g4_d1.json
env CN_DEC_GROUP=4 CN_DEC_STREAMS=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g4_d2.err | tail -1 > gpurun_out/${T}_bf16_g4_d2.json
env CN_DEC_GROUP=6 CN_DEC_STREAMS=1 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g6_d1.err | tail -1 > gpurun_out/${T}_bf16_g6_d1.json
env CN_DEC_GROUP=1 CN_DEC_STREAMS=2 python bench.py --precision bf16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_g1_d2_b.err | tail -1 > gpurun_out/${T}_bf16_g1_d2_b.json
env CN_DEC_GROUP=1 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g1_d2.err | tail -1 > gpurun_out/${T}_mixed16_g1_d2.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=1 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g2_d1.err | tail -1 > gpurun_out/${T}_mixed16_g2_d1.json
env CN_DEC_GROUP=2 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g2_d2.err | tail -1 > gpurun_out/${T}_mixed16_g2_d2.json
env CN_DEC_GROUP=3 CN_DEC_STREAMS=1 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g3_d1.err | tail -1 > gpurun_out/${T}_mixed16_g3_d1.json
env CN_DEC_GROUP=3 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g3_d2.err | tail -1 > gpurun_out/${T}_mixed16_g3_d2.json
env CN_DEC_GROUP=4 CN_DEC_STREAMS=1 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g4_d1.err | tail -1 > gpurun_out/${T}_mixed16_g4_d1.json
env CN_DEC_GROUP=4 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g4_d2.err | tail -1 > gpurun_out/${T}_mixed16_g4_d2.json
env CN_DEC_GROUP=6 CN_DEC_STREAMS=1 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g6_d1.err | tail -1 > gpurun_out/${T}_mixed16_g6_d1.json
env CN_DEC_GROUP=1 CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 24 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_g1_d2_b.err | tail -1 > gpurun_out/${T}_mixed16_g1_d2_b.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["pipeline_consistent"], d["windows"]["clips_per_sec"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_q}
python -m pytest tests/test_gpu_bench_n2.py -x -q -m gpu 2>&1 | tail -3
env X=1 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_s20.err | tail -1 > gpurun_out/${T}_bf16_s20.json
env X=1 python bench.py --precision bf16 --steps 100 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_s100.err | tail -1 > gpurun_out/${T}_bf16_s100.json
env CN_DEC_GROUP=1 python bench.py --precision bf16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16_s20_g1.err | tail -1 > gpurun_out/${T}_bf16_s20_g1.json
env X=1 python bench.py --precision f16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_f16_s20.err | tail -1 > gpurun_out/${T}_f16_s20.json
env X=1 python bench.py --precision bf16+f16dec --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_bf16f16dec_s20.err | tail -1 > gpurun_out/${T}_bf16f16dec_s20.json
env X=1 python bench.py --precision exact --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_exact_s20.err | tail -1 > gpurun_out/${T}_exact_s20.json
env CN_DEC_GROUP=2 python bench.py --precision exact --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_exact_g2.err | tail -1 > gpurun_out/${T}_exact_g2.json
env CN_DEC_GROUP=1 python bench.py --precision exact --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_exact_g1.err | tail -1 > gpurun_out/${T}_exact_g1.json
env X=1 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_mixed16_s20.err | tail -1 > gpurun_out/${T}_mixed16_s20.json
env X=1 python bench.py --precision bf16 --steps 12 --batch 256 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b256_g4.err | tail -1 > gpurun_out/${T}_b256_g4.json
env CN_DEC_GROUP=2 python bench.py --precision bf16 --steps 12 --batch 256 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b256_g2.err | tail -1 > gpurun_out/${T}_b256_g2.json
env CN_DEC_GROUP=1 python bench.py --precision bf16 --steps 12 --batch 256 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b256_g1.err | tail -1 > gpurun_out/${T}_b256_g1.json
env X=1 python bench.py --precision bf16 --steps 64 --batch 16 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b16_g4.err | tail -1 > gpurun_out/${T}_b16_g4.json
env CN_DEC_GROUP=8 python bench.py --precision bf16 --steps 64 --batch 16 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b16_g8.err | tail -1 > gpurun_out/${T}_b16_g8.json
env CN_DEC_GROUP=16 python bench.py --precision bf16 --steps 64 --batch 16 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b16_g16.err | tail -1 > gpurun_out/${T}_b16_g16.json
env CN_DEC_GROUP=1 python bench.py --precision bf16 --steps 64 --batch 16 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_b16_g1.err | tail -1 > gpurun_out/${T}_b16_g1.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["pipeline_consistent"], d["config"]["decode_group"], d["config"]["decode_streams"], d["windows"]["clips_per_sec"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY

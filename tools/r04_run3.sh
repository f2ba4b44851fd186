cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_c}
python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -x -q -m gpu -k "exact" 2>&1 | tail -5
 python bench.py --precision f16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_f16.json
CN_DEC_STREAMS=2 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_mixed16_d2.json
CN_DEC_STREAMS=3 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_mixed16_d3.json
CN_DEC_STREAMS=2 python bench.py --precision exact --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_exact_d2.json
 python bench.py --precision f16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_f16_b.json
CN_DEC_STREAMS=3 python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/${T}_mixed16_d3b.json
python - <<'PY'
import json,glob,sys,os
T=os.environ.get("T","r04_c")
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"))
    except Exception as e: print(f, "ERR", e)
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_b_pytest.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "topk_decisions or greedy_search" 2>&1 | grep -E "^top-k at|^greedy " > gpurun_out/r04_b_floors.txt
for n in 2 3; do
  CN_DEC_STREAMS=$n python bench.py --precision mixed16 --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/r04_b_mixed16_dec$n.json
  CN_DEC_STREAMS=$n python bench.py --precision exact --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 > gpurun_out/r04_b_exact_dec$n.json
done
python bench.py --steps 20 > gpurun_out/r04_b_default.json 2> gpurun_out/r04_b_default.err
cat gpurun_out/r04_b_pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"), {k:v.get("clips_per_sec") for k,v in d.get("also_pipelined",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY

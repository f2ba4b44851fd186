cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_i}
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/lab/dec_only.py 20 bf16 | tail -1
python tools/lab/dec_only.py 20 exact | tail -1
python bench.py --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 64 --also "" 2>gpurun_out/${T}_bf16.err | tail -1 > gpurun_out/${T}_bf16.json
python - <<'PY'
import json,os
d=json.loads(open("gpurun_out/%s_bf16.json" % os.environ["T"]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["encode_ms"], d["decode_ms"])
for k,v in d["parity"]["vs_fp32_mode"].items(): print(k, v["greedy"]["seq_identical"], v["beam3"]["seq_identical"])
PY

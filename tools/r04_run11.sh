cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_n}
python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -x -q -m gpu -k "decode or fused or exact" 2>&1 | tail -3
b() { name=$1; prec=$2; python bench.py --precision $prec --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_$name.err | tail -1 > gpurun_out/${T}_$name.json; }
b x8_bf16 bf16; b x8_mixed16 mixed16; b x8_f16 f16; b x8_exact exact
CN_DB_XCDS=4 python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b x4_bf16 bf16; b x4_mixed16 mixed16
CN_DB_XCDS=2 python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b x2_bf16 bf16; b x2_mixed16 mixed16
python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b x8_bf16_b bf16
python - <<'PY'
import json,glob,os
T=os.environ["T"]
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"), d["pipeline_consistent"])
    except Exception as e: print(f, "ERR", e)
PY

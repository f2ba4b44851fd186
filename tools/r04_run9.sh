# same-box A/B of the rows per decoder block kernel (builds on the GPU box: only decoder.hip recompiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_j}
b() { name=$1; prec=$2; python bench.py --precision $prec --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>gpurun_out/${T}_$name.err | tail -1 > gpurun_out/${T}_$name.json; }
b r4_bf16 bf16; b r4_mixed16 mixed16
CN_DB_ROWS=8 CN_DB_ROWS_SP=8 python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b r8_bf16 bf16; b r8_mixed16 mixed16
CN_DB_ROWS=12 CN_DB_ROWS_SP=12 python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b r12_bf16 bf16; b r12_mixed16 mixed16
python tools/lab/dec_only.py 20 bf16 | tail -1
python conette-audio-captioning_amd/build.py > /dev/null 2>&1
b r4_bf16_b bf16
python - <<'PY'
import json,glob,os
T=os.environ["T"]
for f in sorted(glob.glob("gpurun_out/%s_*.json" % T)):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("encode_ms"), d.get("decode_ms"), d["pipeline_consistent"])
    except Exception as e: print(f, "ERR", e)
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_r}
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for p in bf16 mixed16 bf16; do python bench.py --precision $p --steps 20 --repeat 3 --cpu-clips 0 --parity-clips 0 --also "" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$p', d['value'], d['ms_per_step'], d['pipeline_consistent'], d['config']['decode_group'], d['windows']['clips_per_sec'])"; done

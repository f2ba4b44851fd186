#!/bin/bash
# The bench lines DESIGN.md quotes besides the default one ("keep the evidence you quote"):
#   bash tools/evidence_round.sh TAG   -> gpurun_out/TAG_{default,f16,mixed16,mixed_precision,b256,b16,mixedlen,shard256,soak,exact,greedy,certified*}.json (one JSON line each)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout 600 python3 bench.py "$@" 2> $OUT/${TAG}_${name}.err | tail -1 > $OUT/${TAG}_${name}.json; python3 -c "
import json,sys
d=json.load(open('$OUT/${TAG}_${name}.json'))
print('$name', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'windows', d['windows']['clips_per_sec'], 'consistent', d['pipeline_consistent'])"; }
if [ $# -gt 1 ]; then ONLY=" ${@:2} "; else ONLY=""; fi   # tools/evidence_round.sh TAG [name ...]: only these lines
want() { [ -z "$ONLY" ] || [[ "$ONLY" == *" $1 "* ]]; }
_run() { if want $1; then run "$@"; fi; }
_run default
# BASELINE configs[1]: B = 64 x 10 s, GREEDY decode (beam 1), bf16 -- the throughput line of that config
_run greedy --beam 1 --cpu-clips 0 --parity-clips 0
_run f16 --precision f16 --cpu-clips 0 --parity-clips 0
_run bf16_f16dec --precision bf16+f16dec --cpu-clips 0 --parity-clips 0
_run mixed16 --precision mixed16 --cpu-clips 0 --parity-clips 0
_run mixed_precision --precision mixed --cpu-clips 0 --parity-clips 0
_run b256 --batch 256 --steps 30 --cpu-clips 0 --parity-clips 0
_run b16 --batch 16 --steps 200 --cpu-clips 0 --parity-clips 0
_run mixedlen --workload mixed --steps 30 --cpu-clips 0 --parity-clips 0
_run shard256 --global-batch 256 --steps 30 --cpu-clips 0 --parity-clips 0
_run soak --steps 300 --repeat 5 --cpu-clips 0 --parity-clips 0
_run exact --precision exact --steps 30 --cpu-clips 0 --parity-clips 0
# the id-certified pipeline (round 6): fp16 + margins, uncertified clips re-run exactly; default and peaked synthetic checkpoint, beam 3 and greedy
cert() { name=$1; shift; if want $name; then timeout 600 python3 bench_certified.py "$@" 2> $OUT/${TAG}_${name}.err | tail -1 > $OUT/${TAG}_${name}.json; python3 -c "
import json
d=json.load(open('$OUT/${TAG}_${name}.json'))
print('$name', d['value'], 'clips/s, recompute', d['recompute_fraction'], 'ids identical to exact', d['ids_identical_to_exact'], 'consistent', d['pipeline_consistent'])"; fi; }
cert certified
cert certified_peaked --checkpoint peaked
cert certified_greedy --beam 1
cert certified_peaked_greedy --checkpoint peaked --beam 1
cert certified_f16_peaked --base f16 --checkpoint peaked
cert certified_best --policy best
cert certified_best_peaked --policy best --checkpoint peaked

#!/bin/bash
# A/B of two builds of the library on ONE box: tools/lab/ab.sh libA.so libB.so [rounds] [extra bench args]
L=conette-audio-captioning_amd/libconette_hip.so
cp $L /tmp/lib_keep.so
for r in $(seq 1 ${3:-2}); do
  for v in $1 $2; do
    cp tools/lab/$v $L
    echo -n "$v: "; python bench.py --cpu-clips 0 --parity-clips 0 "${@:4}" 2>&1 | tee -a gpurun_out/ab_last.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipeline_consistent'])"
  done
done
cp /tmp/lib_keep.so $L

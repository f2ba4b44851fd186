#!/bin/bash
# Same-box A/B of per-kernel averages: tools/lab/ab_trace.sh libA.so libB.so 'grep-pattern' [rounds]
L=conette-audio-captioning_amd/libconette_hip.so
for r in $(seq 1 ${4:-2}); do
  for v in $1 $2; do
    cp tools/lab/$v $L
    echo "== $v (round $r)"
    bash tools/quick_trace.sh ab_$v 40 | grep -E "$3" | cut -c1-110
  done
done

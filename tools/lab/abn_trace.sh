#!/bin/bash
# Same-box per-kernel averages of several library builds: tools/lab/abn_trace.sh 'grep-pattern' lib0.so lib1.so ...
L=conette-audio-captioning_amd/libconette_hip.so
P=$1; shift
for v in "$@"; do
  cp tools/lab/$v $L
  echo "== $v"
  CN_BENCH_STRICT=0 bash tools/quick_trace.sh abn_$v 40 | grep -E "$P" | cut -c1-110
done

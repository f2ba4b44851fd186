#!/bin/bash
# Repeat the pipeline-consistency tests N times with each given library build: tools/lab/soak_conc.sh N lib0.so [lib1.so ...]
L=conette-audio-captioning_amd/libconette_hip.so
N=$1; shift
for v in "$@"; do
  cp tools/lab/$v $L
  f=0
  for i in $(seq 1 $N); do
    python -m pytest tests/test_gpu_concurrency.py -q -x -k "pipelined or frontend" > /tmp/soak.log 2>&1 || { f=$((f+1)); grep -E "FAILED|assert" /tmp/soak.log | head -3; }
  done
  echo "== $v: $f failures of $N runs"
done

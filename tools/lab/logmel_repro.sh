#!/bin/bash
# Which build of the log-mel kernel is disturbed by which work beside it?
#   tools/lab/logmel_repro.sh lib1.so lib2.so ...   (libraries under tools/lab/, built with the FE_* knobs of build.py)
L=conette-audio-captioning_amd/libconette_hip.so
cp $L /tmp/lib_keep.so
for v in "$@"; do
  cp tools/lab/$v $L
  echo "== $v"
  PROBE_KINDS=${PROBE_KINDS:-none,decode_unfused,decode_fused} timeout 600 python tools/pipeline_probe3.py 2>&1 | grep -v Warning | tail -8
done
cp /tmp/lib_keep.so $L

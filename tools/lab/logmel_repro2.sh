#!/bin/bash
# tools/lab/logmel_repro2.sh lib.so [env ...]
L=conette-audio-captioning_amd/libconette_hip.so
cp $L /tmp/lib_keep.so
cp tools/lab/$1 $L
shift
env "$@" timeout 600 python tools/lab/logmel_repro2.py 2>&1 | grep -v "Warning\|amdgpu.ids"
cp /tmp/lib_keep.so $L

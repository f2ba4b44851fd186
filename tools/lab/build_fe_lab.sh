#!/bin/bash
# Lab build of the library with the instrumented log-mel kernel (tools/lab/frontend_lab.hip) in place of the product one:
#   tools/lab/build_fe_lab.sh lib_fe_nofill4.so FE_NOFILL=1 FE_NW=4 [FE_VARIANT=40] ["CN_FE_FLAGS=-fno-slp-vectorize -DFE_PK=4 -DCN_LAB"]
# -> tools/lab/<name>; the product library is rebuilt afterwards.  The runs behind profiles/r03_logmel_rootcause.log:
#   lib_fe_nofill4.so        FE_NOFILL=1 FE_NW=4                      (round 2's geometry: other workgroups share the CU)
#   lib_fe_v40.so            ... FE_VARIANT=40                        (every frame twice + per-stage comparison)
#   lib_fe_v296.so           ... FE_VARIANT=296                       (+ every LDS read group repeated and compared)
#   lib_fe_pk{0,1,2,4}.so    ... FE_VARIANT=40 "CN_FE_FLAGS=-fno-slp-vectorize -DFE_PK=n"   (packed fp32 only at one site)
#   lib_fe_lab.so            ... "CN_FE_FLAGS=-fno-slp-vectorize -DFE_PK=4 -DCN_LAB"        (synthetic co-runners)
set -e
name=$1; shift
cd "$(dirname "$0")/../.."
env CN_FE_SRC=tools/lab/frontend_lab.hip CN_ALLOW_PK_HAZARD=1 "$@" python -c "
import importlib
importlib.import_module('conette-audio-captioning_amd.build').build(force=True)"
cp conette-audio-captioning_amd/libconette_hip.so tools/lab/$name
python -c "
import importlib
importlib.import_module('conette-audio-captioning_amd.build').build(force=True)"

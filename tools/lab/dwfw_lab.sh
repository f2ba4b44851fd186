#!/bin/bash
# Build the phase-cut lab of the shipped full-width depthwise kernel: the kernel's source is cut out of csrc/encoder.hip (never copied into the
# repo) with two hooks added, then compiled three times.   bash tools/lab/dwfw_lab.sh && for a in 0 1 2; do tools/lab/dwfw_lab_$a 384; done
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=${TMPDIR:-/tmp}/fwlab; mkdir -p $TMP
python3 - "$ROOT" "$TMP" <<'PY'
import sys
root, tmp = sys.argv[1], sys.argv[2]
src = open(root + '/conette-audio-captioning_amd/csrc/encoder.hip').read()
a = src.index('template <int C, int WW, int TH, int OW0, int NOW, typename XT>\n__device__ __forceinline__ void cn_fw_conv')
b = src.index('// tile of the depthwise kernel of stages 0 / 1: 4 S columns x TH rows per block')
body = src[a:b]
body = body.replace('  if constexpr (SPLIT == 1) {\n#pragma unroll 1\n    for (int c = tid; c < C; c += CT)', '  if (FW_ABL != 2) {\n  if constexpr (SPLIT == 1) {\n#pragma unroll 1\n    for (int c = tid; c < C; c += CT)')
body = body.replace('      cn_fw_conv<C, WW, TH, WW / 2, WW / 2, XT>(xb0 + c, H, h0, dw_w, dw_wp, dw_b[c], c, s_v, PITCH);\n  }\n  __syncthreads();', '      cn_fw_conv<C, WW, TH, WW / 2, WW / 2, XT>(xb0 + c, H, h0, dw_w, dw_wp, dw_b[c], c, s_v, PITCH);\n  }\n  }\n  if (FW_ABL == 1) return;\n  __syncthreads();')
assert 'FW_ABL == 1' in body and 'FW_ABL != 2' in body
open(tmp + '/fw_extract.h', 'w').write(body)
PY
for abl in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $ROOT/conette-audio-captioning_amd/csrc -I $TMP -DFW_ABL=$abl $ROOT/tools/lab/dwfw_lab.hip -o $ROOT/tools/lab/dwfw_lab_$abl; done

#!/bin/bash
# registers / spills / LDS of every kernel of one csrc/*.hip file:   tools/kregs.sh encoder [name-filter]
set -e
f=$1; shift
out=${TMPDIR:-/tmp}/kregs_$f
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCN_RC2_GELU_PK --offload-device-only -c "$(dirname "$0")/../conette-audio-captioning_amd/csrc/$f.hip" -o $out.co 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$out.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out.elf 2>/dev/null || cp $out.co $out.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out.elf | python "$(dirname "$0")/kernel_regs.py" "$@"

"""ISA lint of the built library: no kernel may contain the instruction form that MI355X executes wrongly beside bf16 MFMAs.

Measured on gfx950 (tools/lab/pk_mfma_probe.hip; the sweeps over MFMA shapes and op_sel forms are summarised in profiles/r03_notes.md section 8): a packed-fp32 arithmetic
instruction -- v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32 -- whose op_sel routes the HIGH half of src1 into the low result
(`op_sel:[0,1...]`, any op_sel_hi) returns wrong values in lanes 48-63 in ~2 % of its executions while ANOTHER wave on the same SIMD is
executing v_mfma_f32_16x16x32_bf16 (not beside the f16 / fp8 / 32x32x16 / f32 MFMAs, plain VALU, LDS or memory traffic; every
other op_sel / neg combination, v_pk_mov_b32, v_fma_mix_f32 and the packed f16 instructions are unaffected).  The decoder's GEMMs are
made of exactly that MFMA and run on other streams beside every encoder kernel, so the form must not appear anywhere in the
library: hipcc only produces it when its SLP vectoriser packs complex arithmetic (the log-mel FFT, one horizontal add in the
decoder block kernel), which build.py switches off for those files; this check keeps it that way.
"""
import os
import re
import shutil
import subprocess
import tempfile
from typing import List, Tuple

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# v_pk_{mul,add,fma}_f32 ... op_sel:[0,1   (src0 selector 0, src1 selector 1)
HAZARD = re.compile(r"\bv_pk_(?:mul|add|fma)_f32\b.*\bop_sel:\[0,1[\],]")


def hazards_in_asm(lines) -> List[Tuple[str, str]]:
    """(kernel, instruction) of every occurrence of the hazardous form in a disassembly / assembly listing."""
    out, kern = [], "?"
    for line in lines:
        m = re.match(r"^[0-9a-f]*\s*<?([A-Za-z_][\w$.]*)>?:\s*(;.*)?$", line.strip())
        if m:
            kern = m.group(1)
            continue
        if HAZARD.search(line):
            out.append((kern, " ".join(line.split()[:12])))
    return out


def lint_library(lib_path: str) -> List[Tuple[str, str]]:
    """Disassemble every gfx950 code object bundled in `lib_path` and return the occurrences of the hazardous form."""
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} not found")
    tmp = tempfile.mkdtemp(prefix="cn_isa_lint_")
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp, capture_output=True, text=True, check=True)
        found, n_obj = [], 0
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            n_obj += 1
            r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], capture_output=True, text=True, check=True)
            found += hazards_in_asm(r.stdout.splitlines())
        if n_obj == 0:
            raise RuntimeError(f"no device code object found in {lib_path}")
        return found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    bad = lint_library(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "libconette_hip.so"))
    for k, ins in bad:
        print(f"{k}: {ins}")
    print(f"{len(bad)} hazardous packed-fp32 instruction(s)")
    sys.exit(1 if bad else 0)

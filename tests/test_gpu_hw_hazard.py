"""GPU: the hardware behaviour the library is built around (tools/lab/pk_mfma_probe.hip; profiles/r03_notes.md section 8).

MI355X executes a packed-fp32 instruction with op_sel:[0,1] wrongly (lanes 48-63) beside another wave's v_mfma_f32_16x16x32_bf16;
every other form the library's kernels use -- scalar fp32, packed fp32 without a src1 swizzle -- must be exact beside the same
MFMAs, or the GELU epilogues / conv kernels that share compute units with the decoder's GEMMs are not safe either.  The faulty
form itself is only reported (a later microcode / driver may fix it)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_forms_the_library_uses_are_exact_beside_bf16_mfmas(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pk_mfma_probe")
    src = os.path.join(HERE, "..", "tools", "lab", "pk_mfma_probe.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "40000", "80000", "512"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = re.findall(r"co-runner (.+?)\s*\| victim (.+?)\s*\| wrong\s+(\d+) of (\d+) results, lanes ([0-9a-f]+)", r.stdout)
    assert len(rows) == 60, r.stdout[-2000:]
    faulty = {"v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]"}
    seen = 0
    for corun, form, wrong, total, lanes in rows:
        if form in faulty:
            if int(wrong):
                seen += 1
                print(f"hardware fault observed: {form} beside {corun}: {wrong} of {total}, lanes {lanes}")
                assert int(lanes, 16) & 0x0000ffffffffffff == 0, (corun, form, lanes)   # only ever lanes 48-63
            continue
        assert int(wrong) == 0, (corun, form, wrong, total, lanes)
    print(f"{seen} (form, co-runner) pairs showed the fault on this box")

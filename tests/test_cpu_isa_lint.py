"""The built library must not contain the instruction form MI355X executes wrongly beside bf16 MFMAs (conette_amd/isa_lint.py):
a packed-fp32 arithmetic instruction whose op_sel feeds the high half of src1 to the low result.  CPU test: disassembles the in-tree
library (cross-compiled here by __graft_entry__.build())."""
import os

import pytest

from conette_amd import isa_lint

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pattern_matches_exactly_the_measured_forms():
    hit = ["v_pk_mul_f32 v[160:161], v[160:161], v[228:229] op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]",
           "v_pk_add_f32 v[8:9], v[8:9], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]"]
    safe = ["v_pk_mul_f32 v[28:29], v[28:29], v[24:25]",
            "v_pk_fma_f32 v[152:153], v[152:153], v[228:229], v[160:161] op_sel_hi:[0,1,1]",
            "v_pk_mul_f32 v[38:39], v[32:33], s[42:43] op_sel_hi:[1,0]",
            "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]",
            "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]",
            "v_pk_fma_f32 v[32:33], v[32:33], s[42:43], v[38:39] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]",
            "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]",
            "v_fma_mix_f32 v119, v105, v17, v119 op_sel:[0,1,0] op_sel_hi:[0,1,0]",
            "v_pk_mul_f16 v0, v1, v2 op_sel:[0,1]"]
    listing = ["my_kernel:"] + ["\t" + s for s in hit + safe]
    found = isa_lint.hazards_in_asm(listing)
    assert [ins.split(" op_sel")[0] for _, ins in found] == [s.split(" op_sel")[0] for s in hit]
    assert all(k == "my_kernel" for k, _ in found)


def test_built_library_is_free_of_the_hazardous_form():
    lib = os.path.join(HERE, "..", "conette-audio-captioning_amd", "libconette_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists(isa_lint.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    assert isa_lint.lint_library(lib) == []

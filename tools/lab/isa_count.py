"""Instruction-class census per basic block of one kernel in a hipcc --save-temps .s file (development aid).
   python tools/lab/isa_count.py FILE.s KERNEL_SUBSTRING [min_mfma]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 4
start = next((i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0]), None)
if start is None:
    sys.exit("kernel not found")
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
print(lines[start].split(":")[0])
label, blocks = "entry", collections.OrderedDict()
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        label = m.group(1)
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    blocks.setdefault(label, []).append(t.split()[0])
for lab, ins in blocks.items():
    c = collections.Counter()
    for i in ins:
        if i.startswith("v_mfma"): c["mfma"] += 1
        elif i.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): c["trans"] += 1
        elif i.startswith("v_pk_"): c["vpk"] += 1
        elif i.startswith(("v_accvgpr", "v_mov")): c["mov"] += 1
        elif i.startswith("v_cvt"): c["cvt"] += 1
        elif i.startswith("v_"): c["valu"] += 1
        elif i.startswith("ds_"): c["ds"] += 1
        elif i.startswith(("global_", "buffer_", "scratch_")): c["vmem"] += 1
        elif i.startswith("s_waitcnt"): c["wait"] += 1
        elif i.startswith("s_nop"): c["nop"] += 1
        elif i.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    if c["mfma"] >= min_mfma:
        print(f"  {lab:14s} n={len(ins):5d} ", dict(c))

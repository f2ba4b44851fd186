"""VGPR / AGPR / spill / LDS of every kernel in a gfx950 code object (development aid):
    hipcc <flags> --offload-device-only -c x.hip -o x.co
    clang-offload-bundler --unbundle --type=o --input=x.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=x.elf
    llvm-readelf --notes x.elf | python tools/kernel_regs.py [name-filter]"""
import re
import sys

txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt and flt not in name:
        continue
    ag = re.match(r":\s+(\d+)", blk)
    print(f"vgpr {g('vgpr_count'):>4} agpr {ag.group(1) if ag else '?':>4} spill {g('vgpr_spill_count'):>4} sgpr {g('sgpr_count'):>4} "
          f"lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}  {name[:150]}")

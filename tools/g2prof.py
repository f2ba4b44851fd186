"""Phase timing of the 128x128 encoder GEMM tiles (development aid; needs a profiling build of the library:
CN_G2_PROF=1 python conette-audio-captioning_amd/build.py --force; CN_G2_DEBUG=256 [CN_G2_DEBUG_EPI=2|4] python tools/g2prof.py)"""
import ctypes as C, os, sys, numpy as np, torch
os.environ.setdefault("CN_G2_DEBUG", "256")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
wave = torch.from_numpy(synth.synth_waveforms(64, 320000, 1234)).cuda()
eng.encode(wave); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
eng.lib.conette_debug_g2prof(buf, 1)
eng.encode(wave); torch.cuda.synchronize()
eng.lib.conette_debug_g2prof(buf, 0)
v = list(buf); n = max(v[8], 1)
names = ["prologue (addresses, first stage issue)", "first tile wait + barrier", "later tiles wait + barrier (sum)", "MFMA + LDS reads (sum)",
         "final barrier", "epilogue: activation + LDS staging (direct: all)", "epilogue: barrier", "epilogue: row-chunk stores"]
tot = sum(v[:8])
print(f"128x128 tiles: {n} blocks")
for nm, x in zip(names, v[:8]):
    print(f"{nm:44s} {x / n * 10:9.1f} ns  {100 * x / max(tot, 1):5.1f}%")
print(f"{'total per block':44s} {tot / n * 10:9.1f} ns")

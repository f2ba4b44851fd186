import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
wave = torch.from_numpy(synth.synth_waveforms(64, 320000, 1234)).cuda()
eng.encode(wave); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
eng.lib.conette_debug_dwprof(buf, 1)
eng.encode(wave); torch.cuda.synchronize()
eng.lib.conette_debug_dwprof(buf, 0)
v = list(buf)[:5]; tot = sum(v)
names = ["weights", "conv(loads+fma)+lds write", "barrier", "LN phase A", "LN phase B + stores"]
for n, x in zip(names, v): print(f"{n:28s} {x/1e6:10.1f} Mcycles  {100*x/max(tot,1):5.1f}%")

buf2 = (C.c_ulonglong * 8)()
pass
eng.encode(wave); torch.cuda.synchronize()
pass
v = list(buf2)[:7]; tot = sum(v); nblk = max(list(buf2)[7], 1)
names = ["prologue (A frags, stage0)", "wait DMA + barrier", "GEMM1", "epilogue1 (GELU, H write)", "barrier H", "GEMM2", "final epilogue"]
print("--- fused MLP")
for n, x in zip(names, v): print(f"{n:28s} {x / nblk * 10:10.1f} ns/block  {100*x/max(tot,1):5.1f}%")
print(f"per block: {tot / nblk * 10:.0f} ns over {nblk} blocks (profiling build, CN_MLP_DEBUG=C)")

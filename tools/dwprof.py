"""Phase stamps of the depthwise-conv + LN kernel (profiling build only: conette_debug_dwprof).  The fused-MLP phase
profile lives in the kernel lab (tools/lab/mlp_lab.hip, PROF = 1 instantiation of cn_mlp_rc2_ring_kernel)."""
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


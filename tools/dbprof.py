"""Phase timing of the fused decoder block kernel (development aid): CN_DB_DEBUG=1 python tools/dbprof.py [B]"""
import ctypes as C, os, sys, numpy as np, torch
os.environ.setdefault("CN_DB_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
fe, clip = eng.encode(wave)[:2]
t_audio = fe.shape[1]
lens = torch.full((B,), t_audio, dtype=torch.int32, device="cuda")
bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)].cuda()
forbid = sd["model.forbid_rep_mask"].cuda()
for _ in range(3):
    eng.decode(fe, lens, bos, forbid, 3, 3, 20)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
eng.lib.conette_debug_dbprof(buf, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); eng.decode(fe, lens, bos, forbid, 3, 3, 20); ev[1].record(); torch.cuda.synchronize()
eng.lib.conette_debug_dbprof(buf, 0)
v = list(buf)
n = max(v[9], 1)
names = ["params + P0 (x row) + issue self K/V", "wait: q|k|v GEMMs", "self-attention + issue cross K/V", "wait: out-proj GEMM",
         "LN1", "wait: cross-q GEMM", "cross-attention", "wait: cross out-proj GEMM", "LN2 + stores"]
print(f"decode {ev[0].elapsed_time(ev[1]):.3f} ms; block-kernel instances {n}")
tot = sum(v[:9])
for nm, x in zip(names, v[:9]):
    print(f"{nm:42s} {x / n * 10:9.1f} ns  {100 * x / max(tot, 1):5.1f}%")
print(f"{'total per block (row wave 0)':42s} {tot / n * 10:9.1f} ns")

sv = v[16:]
n3 = max(sv[9], 1)
names3 = ["loads issued + n_active arrives", "barrier (prefix/anc in LDS)", "masking", "softmax stats", "per-thread top-k",
          "wave top-k + barrier", "merge + barrier", "bookkeeping"]
print(f"--- search step kernel (block 0), instances {n3}")
for nm, x in zip(names3, sv[:8]):
    print(f"{nm:42s} {x / n3 * 10:9.1f} ns")
print(f"{'total':42s} {sum(sv[:8]) / n3 * 10:9.1f} ns")

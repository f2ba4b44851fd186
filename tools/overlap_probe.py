"""Development probe: do encode and decode overlap when issued on two streams?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine, make_partitioned_streams

dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16", device=dev)
B = 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).to(dev)
lens = torch.full((B,), 31, dtype=torch.int32, device=dev)
bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)].to(dev)
forbid = sd["model.forbid_rep_mask"].to(dev)
fe = [eng.decode_input_buffer(B, 31, 3, 20, slot=i) for i in range(2)]
clip = torch.empty((B, 527), device=dev)
for sl in range(2):
    for _ in range(3):
        eng.encode(wave, out=(fe[sl], clip))
        eng.decode(fe[sl], lens, bos, forbid, 3, 3, 20, clone=False, slot=sl)
torch.cuda.synchronize()
N = 10
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
enc_only = t(lambda: [eng.encode(wave, out=(fe[0], clip)) for _ in range(N)])
dec_only = t(lambda: [eng.decode(fe[1], lens, bos, forbid, 3, 3, 20, clone=False, slot=1) for _ in range(N)])
print(f"encode only {enc_only:.2f} ms, decode only {dec_only:.2f} ms")
for name, (s1, s2) in (("plain", (torch.cuda.Stream(dev), torch.cuda.Stream(dev))),
                       ("prio", (torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1))),
                       ("masked/8", make_partitioned_streams(dev, 8)), ("masked/4", make_partitioned_streams(dev, 4))):
    def both():
        for _ in range(N):
            with torch.cuda.stream(s1):
                eng.encode(wave, out=(fe[0], clip))
            with torch.cuda.stream(s2):
                eng.decode(fe[1], lens, bos, forbid, 3, 3, 20, clone=False, slot=1)
    both(); torch.cuda.synchronize()
    print(f"{name}: independent encode || decode: {t(both):.2f} ms per pair")
    def enc_masked():
        for _ in range(N):
            with torch.cuda.stream(s1):
                eng.encode(wave, out=(fe[0], clip))
    def dec_masked():
        for _ in range(N):
            with torch.cuda.stream(s2):
                eng.decode(fe[1], lens, bos, forbid, 3, 3, 20, clone=False, slot=1)
    print(f"   encode alone on s1 {t(enc_masked):.2f}  decode alone on s2 {t(dec_masked):.2f}")

# ---- two decode chains on two streams (+ optional encoder on a third) ---------------------------------
sA, sB, sE = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def two_dec():
    for _ in range(N):
        with torch.cuda.stream(sA):
            eng.decode(fe[0], lens, bos, forbid, 3, 3, 20, clone=False, slot=0)
        with torch.cuda.stream(sB):
            eng.decode(fe[1], lens, bos, forbid, 3, 3, 20, clone=False, slot=1)
try:
    two_dec(); torch.cuda.synchronize()
    print(f"two decodes on two streams: {t(two_dec):.2f} ms per pair (one decode alone {dec_only:.2f})")
except Exception as e:
    print("two-decode probe failed:", e)

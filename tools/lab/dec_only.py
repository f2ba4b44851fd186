"""Decode alone (no encode beside it), for a kernel trace: python tools/lab/dec_only.py [passes] [precision]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision=sys.argv[2] if len(sys.argv) > 2 else "bf16")
if len(sys.argv) > 3: eng.set_decode_graph(False)
dev = torch.device("cuda:0")
wave = torch.from_numpy(synth.synth_waveforms(64, 320000, 1234)).to(dev)
fe, _ = eng.encode(wave)
t = fe.shape[1]
lens = torch.full((64,), t, dtype=torch.int32, device=dev)
bos = sd["model.task_id_to_token_id"][torch.zeros(64, dtype=torch.long)].to(dev)
forbid = sd["model.forbid_rep_mask"].to(dev)
for _ in range(3): eng.decode(fe, lens, bos, forbid, 3, 3, 20, clone=False)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n): eng.decode(fe, lens, bos, forbid, 3, 3, 20, clone=False)
ev[1].record(); torch.cuda.synchronize()
print("decode alone: %.3f ms per search" % (ev[0].elapsed_time(ev[1]) / n))

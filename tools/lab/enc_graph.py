"""Lab: the encoder as a captured hipGraph against eager launches (encoder alone): python tools/lab/enc_graph.py [passes] [precision] [batch]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision=sys.argv[2] if len(sys.argv) > 2 else "bf16")
eng.set_encode_reserved_cus(24)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
fe = eng.decode_input_buffer(B, eng.lib.conette_num_audio_frames(320000), 3, 20, slot=0)
clip = torch.empty((B, 527), device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): eng.encode(wave, out=(fe, clip))
s.synchronize()
ref = fe.clone()
def timed(fn):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(s):
        ev[0].record(s)
        for _ in range(n): fn()
        ev[1].record(s)
    s.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
print("eager: %.3f ms per pass" % timed(lambda: eng.encode(wave, out=(fe, clip))))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        eng.encode(wave, out=(fe, clip))
s.synchronize()
fe.zero_()
print("graph: %.3f ms per pass" % timed(g.replay))
print("same output:", bool(torch.equal(fe, ref)))
print("eager again: %.3f ms per pass" % timed(lambda: eng.encode(wave, out=(fe, clip))))

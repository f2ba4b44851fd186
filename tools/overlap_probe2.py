"""Encode-only / decode-only / pipelined step times with the bench's buffers and streams (development aid)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
t_audio = eng.lib.conette_num_audio_frames(320000)
lens = torch.full((B,), t_audio, dtype=torch.int32, device="cuda")
bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)].cuda()
forbid = sd["model.forbid_rep_mask"].cuda()
s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
fe = [eng.decode_input_buffer(B, t_audio, 3, 20, slot=i) for i in range(2)]
clip = [torch.empty((B, 527), device="cuda") for _ in range(2)]
ev_e = [torch.cuda.Event() for _ in range(2)]; ev_d = [torch.cuda.Event() for _ in range(2)]
def enc(i):
    with torch.cuda.stream(s_enc):
        eng.encode(wave, out=(fe[i & 1], clip[i & 1])); ev_e[i & 1].record(s_enc)
def dec(i, wait=True):
    with torch.cuda.stream(s_dec):
        if wait: s_dec.wait_event(ev_e[i & 1])
        eng.decode(fe[i & 1], lens, bos, forbid, 3, 3, 20, clone=False, slot=i & 1); ev_d[i & 1].record(s_dec)
for i in range(6): enc(i); dec(i)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("encode only      %.3f ms/step" % timeit(lambda i: enc(i)))
print("decode only      %.3f ms/step" % timeit(lambda i: dec(i, wait=False)))
def both(i):
    if i >= 2:
        with torch.cuda.stream(s_enc): s_enc.wait_event(ev_d[i & 1])
    enc(i); dec(i)
print("pipelined        %.3f ms/step" % timeit(both))
def serial(i):
    enc(i); dec(i); torch.cuda.synchronize()
print("serial (sync)    %.3f ms/step" % timeit(serial))

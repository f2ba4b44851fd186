"""Encoder alone (no decode beside it), for a kernel trace: python tools/lab/enc_only.py [passes] [reserved_cus] [precision] [batch]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision=sys.argv[3] if len(sys.argv) > 3 else "bf16")
eng.set_encode_reserved_cus(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
fe = eng.decode_input_buffer(B, eng.lib.conette_num_audio_frames(320000), 3, 20, slot=0)
clip = torch.empty((B, 527), device="cuda")
for _ in range(3): eng.encode(wave, out=(fe, clip))
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n): eng.encode(wave, out=(fe, clip))
ev[1].record(); torch.cuda.synchronize()
print("encode alone: %.3f ms per pass" % (ev[0].elapsed_time(ev[1]) / n))

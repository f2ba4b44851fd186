"""Does the NUMBER of small dependent launches on another stream slow the encoder, whatever they do?
Encoder passes alone, then beside a replayed graph of N trivial dependent kernels (one 64-thread add each) per pass.
python tools/lab/launch_interference.py [passes]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
eng.set_encode_reserved_cus(24)
dev = torch.device("cuda:0")
wave = torch.from_numpy(synth.synth_waveforms(64, 320000, 1234)).to(dev)
fe = eng.decode_input_buffer(64, eng.lib.conette_num_audio_frames(320000), 3, 20, slot=0)
clip = torch.empty((64, 527), device=dev)
for _ in range(3): eng.encode(wave, out=(fe, clip))
torch.cuda.synchronize()


def graph_of(k, numel):
    """a chain of k dependent tiny kernels (x += 1 on `numel` floats), captured once"""
    x = torch.zeros(numel, device=dev)
    s = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(k): x.add_(1.0)
    return g, s


def run(g=None, s=None, reps=1):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    s_enc = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s_enc):
        ev[0].record()
    for _ in range(n):
        if g is not None:
            with torch.cuda.stream(s):
                for _ in range(reps): g.replay()
        with torch.cuda.stream(s_enc):
            eng.encode(wave, out=(fe, clip))
    with torch.cuda.stream(s_enc):
        ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


print("encoder alone: %.3f ms per pass" % run())
for k, numel in ((100, 64), (300, 64), (600, 64), (300, 1 << 16), (300, 1 << 20)):
    g, s = graph_of(k, numel)
    print("beside a chain of %4d dependent launches of %8d floats per pass: %.3f ms per pass" % (k, numel, run(g, s)))
    del g
print("encoder alone again: %.3f ms per pass" % run())

"""How sensitive is the encoder to HBM traffic from another stream?  python tools/lab/enc_with_traffic.py [passes]
One device-to-device copy of N MB (read N + write N) is enqueued on a side stream per encode pass (paced by events, so the
copy of pass i runs beside the encode of pass i); prints the encode time per pass for several N.  (The decode chains beside
an encoder move ~3.3 GB per step in bf16 -- mostly the weights its kernels re-fetch once per XCD and step.)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
eng.set_encode_reserved_cus(24)
B = 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
fe = eng.decode_input_buffer(B, eng.lib.conette_num_audio_frames(320000), 3, 20, slot=0)
clip = torch.empty((B, 527), device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for mb in (0, 400, 800, 1650, 3300, 0):
    src = torch.empty(max(mb, 1) << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(3): eng.encode(wave, out=(fe, clip))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(n):
        start = torch.cuda.Event(); start.record(main)
        if mb:
            with torch.cuda.stream(side):
                side.wait_event(start)
                dst.copy_(src, non_blocking=True)
        eng.encode(wave, out=(fe, clip))
        if mb:
            main.wait_stream(side)
    ev[1].record(); torch.cuda.synchronize()
    print("copy of %5d MB per pass beside the encoder (%.1f GB moved): encode %.3f ms per pass" % (mb, 2 * mb / 1024, ev[0].elapsed_time(ev[1]) / n))

"""Which work on another stream disturbs a log-mel front end running beside it?  (development probe behind the note in
profiles/r02_notes.md: before the front end took a compute unit's LDS for itself, the decoder's GEMM workgroups did.)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
dev = torch.device("cuda:0")
B, beam, max_pred, min_pred = 64, 3, 20, 3
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).to(dev)
t = eng.lib.conette_num_audio_frames(320000)
lens = torch.full((B,), t, dtype=torch.int32, device=dev)
bos = torch.full((B,), 1, dtype=torch.int32, device=dev)
fe0, _ = eng.encode(wave)
lm0 = eng.frontend_logmel(wave)
torch.cuda.synchronize()
caps = torch.randint(3, 5000, (B, 12), device=dev)
s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=int(os.environ.get("PROBE_PRIO", "-1")))
def other(kind):
    if kind == "decode_fused": eng.set_decode_fusion(True); eng.decode(fe0, lens, bos, None, beam, min_pred, max_pred, slot=1)
    elif kind == "decode_unfused": eng.set_decode_fusion(False); eng.decode(fe0, lens, bos, None, beam, min_pred, max_pred, slot=1)
    elif kind == "forcing_onepass": eng.set_forcing_stepwise(False); eng.forcing(fe0, lens, caps)
    elif kind == "forcing_stepwise": eng.set_forcing_stepwise(True); eng.forcing(fe0, lens, caps)
    elif kind == "greedy": eng.greedy(fe0, lens, bos, None, min_pred, max_pred)
    elif kind == "encode": eng.encode(wave, slot=1)
    elif kind.startswith("dec_b"):  # dec_b<beam>_p<max_pred>_B<batch>
        bm, mp, bb = [int(x[1:]) for x in kind.split("_")[1:]]
        eng.set_decode_fusion(True); eng.decode(fe0[:bb], lens[:bb], bos[:bb], None, bm, min(min_pred, mp), mp, slot=2)
    elif kind == "torch_small":
        a_ = getattr(other, "_a", None)
        if a_ is None:
            other._a = a_ = (torch.randn(192, 256, device=dev, dtype=torch.bfloat16), torch.randn(256, 5632, device=dev, dtype=torch.bfloat16), torch.randn(192, 5632, device=dev))
        for _ in range(150):
            y_ = a_[0] @ a_[1]
            z_ = torch.softmax(a_[2], dim=1)
    elif kind == "torch_elem":
        a_ = getattr(other, "_b", None)
        if a_ is None:
            other._b = a_ = torch.randn(64, 31, 768, device=dev)
        for _ in range(300):
            a_ = a_ * 1.0001 + 0.001
    elif kind == "matmul": (torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)).sum()
for kind in os.environ.get("PROBE_KINDS", "none,matmul,encode,decode_fused,decode_unfused,forcing_onepass,forcing_stepwise,greedy").split(","):
    bad = 0
    n = 24
    for it in range(n):
        with torch.cuda.stream(s_b):
            if kind != "none":
                other(kind)
        with torch.cuda.stream(s_a):
            lms = [eng.frontend_logmel(wave) for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(x, lm0) else 1 for x in lms)
    print(f"{kind:18s}: {bad} of {4 * n} concurrent log-mel runs differ from the solo result")

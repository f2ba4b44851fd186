"""Where are the wrong log-mel frames (lab builds of the front end beside a decode)?  frame -> (block, wave, iteration), size of the damage."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
dev = torch.device("cuda:0")
B, beam, max_pred, min_pred = 64, 3, 20, 3
NW = int(os.environ.get("FE_NW", "4"))
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).to(dev)
t = eng.lib.conette_num_audio_frames(320000)
lens = torch.full((B,), t, dtype=torch.int32, device=dev)
bos = torch.full((B,), 1, dtype=torch.int32, device=dev)
fe0, _ = eng.encode(wave)
lm0 = eng.frontend_logmel(wave)
torch.cuda.synchronize()
F = lm0.shape[-2] if lm0.ndim == 3 else lm0.shape[1]
print("logmel shape", tuple(lm0.shape))
s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=int(os.environ.get("PRIO", "-1")))
eng.set_decode_fusion(os.environ.get("FUSED", "0") == "1")
grid = min((B * F + NW - 1) // NW, 256)
for it in range(int(os.environ.get("RUNS", "6"))):
    with torch.cuda.stream(s_b):
        eng.decode(fe0, lens, bos, None, beam, min_pred, max_pred, slot=1)
    with torch.cuda.stream(s_a):
        lm = eng.frontend_logmel(wave)
    torch.cuda.synchronize()
    d = (lm != lm0).reshape(B * F, -1)
    bad = d.any(dim=1).nonzero().flatten().tolist()
    a, r = lm.reshape(B * F, -1), lm0.reshape(B * F, -1)
    tt = (a[:, 209] >= 7.0e6).nonzero().flatten().tolist()
    if tt:
        dur = {c: (int(a[c, 209].item() - 7.0e6), int(a[c, 210].item() - 7.0e6)) for c in tt}
        print("   pass durations (10 ns ticks) of wrong frames:", [dur[c] for c in tt if c in set(bad)][:30])
        print("   pass durations of a sample of right frames:  ", [dur[c] for c in tt if c not in set(bad)][:30])
    rr = (a[:, 215] >= 6.0e6).nonzero().flatten().tolist()
    if rr or os.environ.get("FE_REREAD"):
        print(f"   frames where a repeated LDS read returned something else: {len(rr)} (wrong frames {len(bad)}, both {len(set(rr) & set(bad))})")
        for c in rr[:40]:
            rf = int(a[c, 215].item() - 6.0e6)
            m = sum(int(a[c, 211 + q].item() - 5.0e6) << (16 * q) for q in range(4))
            print(f"      frame {c:6d}: read sites (0 twiddles-1, 1 stage-2 data, 2 twiddles-2, 3 stage-3 data) {[i for i in range(4) if rf >> i & 1]}, lanes {m:016x}")
    st = (a[:, 220] >= 4.0e6).nonzero().flatten().tolist()
    if st or os.environ.get("FE_STAGES"):
        print(f"   frames with a stage record: {len(st)} (wrong frames {len(bad)})")
        for c in st[:40]:
            sf = int(a[c, 220].item() - 4.0e6)
            m = sum(int(a[c, 216 + q].item() - 5.0e6) << (16 * q) for q in range(4))
            print(f"      frame {c:6d}: stages that differ between the passes (0 products, 1-3 FFT stages, 4 power): {[i for i in range(5) if sf >> i & 1]}, lanes of the first: {m:016x}")
    xl = (a[:, 221] >= 3.0e6).nonzero().flatten().tolist()
    if xl or os.environ.get("FE_XLOAD"):
        print(f"   frames whose reloaded samples / products differ: {len(xl)}; of them wrong: {len(set(xl) & set(bad))}; wrong frames NOT flagged: {len(set(bad) - set(xl))}; codes",
              {c: hex(int(a[c, 221].item() - 3.0e6)) for c in xl[:12]})
    twice = (a[:, 222] >= 2.0e6).nonzero().flatten().tolist()
    if twice or os.environ.get("FE_TWICE"):
        print(f"   frames whose two passes disagree: {len(twice)}; of them wrong in the stored (first) pass: {len(set(twice) & set(bad))}; wrong frames NOT flagged: {len(set(bad) - set(twice))}")
    coded = (a[:, 223] >= 1.0e6).nonzero().flatten().tolist()
    print(f"run {it}: {len(bad)} wrong frames of {B * F}; frames carrying a diagnostic code: {len(coded)}", {c: int(a[c, 223].item() - 1.0e6) for c in coded[:24]})
    for fr in bad[:24]:
        nb = int(d[fr].sum())
        dv = (a[fr] - r[fr]).abs()
        fin = bool(torch.isfinite(a[fr]).all())
        print(f"   frame {fr:6d} = block {(fr // NW) % grid:3d} wave {fr % NW} iter {fr // (NW * grid):3d} | {nb:3d} of {d.shape[1]} bins differ, max |diff| {float(dv.max()):.4g}, "
              f"first bin {int(d[fr].nonzero()[0])}, finite {fin}")

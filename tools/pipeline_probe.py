"""Does a step give the same result alone and next to other work?  (development probe)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
eng.set_encode_reserved_cus(int(os.environ.get("CN_ENC_RESERVE", "16")))
dev = torch.device("cuda:0")
B, beam, max_pred, min_pred = 64, 3, 20, 3
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).to(dev)
t = eng.lib.conette_num_audio_frames(320000)
lens = torch.full((B,), t, dtype=torch.int32, device=dev)
bos = torch.full((B,), 1, dtype=torch.int32, device=dev)
def solo(slot):
    fe = eng.decode_input_buffer(B, t, beam, max_pred, slot=slot)
    clip = torch.empty((B, 527), device=dev)
    eng.encode(wave, out=(fe, clip))
    out = eng.decode(fe, lens, bos, None, beam, min_pred, max_pred, clone=True, slot=slot)
    torch.cuda.synchronize()
    return fe.clone(), out["best_preds"], out["best_lprobs"]
fe0, p0, l0 = solo(0)
for k in range(3):
    fe1, p1, l1 = solo(0)
    print("solo repeat", k, "fe equal", torch.equal(fe0, fe1), "preds equal", torch.equal(p0, p1), "lprobs equal", torch.equal(l0, l1))
# overlapped: encode on one stream while a decode of the SAME input runs on another
s_enc, s_dec = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
fe_a = eng.decode_input_buffer(B, t, beam, max_pred, slot=1)
fe_b = eng.decode_input_buffer(B, t, beam, max_pred, slot=2)
clip = torch.empty((B, 527), device=dev)
eng.encode(wave, out=(fe_a, clip)); torch.cuda.synchronize()
for k in range(4):
    with torch.cuda.stream(s_dec):
        out = eng.decode(fe_a, lens, bos, None, beam, min_pred, max_pred, clone=True, slot=1)
    with torch.cuda.stream(s_enc):
        eng.encode(wave, out=(fe_b, clip), slot=1)
    torch.cuda.synchronize()
    print("overlap", k, "decode preds equal solo", torch.equal(out["best_preds"], p0), "lprobs", torch.equal(out["best_lprobs"], l0),
          "| encode fe equal solo", torch.equal(fe_b, fe0), "max|dfe|", float((fe_b - fe0).abs().max()))

# the bench's two-slot pipeline, every step checked against the solo pass
n_slot = int(os.environ.get("PROBE_SLOTS", "2"))
n_dec = max(1, n_slot - 1)
s_decs = [torch.cuda.Stream(dev, priority=-1) for _ in range(n_dec)]
sl = [dict(fe=eng.decode_input_buffer(B, t, beam, max_pred, slot=10 + k), clip=torch.empty((B, 527), device=dev),
           enc_done=torch.cuda.Event(), dec_done=torch.cuda.Event()) for k in range(n_slot)]
keep = []
for i in range(int(os.environ.get("PROBE_STEPS", "40"))):
    s = sl[i % n_slot]
    sd_ = s_decs[i % n_dec]
    with torch.cuda.stream(s_enc):
        if i >= n_slot:
            s_enc.wait_event(s["dec_done"])
        eng.encode(wave, out=(s["fe"], s["clip"]), slot=i & 1)
        s["enc_done"].record(s_enc)
    with torch.cuda.stream(sd_):
        sd_.wait_event(s["enc_done"])
        out = eng.decode(s["fe"], lens, bos, None, beam, min_pred, max_pred, clone=True, slot=10 + i % n_slot)
        fe_c = s["fe"].clone()
        s["dec_done"].record(sd_)
    keep.append((out["best_preds"], out["best_lprobs"], fe_c))
torch.cuda.synchronize()
bad = [(i, bool(torch.equal(f, fe0)), int((p != p0).any(dim=1).sum()), int((l != l0).sum())) for i, (p, l, f) in enumerate(keep)
       if not (torch.equal(p, p0) and torch.equal(l, l0))]
print("pipeline steps differing from solo (step, fe equal, rows with other ids, lprob diffs):", bad[:12], "count", len(bad))

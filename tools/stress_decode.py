"""Repeatability stress for the hand-counted vmcnt kernels (dec_block.h, dec_ffn.h): the same decode many times, under
concurrent encoder load on another stream, must reproduce bit-identical outputs; different batch sizes must agree
row by row.  python tools/stress_decode.py [iters]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import conette_amd
from conette_amd import synth
from conette_amd.engine import Engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict().items()}
eng = Engine(sd, precision="bf16")
eng.set_decode_graph(False)   # eager launches: different timing relative to the encoder stream every time
B = 64
wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 77)).cuda()
fe, _ = eng.encode(wave)
lens = torch.full((B,), fe.shape[1], dtype=torch.int32)
bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)]
forbid = sd["model.forbid_rep_mask"]
ref = eng.decode(fe, lens, bos, forbid, 3, 3, 20, want_trace=True)
s2 = torch.cuda.Stream()
bad = 0
for it in range(iters):
    with torch.cuda.stream(s2):
        eng.encode(wave[: 8 + (it % 5) * 8].contiguous())     # concurrent load, varying shape
    out = eng.decode(fe, lens, bos, forbid, 3, 3, 20, want_trace=True)
    for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs", "trace_val"):
        if not torch.equal(out[k], ref[k]):
            bad += 1
            print("MISMATCH iter", it, k, float((out[k].float() - ref[k].float()).abs().max()))
            break
torch.cuda.synchronize()
sub = eng.decode(fe[:13].contiguous(), lens[:13], bos[:13], forbid, 3, 3, 20)
ok_sub = torch.equal(sub["best_preds"][:, : ref["best_preds"].shape[1]], ref["best_preds"][:13]) and torch.equal(sub["best_lprobs"], ref["best_lprobs"][:13])
print(f"iterations {iters}, mismatching {bad}, 13-clip sub-batch identical: {ok_sub}")
sys.exit(1 if bad or not ok_sub else 0)

"""Stage-by-stage GPU-vs-oracle diagnostic (development aid; writes a text report to stdout)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conette_amd  # noqa
from conette_amd import synth
from conette_amd.engine import Engine
from oracle import cpu_ref as O


def err(name, got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    d = (got - ref).abs()
    rel = d.max() / (ref.abs().max() + 1e-12)
    print(f"  {name:18s} shape {tuple(got.shape)!s:24s} max|d| {d.max():.3e} mean|d| {d.mean():.3e} "
          f"ref_rms {ref.pow(2).mean().sqrt():.3e} rel_max {rel:.3e} nan {int(torch.isnan(got).sum())}", flush=True)


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10
    T0 = time.time()
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    import torch as _t
    print(f"imports done t={time.time()-T0:.1f}s", flush=True)
    sd_np = synth.synth_state_dict()
    print(f"synth done t={time.time()-T0:.1f}s", flush=True)
    sd = O.to_torch(sd_np)
    cfg = synth.synth_config_dict()
    L = int(secs * 32000)
    lengths = [L] * B
    if B > 1:
        lengths[-1] = L // 2
    wav = torch.from_numpy(synth.synth_waveforms(B, L, 1234, lengths=lengths))
    t0 = time.time()
    eng = Engine(sd, precision=prec)
    print(f"[{prec}] engine created in {time.time()-t0:.2f}s", flush=True)
    taps = {}
    with torch.no_grad():
        ref = O.convnext_encode(sd, wav, torch.tensor([[n] for n in lengths]), taps)
    print(f"oracle encode done t={time.time()-T0:.1f}s", flush=True)
    x = wav.cuda()
    fe, clip, gt = eng.encode(x, taps=True)
    torch.cuda.synchronize()
    err("logmel", gt["logmel"], taps["logmel"][:, 0])
    err("stem", gt["stem"].permute(0, 3, 1, 2), taps["stem"])
    for i in range(4):
        if i > 0:
            err(f"down{i}", gt[f"down{i}"].permute(0, 3, 1, 2), taps[f"down{i}"])
        err(f"stage{i}_block0", gt[f"stage{i}_block0"].permute(0, 3, 1, 2), taps[f"stage{i}_block0"])
        err(f"stage{i}", gt[f"stage{i}"].permute(0, 3, 1, 2), taps[f"stage{i}"])
    err("frame_embs", fe, ref["frame_embs"].transpose(1, 2))
    err("clip_probs", clip, ref["clipwise_output"])
    # decoder: feed the ORACLE's frame_embs so that errors do not compound
    fe_ref = ref["frame_embs"].transpose(1, 2).contiguous()
    lens = ref["frame_embs_lens"]
    mem, mask = O.encode_audio(sd, fe_ref, torch.stack([torch.full_like(lens, 768), lens], 1))
    bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)]
    for beam in (1, 3):
        trace = []
        with torch.no_grad():
            rp, rl, rmp, rml = O.generate(sd, mem, mask, bos, vocab_size=eng.vocab_size, beam_size=beam,
                                          forbid_rep_mask=sd["model.forbid_rep_mask"], trace=trace)
            lg0 = O.decoder_forward(sd, mem.repeat_interleave(beam, 0).permute(2, 0, 1).contiguous(),
                                    mask.repeat_interleave(beam, 0), bos.repeat_interleave(beam)[None])[-1]
        print(f"oracle generate beam {beam} done t={time.time()-T0:.1f}s", flush=True)
        out = eng.decode(fe_ref.cuda(), lens, bos, sd["model.forbid_rep_mask"], beam, 3, 20, want_step0_logits=True)
        torch.cuda.synchronize()
        ps, bm = out["sizes"].tolist()
        print(f" beam {beam}: sizes {ps},{bm} ref {rmp.shape[-1]},{rp.shape[-1]}")
        err("step0_logits", out["step0_logits"], lg0)
        gp = out["best_preds"][:, :bm].cpu().long()
        print("  preds equal:", gp.shape == rp.shape and bool((gp == rp).all()))
        print("  gpu :", gp.tolist())
        print("  ref :", rp.tolist())
        err("lprobs", out["best_lprobs"], rl)
        err("mult_lprobs", out["mult_lprobs"], rml)
        gm = out["mult_preds"][:, :, :ps].cpu().long()
        print("  mult_preds equal:", gm.shape == rmp.shape and bool((gm == rmp).all()))
        m = [c["margin"] for st in trace for c in st]
        print("  min margin %.4g" % min(m))
    # timing
    for _ in range(2):
        eng.encode(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        eng.encode(x)
    torch.cuda.synchronize()
    print(f" encode B={B}: {(time.time()-t0)/5*1e3:.2f} ms")
    t0 = time.time()
    for _ in range(3):
        eng.decode(fe_ref.cuda(), lens, bos, sd["model.forbid_rep_mask"], 3, 3, 20)
    torch.cuda.synchronize()
    print(f" decode B={B} beam 3: {(time.time()-t0)/3*1e3:.2f} ms")


if __name__ == "__main__":
    main()

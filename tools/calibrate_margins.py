"""Calibration of the id certificate (precision "certified", round 6): how large can a top-k margin of a 16-bit search be while
the exact search decides otherwise?

For every base precision, both synthetic checkpoints and beam 1 / 3: N clips run through the base precision (encoder + decoder,
with the device-side margins of conette_decode) and through the exact precision, both with per-call traces.  For a clip, the
searches walk the SAME trajectory up to the first call d whose (parents, tokens) differ; at d both saw the same prefixes, so the
base precision's own margin m[d] is a margin that did NOT protect the decision.  A tolerance t(i) = a + b (i + 1) certifies ids
iff it exceeds every such m[d]: the tool prints them (per step bucket), the candidate-value errors along shared trajectories,
what ``engine.CERT_TOL`` flags (recompute fraction) and whether it misses any diverging clip.

    python tools/calibrate_margins.py [--clips 512] [--out profiles/r06_margin_calibration.txt]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def first_divergence(sel_a, sel_b):
    """sel (max_pred, B, beam, 2) -> (B,) first step whose picks differ (max_pred where none does)"""
    mp = sel_a.shape[0]
    diff = (sel_a != sel_b).flatten(2).any(dim=2)            # (mp, B)
    step = torch.arange(mp, device=sel_a.device)[:, None].expand_as(diff)
    return torch.where(diff, step, mp).amin(dim=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=512)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--beams", default="1,3", help="beam sizes, comma separated")
    ap.add_argument("--bases", default="f16,mixed16,bf16+f16dec,bf16")
    ap.add_argument("--seed0", type=int, default=900000, help="clip i is generated from seed0 + i; lengths from default_rng(seed0 % 1000 + 7)")
    args = ap.parse_args()
    from conette_amd import synth
    from conette_amd import engine as E

    dev = torch.device("cuda:0")
    lines = []

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        lines.append(s)

    n, bsz = args.clips, args.batch
    L = 10 * 32000
    rng = np.random.default_rng(7 if args.seed0 == 900000 else args.seed0 % 1000 + 7)
    # half of the clips full length (the benchmark's workload), half ragged 1-10 s
    lengths = [L if i % 2 == 0 else int(rng.integers(32000, L)) for i in range(n)]
    wave_all = torch.from_numpy(synth.synth_waveforms(n, L, args.seed0, lengths=lengths))
    max_pred, min_pred = 20, 3
    say(f"# margin calibration: {n} clips (half 10 s, half 1-10 s; seeds {args.seed0} + i), max_pred {max_pred}, min_pred {min_pred}")
    for recipe in ("default", "peaked"):
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(recipe=recipe).items()}
        forbid = sd["model.forbid_rep_mask"].to(torch.bool).to(dev)
        vocab = int(sd["model.decoder.classifier.weight"].shape[0])
        for base in args.bases.split(","):
            eng = E.Engine(sd, precision=f"certified:{base}", device=dev)
            for beam in [int(v) for v in args.beams.split(",")]:
                a, b, c = E.CERT_TOL[base][E.cert_kind(beam)]
                rel = {"flagged": 0, "wrong": 0, "missed": 0}
                unprot = []      # (step, base margin) at the first diverging call
                unprot_final = []
                val_err = np.zeros(max_pred)
                flagged = missed = diverged = 0
                for s0 in range(0, n, bsz):
                    w = wave_all[s0:s0 + bsz].to(dev)
                    nb = w.shape[0]
                    lens_s = torch.tensor(lengths[s0:s0 + nb])
                    t = eng.lib.conette_num_audio_frames(L)
                    from conette_amd.preprocessor import frame_embs_lens
                    flens = frame_embs_lens(lens_s, L, t)
                    bos = torch.full((nb,), vocab - 7, dtype=torch.int32)
                    fe, _ = eng.encode(w)
                    r = eng.decode(fe, flens, bos, forbid, beam, min_pred, max_pred, want_trace=True, want_margins=True)
                    fx, _ = eng.encode(w, exact=True)
                    x = eng.decode(fx, flens, bos, forbid, beam, min_pred, max_pred, want_trace=True, exact=True)
                    d = first_divergence(r["trace_sel"], x["trace_sel"])        # (nb,)
                    m2 = r["margins"]                                          # (nb, 2, max_pred + 1)
                    m = torch.minimum(m2[:, 0], m2[:, 1])                       # the strict certificate's margin per step
                    same_final = (r["best_preds"] == x["best_preds"]).all(dim=1)
                    flag = eng.uncertified(m2, r["best_lprobs"], beam=beam, order=True)
                    # the relaxed policy ("certified-best"): membership plane only; wrong = another best caption, or another SET of hypotheses
                    flag_b = eng.uncertified(m2, r["best_lprobs"], beam=beam, order=False)
                    srt = lambda t: torch.sort(t.to(torch.int64).mul(torch.tensor([7919 ** j % 1000003 for j in range(t.shape[-1])], device=t.device)).sum(-1), dim=1).values
                    same_set = (srt(r["mult_preds"]) == srt(x["mult_preds"])).all(dim=1)
                    wrong_b = ~(same_final & same_set)
                    rel["flagged"] += int(flag_b.sum()); rel["wrong"] += int(wrong_b.sum()); rel["missed"] += int((wrong_b & ~flag_b).sum())
                    for i in range(nb):
                        di = int(d[i])
                        if di < max_pred:
                            unprot.append((di, float(m[i, di])))
                        elif not bool(same_final[i]):
                            unprot_final.append(float(m2[i, 0, max_pred]))
                        wrong = di < max_pred or not bool(same_final[i])
                        diverged += wrong
                        flagged += bool(flag[i])
                        missed += wrong and not bool(flag[i])
                    # candidate-value error along shared trajectories
                    dv = (r["trace_val"] - x["trace_val"]).abs()                 # (mp, nb, beam)
                    live = (x["trace_sel"][..., 1] >= 0)
                    step = torch.arange(max_pred, device=dev)[:, None, None]
                    shared = live & (step <= d[None, :, None])
                    dv = torch.where(shared & torch.isfinite(dv), dv, torch.zeros_like(dv))
                    val_err = np.maximum(val_err, dv.flatten(1).amax(dim=1).cpu().numpy())
                um = np.zeros(max_pred)
                for di, mv in unprot:
                    um[di] = max(um[di], mv)
                say(f"{recipe:8s} {base:12s} beam {beam}: diverging clips {diverged}/{n}  flagged by CERT_TOL {flagged}/{n} "
                    f"({flagged / n:.3f})  MISSED {missed}")
                say("    largest unprotecting margin by step :", " ".join(f"{v:.4f}" for v in um))
                say(f"    tolerance a + b (i + 1), final c      : a = {a}, b = {b}, c = {c}")
                say("    max |candidate value error| by step :", " ".join(f"{v:.4f}" for v in val_err))
                say(f"    relaxed policy (membership plane only: best caption + set of hypotheses): flagged {rel['flagged']}/{n} "
                    f"({rel['flagged'] / n:.3f}), clips with another best caption or hypothesis set {rel['wrong']}, MISSED {rel['missed']}")
                if unprot_final:
                    say(f"    final best-beam choice: largest unprotecting margin {max(unprot_final):.5f} (tolerance {c})")
            del eng
            torch.cuda.synchronize()
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

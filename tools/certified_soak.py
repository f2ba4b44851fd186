"""Soak of the id certificate: N clips (default 4096; half 10 s, half 1-10 s, seeds disjoint from the tests' and the calibration's) through
precision "certified" (strict) and "certified-best" for each base precision, both synthetic checkpoints, greedy and beam 3, against the
exact precision on the same waveforms.  Strict: best_preds, mult_preds (in order) and sizes must be identical on EVERY clip; best: best_preds
identical and mult_preds the same set of hypotheses.  Prints mismatches (expected: 0) and recompute fractions.

    python tools/certified_soak.py [--clips 4096] [--bases mixed16,f16] [--out profiles/r06_certified_soak.txt]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--bases", default="mixed16,f16")
    ap.add_argument("--out", default=None)
    ap.add_argument("--beams", default="1,3", help="beam sizes, comma separated")
    ap.add_argument("--seed0", type=int, default=3000000)
    args = ap.parse_args()
    import conette_amd  # noqa: F401
    from conette_amd import synth
    from conette_amd import engine as E
    from conette_amd.preprocessor import frame_embs_lens

    dev = torch.device("cuda:0")
    lines = []

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        lines.append(s)

    n, bsz, L = args.clips, args.batch, 320000
    rng = np.random.default_rng(2026 + args.seed0 % 1000003)
    lengths = [L if i % 2 == 0 else int(rng.integers(32000, L)) for i in range(n)]
    say(f"# certified soak: {n} clips (half 10 s, half 1-10 s; seeds {args.seed0} + i), max_pred 20, min_pred 3, against the exact precision")
    # (the waveforms live in HBM -- 1.28 MB per clip, 288 GB there -- not in host memory: one batch at a time is generated and moved)
    waves = [torch.from_numpy(synth.synth_waveforms(min(bsz, n - s0), L, args.seed0 + s0, lengths=lengths[s0:s0 + bsz])).to(dev) for s0 in range(0, n, bsz)]
    max_pred, min_pred = 20, 3
    hyp_hash = None
    for recipe in ("default", "peaked"):
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(recipe=recipe).items()}
        forbid = sd["model.forbid_rep_mask"].to(torch.bool).to(dev)
        vocab = int(sd["model.decoder.classifier.weight"].shape[0])
        if hyp_hash is None:
            hyp_hash = torch.tensor([pow(7919, j, 1000003) for j in range(max_pred)], device=dev)
        for base in args.bases.split(","):
            for policy in ("certified", "certified-best"):
                eng = E.Engine(sd, precision=f"{policy}:{base}", device=dev)
                t = eng.lib.conette_num_audio_frames(L)
                for beam in [int(v) for v in args.beams.split(",")]:
                    if policy == "certified-best" and beam == 1:
                        continue            # (greedy has no pick order: the two policies coincide)
                    bad_best = bad_mult = bad_set = rec = 0
                    for bi, w in enumerate(waves):
                        nb = w.shape[0]
                        flens = frame_embs_lens(torch.tensor(lengths[bi * bsz: bi * bsz + nb]), L, t)
                        bos = torch.full((nb,), vocab - 7, dtype=torch.int32)
                        fe, _ = eng.encode(w)
                        c = eng.generate_certified(w, fe, flens, bos, forbid, beam, min_pred, max_pred)
                        fx, _ = eng.encode(w, exact=True)
                        x = eng.decode(fx, flens, bos, forbid, beam, min_pred, max_pred, exact=True)
                        rec += int(c["recomputed"].sum())
                        bad_best += int((c["best_preds"] != x["best_preds"]).any(dim=1).sum())
                        bad_mult += int((c["mult_preds"] != x["mult_preds"]).flatten(1).any(dim=1).sum())
                        hs = lambda m: torch.sort((m.to(torch.int64) * hyp_hash).sum(-1), dim=1).values
                        bad_set += int((hs(c["mult_preds"]) != hs(x["mult_preds"])).any(dim=1).sum())
                    ok = (bad_best == 0 and bad_mult == 0) if policy == "certified" else (bad_best == 0 and bad_set == 0)
                    say(f"{recipe:8s} {policy + ':' + base:24s} beam {beam}: recomputed {rec}/{n} ({rec / n:.3f})  other best caption {bad_best}  "
                        f"other slot table {bad_mult}  other hypothesis set {bad_set}  -> {'OK' if ok else 'MISMATCH'}")
                del eng
                torch.cuda.synchronize()
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

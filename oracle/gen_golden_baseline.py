"""ORACLE tooling: golden fixture of the reference's BaselinePLM (pl_modules/baseline.py) -- SURVEY.md section 8 (f) 4.

Runs ONLY in the build container: imports the reference's own BaselinePLM through oracle/refshim.py, loads the synthetic
BaselinePLM-layout checkpoint (conette_amd.synth.synth_baseline_state_dict: no encoder, no task tokens) with the module's
own ``load_state_dict`` (which builds the model from the tokenizer state, base.py:76-103) and records, on the frame embeddings
of a committed CoNeTTE fixture used as the precomputed features such a model consumes, the outputs of its three decode
methods.  Data only.

    python -m oracle.gen_golden_baseline      # writes tests/golden/baseline/baseline_b4.npz
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conette_amd  # noqa: E402,F401
from conette_amd import synth  # noqa: E402
from oracle import refshim  # noqa: E402


def main() -> None:
    refshim.install()
    from conette.pl_modules.baseline import BaselinePLM
    import conette.nn.decoding.beam as beam_mod

    torch.manual_seed(0)
    sd_np = synth.synth_baseline_state_dict()
    sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sd_np.items()}
    plm = BaselinePLM(beam_size=3, max_pred_size=20)
    print(plm.load_state_dict(sd, strict=True))
    plm.eval()
    src = np.load(os.path.join(ROOT, "tests", "golden", "b4_10s_beam3_clotho.npz"))
    fe = torch.from_numpy(src["frame_embs"])                       # (4, 31, 768)
    shape = torch.from_numpy(src["audio_shape"]).clone()           # (4, 2): [768, T]
    shape[1, 1] = 17                                                # ragged: two clips shorter than the padded length
    shape[3, 1] = 9
    batch = {"audio": fe[:, None], "audio_shape": shape}           # (B, 1, T, 768) as the HDF datasets deliver it

    trace = []
    orig = beam_mod._select_k_next_toks

    def rec(logits_i, prev_sum_lprobs, is_first):
        out = orig(logits_i=logits_i, prev_sum_lprobs=prev_sum_lprobs, is_first=is_first)
        k = logits_i.shape[0]
        lg = logits_i[0:1] if is_first else logits_i
        cand = torch.log_softmax(lg, dim=1) if is_first else prev_sum_lprobs[:, None] + torch.log_softmax(lg, dim=1)
        top = torch.topk(cand.reshape(-1), min(k + 1, cand.numel())).values
        trace.append((out[0].tolist(), out[1].tolist(), out[2].tolist(), float(top[k - 1] - top[k]) if top.numel() > k else float("inf")))
        return out

    beam_mod._select_k_next_toks = rec
    with torch.no_grad():
        gen = plm(batch, "generate")
        beam_mod._select_k_next_toks = orig
        greedy = plm(batch, "greedy")                               # (B, V, steps) masked logits (greedy.py)
        caps = torch.nn.functional.pad(gen["preds"], (1, 0), value=plm.bos_id)   # <bos> + best caption (+ <eos> / pad)
        forcing = plm.decode_audio(plm.encode_audio(batch["audio"], batch["audio_shape"]), "forcing", caps_in=caps[:, :-1])
    rec_ = dict(
        src=np.asarray("b4_10s_beam3_clotho"), audio_shape=shape.numpy(),
        preds=gen["preds"].numpy(), lprobs=gen["lprobs"].numpy(), mult_preds=gen["mult_preds"].numpy(),
        mult_lprobs=gen["mult_lprobs"].numpy(), cands=np.asarray(json.dumps(gen["cands"])),
        trace_parent=np.asarray(json.dumps([t[0] for t in trace])), trace_token=np.asarray(json.dumps([t[1] for t in trace])),
        trace_sum=np.asarray(json.dumps([t[2] for t in trace])), trace_margin=np.asarray([t[3] for t in trace]),
        greedy_ids=greedy.argmax(dim=1).numpy(), greedy_logits_sub=greedy[:, ::37, :].numpy(),
        caps_in=caps[:, :-1].numpy(), forcing_logits_sub=forcing[:, ::37, :].numpy(),
    )
    out_dir = os.path.join(ROOT, "tests", "golden", "baseline")
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "baseline_b4.npz"), **rec_)
    print("preds", gen["preds"].tolist(), "lprobs", gen["lprobs"].numpy().round(4), "greedy", greedy.shape, "forcing", forcing.shape,
          "min margin %.4g" % min(t[3] for t in trace))


if __name__ == "__main__":
    main()

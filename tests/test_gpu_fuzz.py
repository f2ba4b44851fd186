"""GPU: seeded random configurations of the operator API against the CPU oracle (ids exact, scores 2e-4).

The golden scenarios pin nine hand-picked shapes; this draws others -- batch 1-6, ragged lengths from the 0.24 s minimum to 20 s
(odd sample counts included), sampling rates that need the resampler, beam 1-8, min / max caption length, every forbid_rep_mode,
per-clip tasks -- and requires the "exact" precision (fp16 hi / lo operand pairs), the "certified" precision (round 6: the library's
default) and, for the smaller draws, the fp32 precision to return the oracle's token ids and candidates (exact / fp32: scores too).  Seeds are fixed: a failure is reproducible by its case number."""
import os

import numpy as np
import pytest
import torch

from conette_amd import synth

pytestmark = pytest.mark.gpu
TAGS = {i: f"tag{i}" for i in range(527)}


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    from conette_amd import CoNeTTEModel
    d = synth.write_pretrained_dir(str(tmp_path_factory.mktemp("conette_fuzz")))
    mk = lambda p: CoNeTTEModel.from_pretrained(d, precision=p, offline=True, audioset_idx_to_name=TAGS, stopwords=synth.synth_stopwords())
    return {"exact": mk("exact"), "fp32": mk("fp32"), "certified": mk("certified")}


def _draw(case):
    rng = np.random.default_rng(9000 + case)
    b = int(rng.integers(1, 7))
    sr = int(rng.choice([32000, 32000, 16000, 44100, 48000]))
    lens = []
    for _ in range(b):
        kind = rng.integers(0, 4)
        sec = rng.uniform(0.25, 1.0) if kind == 0 else (rng.uniform(1.0, 10.0) if kind < 3 else rng.uniform(10.0, 20.0))
        lens.append(max(int(sec * sr) | int(rng.integers(0, 2)), int(0.25 * sr) + 1))   # odd sample counts half of the time
    beam = int(rng.choice([1, 2, 3, 3, 4, 5, 8]))
    min_pred = int(rng.integers(0, 5))
    max_pred = int(rng.integers(max(min_pred + 1, 5), 26))
    mode = [None, "none", "all", "content_words"][int(rng.integers(0, 4))]
    return b, sr, lens, beam, min_pred, max_pred, mode, rng


@pytest.mark.parametrize("case", range(24))
def test_random_configuration_against_the_oracle(case, models, synth_weights, synth_cfg):
    from oracle import cpu_ref as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    b, sr, lens, beam, min_pred, max_pred, mode, rng = _draw(case)
    tasks = [str(rng.choice(list(synth_cfg["task_names"]))) for _ in range(b)]
    wav = synth.synth_waveforms(b, max(lens), 31000 + case, lengths=lens)
    x = [torch.from_numpy(wav[i : i + 1, :n].copy()) for i, n in enumerate(lens)]
    kw = dict(task=tasks, beam_size=beam, min_pred_size=min_pred, max_pred_size=max_pred)
    if mode is not None:
        kw["forbid_rep_mode"] = mode
    with torch.no_grad():
        ref = O.model_forward(synth_weights, synth_cfg, x, sr=[sr] * b, **kw)
    # "certified" (round 6: the fp16 pipeline + the id certificate; uncertified clips re-run through the exact context) is held to
    # the same IDS as the exact precision; its scores / tag probabilities are fp16-pipeline values wherever the clip was certified
    precs = ["exact", "certified"] + (["fp32"] if sum(lens) / sr < 25 else [])
    for prec in precs:
        out = models[prec](x, sr=[sr] * b, **kw)
        what = (case, prec, b, sr, lens, beam, min_pred, max_pred, mode, tasks)
        lp_tol, tag_tol = (0.05, 0.006) if prec == "certified" else (2e-4, 1e-4)
        assert out["preds"].cpu().tolist() == ref["preds"].tolist(), what
        assert out["mult_preds"].cpu().tolist() == ref["mult_preds"].tolist(), what
        assert out["cands"] == ref["cands"] and out["tasks"] == ref["tasks"], what
        np.testing.assert_allclose(out["lprobs"].cpu().numpy(), ref["lprobs"].numpy(), atol=lp_tol, err_msg=str(what))
        np.testing.assert_allclose(out["mult_lprobs"].cpu().numpy(), ref["mult_lprobs"].numpy(), atol=lp_tol, err_msg=str(what))
        np.testing.assert_allclose(out["tags_probs"].cpu().numpy(), ref["tags_probs"].numpy(), rtol=1e-3, atol=tag_tol, err_msg=str(what))

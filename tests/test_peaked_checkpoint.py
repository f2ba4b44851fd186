"""The "peaked" synthetic checkpoint (conette_amd.synth PEAKED, round 5): few candidates far above a noise floor, like a trained
captioner, where the default recipe's Gaussian logits put every top-k decision within a few percent of a logit of its runner-up.

CPU: the oracle reproduces the fixtures the imported reference produced with this checkpoint (oracle/gen_golden.py, tests/golden/peaked/).
GPU: on these fixtures the 16-bit precisions are held to the reference's TOKEN IDS -- every top-k call whose reference margin
exceeds 0.25 (bf16) / 0.03 (f16) must pick the reference's parents and tokens, at least 80 % of the greedy calls (and the measured
share of the beam-3 calls: hypotheses of a beam search are near each other in any checkpoint) must be above that margin, and a clip
whose calls all are must return the reference's caption from the waveform (with the default recipe 0-35 % of the
calls were above the margin, so the id assertions of test_gpu_parity.py say little about bf16)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from tests import golden_util as G

PK_DIR = os.path.join(G.GOLDEN, "peaked")
PK = sorted(f[:-4] for f in os.listdir(PK_DIR) if f.endswith(".npz"))
TOL = {"bf16": 0.25, "f16": 0.03125, "fp32": 5e-4, "exact": 5e-4}
# share of a fixture's top-k calls whose effective reference margin (gap to the first rejected candidate and gaps between
# consecutive picks) exceeds TOL["bf16"] = 0.25: measured 0.93 / 0.89 (greedy: 57 / 47 calls), 0.68 / 0.46 (beam 3: 93 / 50 calls -- the
# hypotheses of a beam search are near each other in any checkpoint; at the f16 tolerance 0.03 the shares are 0.98 / 0.98 / 0.96 / 0.82).
# The default recipe's fixtures: 0-35 %.  Held to these floors:
ABOVE_FLOOR = {"pk_b8_10s_beam1_clotho": 0.8, "pk_b4_mixed_beam1_tasks": 0.8, "pk_b8_10s_beam3_clotho": 0.6, "pk_b4_mixed_beam3_tasks": 0.4}


def _load(name):
    return np.load(os.path.join(PK_DIR, name + ".npz"))


@pytest.fixture(scope="module")
def peaked_weights():
    from conette_amd import synth
    return O.to_torch(synth.synth_state_dict(recipe="peaked"))


def _calls(g, beam, max_pred):
    par, tok, sums, margin = G.trace_of(g)
    bsz = len(g["lengths"])
    k = [beam] * bsz
    calls, ci = [], 0
    for step in range(max_pred):
        for clip in range(bsz):
            if k[clip] == 0:
                continue
            eff = min([float(margin[ci])] + [sums[ci][i] - sums[ci][i + 1] for i in range(len(par[ci]) - 1)])
            calls.append((step, clip, par[ci], tok[ci], sums[ci], eff))
            k[clip] -= k[clip] if step == max_pred - 1 else sum(1 for t in tok[ci] if t == 2)
            ci += 1
        if ci == len(par):
            break
    assert ci == len(par)
    return calls


@pytest.mark.parametrize("name", PK)
def test_oracle_matches_reference_fixture_peaked(name, peaked_weights, synth_cfg):
    g = _load(name)
    x, kw = G.inputs(g)
    trace = []
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        out = O.model_forward(peaked_weights, synth_cfg, x, sr=32000, trace=trace, **kw)
    assert out["preds"].tolist() == g["preds"].tolist()
    assert out["mult_preds"].tolist() == g["mult_preds"].tolist()
    np.testing.assert_allclose(out["lprobs"].numpy(), g["lprobs"], atol=1e-5)
    par, tok, _, _ = G.trace_of(g)
    flat = [c for st in trace for c in st]
    assert [c["parent"] for c in flat] == par and [c["token"] for c in flat] == tok
    assert out["cands"] == json.loads(str(g["cands"]))
    # the recipe's point: most decisions are far from a tie
    beam = kw.get("beam_size", synth_cfg["beam_size"])
    calls = _calls(g, beam, kw.get("max_pred_size", synth_cfg["max_pred_size"]))
    share = np.mean([c[5] > TOL["bf16"] for c in calls])
    assert share >= ABOVE_FLOOR[name], (name, share)


@pytest.fixture(scope="module")
def peaked_engines(peaked_weights):
    from conette_amd.engine import Engine
    return {p: Engine(peaked_weights, precision=p) for p in ("bf16", "f16", "exact")}


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16", "f16", "exact"])
@pytest.mark.parametrize("name", PK)
def test_ids_match_reference_on_peaked_checkpoint(name, prec, peaked_engines, peaked_weights, synth_cfg):
    """conette_decode from the reference's own frame embeddings: every call above the margin picks the reference's (parents, tokens)."""
    g = _load(name)
    kw = json.loads(str(g["kw"]))
    eng = peaked_engines[prec]
    beam = kw.get("beam_size", synth_cfg["beam_size"])
    min_pred = kw.get("min_pred_size", synth_cfg["min_pred_size"])
    max_pred = kw.get("max_pred_size", synth_cfg["max_pred_size"])
    tasks = json.loads(str(g["tasks"]))
    names = list(synth_cfg["task_names"])
    bos = peaked_weights["model.task_id_to_token_id"][torch.as_tensor([names.index(t) for t in tasks])]
    forbid = peaked_weights["model.forbid_rep_mask"]
    fe = torch.from_numpy(g["frame_embs"]).cuda()
    lens = torch.from_numpy(g["audio_shape"][:, 1].astype(np.int32))
    out = eng.decode(fe, lens, bos, forbid, beam, min_pred, max_pred, want_trace=True)
    torch.cuda.synchronize()
    sel = out["trace_sel"].cpu().numpy()
    calls = _calls(g, beam, max_pred)
    tol = TOL[prec]
    diverged, n_checked, n_same = set(), 0, 0
    for step, clip, par, tok, sums, eff in calls:
        if clip in diverged:
            continue
        k = len(par)
        same = sel[step, clip, :k, 0].tolist() == par and sel[step, clip, :k, 1].tolist() == tok
        n_same += same
        if eff <= tol:          # a near-tie at this precision: either outcome is acceptable, later calls of the clip are not comparable
            if not same:
                diverged.add(clip)
            continue
        assert same, (name, prec, step, clip, sel[step, clip, :k].tolist(), par, tok, eff)
        n_checked += 1
    print(f"peaked {name}/{prec}: {n_checked} of {len(calls)} calls above {tol} verified, {n_same} identical, diverged clips {sorted(diverged)}")
    # (this sequential comparison stops at a clip's first near-tie that falls the other way: its count is no coverage measure --
    # test_topk_at_reference_states_peaked checks every call independently)
    if prec == "exact":
        assert n_same == len(calls) and not diverged
    keep = [b for b in range(len(g["lengths"])) if b not in diverged]
    bm = int(out["sizes"][1].item())
    got = out["best_preds"][:, :bm].cpu().numpy()
    ref = g["preds"]
    w = min(got.shape[1], ref.shape[1])
    assert np.array_equal(got[keep][:, :w], ref[keep][:, :w]), (name, prec)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16", "f16", "exact"])
@pytest.mark.parametrize("name", PK)
def test_topk_at_reference_states_peaked(name, prec, peaked_engines, peaked_weights, synth_cfg):
    """Every top-k call of the reference's search checked independently at the reference's own prefixes (golden_util.
    topk_at_reference_states): all calls above the precision's margin pick the reference's (parents, tokens), and that is at least
    the fixture's floor share of ALL calls -- 80 % of the greedy decisions in bf16, the benchmarked dtype."""
    g = _load(name)
    kw = json.loads(str(g["kw"]))
    beam = kw.get("beam_size", synth_cfg["beam_size"])
    calls = _calls(g, beam, kw.get("max_pred_size", synth_cfg["max_pred_size"]))
    tol = TOL[prec]
    n_checked, n_same, n_eligible = G.topk_at_reference_states(
        g, calls, peaked_engines[prec], peaked_weights, synth_cfg, peaked_weights["model.forbid_rep_mask"], tol,
        sum_atol=(lambda step: 2e-4 * (step + 1)) if prec == "exact" else (lambda step: 0.2 * (1.0 if prec == "bf16" else 0.125)), tag=(name, prec))
    print(f"peaked top-k at reference states {name}/{prec}: {n_checked} of {len(calls)} calls above {tol} verified, {n_same} identical")
    if prec == "exact":
        assert n_same == len(calls)
    else:
        floor = ABOVE_FLOOR[name] if prec == "bf16" else min(0.95, ABOVE_FLOOR[name] + 0.15)   # (f16's margin is an eighth of bf16's)
        assert n_checked >= floor * len(calls), (n_checked, len(calls))


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("name", [n for n in PK if "beam1" in n])
def test_greedy_captions_from_waveform_on_peaked_checkpoint(name, prec, peaked_weights, synth_cfg):
    """The whole path in the benchmarked dtype -- waveform -> HIP encoder -> HIP greedy search -- against the reference's captions:
    every clip whose reference decisions all have a margin above the precision's tolerance must return the reference's token ids."""
    from conette_amd import CoNeTTEConfig, synth
    from conette_amd.model import CoNeTTEModel
    g = _load(name)
    x, kw = G.inputs(g)
    sd = dict(peaked_weights)
    sd["_extra_state_"] = torch.from_numpy(synth.extra_state_tensor())
    model = CoNeTTEModel(CoNeTTEConfig(**synth_cfg), device="cuda:0", state_dict=sd, precision=prec,
                         audioset_idx_to_name={i: f"tag{i}" for i in range(527)})
    out = model(x, sr=32000, **kw)
    got, ref = out["preds"].cpu().numpy(), g["preds"]
    calls = _calls(g, 1, kw.get("max_pred_size", synth_cfg["max_pred_size"]))
    # (the encoder's 16-bit rounding moves the logits a little more than the decoder alone: twice the decode-only margin)
    safe = [b for b in range(len(g["lengths"])) if all(c[5] > 2 * TOL[prec] for c in calls if c[1] == b)]
    w = min(got.shape[1], ref.shape[1])
    same = [b for b in range(len(g["lengths"])) if np.array_equal(got[b, :w], ref[b, :w])]
    print(f"peaked {name}/{prec}: {len(same)} of {len(ref)} captions identical to the reference's; {len(safe)} clips have every margin above {2 * TOL[prec]}")
    assert set(safe) <= set(same), (name, prec, safe, same)
    assert len(safe) >= 1 and len(same) >= len(ref) // 2

"""CPU: the oracle (oracle/cpu_ref.py) reproduces the fixtures generated from the imported
reference (oracle/gen_golden.py).  This is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from tests import golden_util as G


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_oracle_matches_reference_fixture(name, synth_weights, synth_cfg):
    g = G.load(name)
    x, kw = G.inputs(g)
    taps, trace = {}, []
    torch.manual_seed(0)
    with torch.no_grad():
        out = O.model_forward(synth_weights, synth_cfg, x, sr=32000, taps=taps, trace=trace, **kw)
    # intermediates: the oracle uses the same torch primitives as the reference -> tight
    for k in G.TAPS:
        np.testing.assert_allclose(G.sub(taps[k]), g["sub_" + k], rtol=1e-4, atol=1e-4, err_msg=k)
        assert abs(float(taps[k].double().sum()) - float(g["sum_" + k])) <= 1e-5 * float(g["abs_" + k]) + 1e-3
    np.testing.assert_allclose(out["tags_probs"].numpy(), g["tags_probs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(taps["memory"].numpy(), g["memory"], rtol=1e-4, atol=1e-4)
    # search: ids bit-exact, scores to 1e-5, every per-step (parent, token) identical
    assert out["preds"].tolist() == g["preds"].tolist()
    assert out["mult_preds"].tolist() == g["mult_preds"].tolist()
    np.testing.assert_allclose(out["lprobs"].numpy(), g["lprobs"], atol=1e-5)
    np.testing.assert_allclose(out["mult_lprobs"].numpy(), g["mult_lprobs"], atol=1e-5)
    par, tok, sums, _ = G.trace_of(g)
    flat = [c for st in trace for c in st]
    assert [c["parent"] for c in flat] == par
    assert [c["token"] for c in flat] == tok
    assert out["cands"] == json.loads(str(g["cands"]))
    assert out["mult_cands"] == json.loads(str(g["mult_cands"]))
    assert out["tasks"] == json.loads(str(g["tasks"]))
    assert out["tags"] == json.loads(str(g["tags"]))


def test_oracle_api_cases(synth_weights, synth_cfg):
    from conette_amd import synth
    with open(G.GOLDEN + "/api_cases.json") as f:
        api = json.load(f)
    wav = torch.from_numpy(synth.synth_waveforms(2, 64000, 7000))
    f = lambda *a, **k: O.model_forward(synth_weights, synth_cfg, *a, **k)
    with torch.no_grad():
        assert f(wav[0], sr=32000)["preds"].tolist() == api["rank1_preds"]
        assert f(wav[0][None], sr=32000)["preds"].tolist() == api["rank2_preds"]
        assert f(wav[0:1, None], sr=32000)["preds"].tolist() == api["rank3_preds"]
        o4 = f(torch.stack([wav[0], wav[1]], 0), sr=32000)
        assert o4["preds"].tolist() == api["stereo_preds"]
        o5 = f([wav[0][None], wav[1][None]], sr=[32000, 32000], task=["clotho", "audiocaps"])
        assert o5["preds"].tolist() == api["list_preds"] and o5["cands"] == api["list_cands"]
        pre = O.preprocessor_forward(synth_weights, wav[:, None], 32000)
        o6 = f(pre["audio"], x_shapes=pre["audio_shape"], preprocess=False, task="clotho")
        assert o6["preds"].tolist() == api["nopre_preds"] and sorted(o6.keys()) == api["nopre_keys"]
        o7 = f(wav[:, None], sr=32000, task="clotho")
        assert sorted(o7.keys()) == api["full_keys"]
        assert f(wav[0][::2].contiguous(), sr=16000)["preds"].tolist() == api["sr16k_preds"]
    with pytest.raises(ValueError) as e:
        f(wav[0], sr=32000, task="not_a_task")
    assert str(e.value) == api["bad_task_error"]
    with pytest.raises(ValueError) as e:
        f(wav[:, None], sr=32000, task=["clotho"])
    assert str(e.value) == api["bad_ntasks_error"]
    with pytest.raises(ValueError):
        f(wav[0], sr=16000, x_shapes=torch.tensor([[64000]]))


def test_oracle_teacher_forcing_matches_reference_fixture(synth_weights):
    """the oracle's restatement of forcing.py against logits produced by the reference itself
    (oracle/gen_golden_forcing.py)."""
    import os
    g = np.load(os.path.join(G.GOLDEN, "forcing", "forcing_ragged.npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    logits = O.teacher_forcing(synth_weights, torch.from_numpy(g["frame_embs"]), torch.from_numpy(g["audio_shape"]),
                               torch.from_numpy(g["caps_in"]))
    assert tuple(logits.shape) == tuple(g["logits"].shape)  # (B, V, cap_len)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=2e-4)
    bad = torch.from_numpy(g["caps_in"]).clone()
    bad[0, 0] = 1  # <bos> left in place of the task token (conette.py:399-404)
    with pytest.raises(ValueError, match="BOS was not replaced"):
        O.teacher_forcing(synth_weights, torch.from_numpy(g["frame_embs"]), torch.from_numpy(g["audio_shape"]), bad)


@pytest.mark.parametrize("name", ["greedy_bos", "greedy_task"])
def test_oracle_greedy_search_matches_reference_fixture(name, synth_weights):
    """the oracle's restatement of greedy.py against masked logits produced by the reference."""
    import os
    g = np.load(os.path.join(G.GOLDEN, "forcing", name + ".npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    mem, mask = O.encode_audio(synth_weights, torch.from_numpy(g["frame_embs"]), torch.from_numpy(g["audio_shape"]))
    fm = synth_weights["model.forbid_rep_mask"] if int(g["use_forbid"]) else None
    lg = O.greedy_search(synth_weights, mem, mask, int(g["bos_id"]), vocab_size=synth_weights["model.decoder.classifier.weight"].shape[0],
                         min_pred_size=int(g["min_pred"]), max_pred_size=int(g["max_pred"]), forbid_rep_mask=fm)
    ref = torch.from_numpy(g["logits"])
    assert tuple(lg.shape) == tuple(ref.shape)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(lg), fin)                    # EOS floor, forbid-repeat, finished-clip fill
    np.testing.assert_allclose(lg[fin].numpy(), ref[fin].numpy(), rtol=1e-4, atol=2e-4)
    assert torch.equal(lg.argmax(dim=1), ref.argmax(dim=1))

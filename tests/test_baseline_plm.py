"""BaselinePLM checkpoints (reference pl_modules/baseline.py:84-140,309-410): a BaselinePLM-layout state
dict -- tokenizer state, projection, decoder, forbid mask; NO audio encoder, NO task tokens -- loaded by conette_amd.baseline.BaselinePLM
into a decoder-only HIP context.  The fixture (tests/golden/baseline/baseline_b4.npz) holds what the REFERENCE's own BaselinePLM
returns for the same synthetic checkpoint on the frame embeddings of a committed CoNeTTE fixture (oracle/gen_golden_baseline.py).

CPU: the oracle's decode functions, prompted with the plain <bos>, reproduce the fixture.  GPU: the three decode methods of the
HIP path against it (ids bit-exact in the exact / fp32 precisions)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from tests import golden_util as G

FIX = os.path.join(G.GOLDEN, "baseline", "baseline_b4.npz")


@pytest.fixture(scope="module")
def baseline_sd():
    from conette_amd import synth
    sd = synth.synth_baseline_state_dict()
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}


@pytest.fixture(scope="module")
def fixture():
    g = np.load(FIX)
    src = G.load(str(g["src"]))
    return g, torch.from_numpy(src["frame_embs"]), torch.from_numpy(g["audio_shape"])


def _oracle_weights(sd):
    w = {"model." + k: v.float() if v.is_floating_point() else v for k, v in sd.items() if isinstance(v, torch.Tensor) and k != "forbid_rep_mask"}
    return w, sd["forbid_rep_mask"].bool()


def test_oracle_reproduces_reference_baseline_plm(baseline_sd, fixture):
    g, fe, shape = fixture
    w, forbid = _oracle_weights(baseline_sd)
    v = w["model.decoder.classifier.weight"].shape[0]
    assert v == 5624                                    # no task tokens (CoNeTTE: 5631)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        mem, mask = O.encode_audio(w, fe, shape)
        bos = torch.ones(fe.shape[0], dtype=torch.long)     # the plain <bos> (baseline.py:380-391)
        trace = []
        preds, lprobs, mult_preds, mult_lprobs = O.generate(w, mem, mask, bos, vocab_size=v, beam_size=3, min_pred_size=3, max_pred_size=20,
                                                            forbid_rep_mask=forbid, trace=trace)
        greedy = O.greedy_search(w, mem, mask, 1, vocab_size=v, min_pred_size=3, max_pred_size=20, forbid_rep_mask=forbid)
        forcing = O.teacher_forcing(w, fe, shape, torch.from_numpy(g["caps_in"]), require_task_token=False)
    assert preds.tolist() == g["preds"].tolist() and mult_preds.tolist() == g["mult_preds"].tolist()
    np.testing.assert_allclose(lprobs.numpy(), g["lprobs"], atol=1e-5)
    np.testing.assert_allclose(mult_lprobs.numpy(), g["mult_lprobs"], atol=1e-5)
    flat = [c for st in trace for c in st]
    assert [c["parent"] for c in flat] == json.loads(str(g["trace_parent"])) and [c["token"] for c in flat] == json.loads(str(g["trace_token"]))
    assert greedy.argmax(dim=1).tolist() == g["greedy_ids"].tolist()
    ref_g, got_g = torch.from_numpy(g["greedy_logits_sub"]), greedy[:, ::37, :]
    fin = torch.isfinite(ref_g)
    assert torch.equal(fin, torch.isfinite(got_g))
    np.testing.assert_allclose(got_g[fin].numpy(), ref_g[fin].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(forcing[:, ::37, :].numpy(), g["forcing_logits_sub"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["exact", "fp32", "bf16", "f16"])
def test_baseline_plm_on_gpu_matches_reference(prec, baseline_sd, fixture):
    from conette_amd.baseline import BaselinePLM
    g, fe, shape = fixture
    plm = BaselinePLM(baseline_sd, beam_size=3, max_pred_size=20, precision=prec)
    batch = {"audio": fe[:, None], "audio_shape": shape}     # (B, 1, T, 768): FrameIdentEncoder squeezes the channel (ident.py:19-21)
    out = plm(batch, "generate")
    greedy = plm(batch, "greedy").cpu()
    forcing = plm.decode_audio(plm.encode_audio(batch["audio"], batch["audio_shape"]), "forcing", caps_in=torch.from_numpy(g["caps_in"])).cpu()
    exact = prec in ("exact", "fp32")
    r16 = {"bf16": 1.0, "f16": 0.125}.get(prec, 0.0)
    # teacher forcing and the first greedy step do not depend on earlier decisions: comparable in every precision
    np.testing.assert_allclose(forcing[:, ::37, :].numpy(), g["forcing_logits_sub"], rtol=1e-4 if exact else 0, atol=2e-4 if exact else 0.35 * r16)
    ref_g = torch.from_numpy(g["greedy_logits_sub"])
    got_g = greedy[:, ::37, : ref_g.shape[2]]
    fin0 = torch.isfinite(ref_g[:, :, 0])
    np.testing.assert_allclose(got_g[:, :, 0][fin0].numpy(), ref_g[:, :, 0][fin0].numpy(), rtol=1e-4 if exact else 0, atol=2e-4 if exact else 0.35 * r16)
    if exact:
        assert out["preds"].cpu().tolist() == g["preds"].tolist()
        assert out["mult_preds"].cpu().tolist() == g["mult_preds"].tolist()
        np.testing.assert_allclose(out["lprobs"].cpu().numpy(), g["lprobs"], atol=1e-4)
        assert out["cands"] == json.loads(str(g["cands"]))
        assert greedy.argmax(dim=1)[:, : g["greedy_ids"].shape[1]].tolist() == g["greedy_ids"].tolist()
        fin = torch.isfinite(ref_g)
        assert torch.equal(fin, torch.isfinite(got_g))
        np.testing.assert_allclose(got_g[fin].numpy(), ref_g[fin].numpy(), rtol=1e-4, atol=2e-4)
    else:
        same = sum(int(a == b) for a, b in zip(out["preds"].cpu().tolist(), [r[: out["preds"].shape[1]] for r in g["preds"].tolist()]))
        print(f"baseline {prec}: {same} of {len(g['preds'])} beam-3 captions identical to the reference's")
        assert np.all(np.abs(out["lprobs"].cpu().numpy() - g["lprobs"]) < 0.5)


@pytest.mark.gpu
def test_decoder_only_context_refuses_encode(baseline_sd):
    from conette_amd.baseline import BaselinePLM
    plm = BaselinePLM(baseline_sd, precision="bf16")
    with pytest.raises(RuntimeError, match="decoder-only"):
        plm.engine.encode(torch.zeros((1, 32000), device="cuda"))
    with pytest.raises(ValueError, match="caps_in"):
        plm.decode_audio({"frame_embs": torch.zeros((1, 4, 768)), "frame_embs_lens": torch.tensor([4])}, "forcing")
    with pytest.raises(ValueError, match="Unknown argument"):
        plm.decode_audio({"frame_embs": torch.zeros((1, 4, 768)), "frame_embs_lens": torch.tensor([4])}, "sample")


# decode hyper-parameters of the reference's BaselinePLM.__init__ (pl_modules/baseline.py:36-52)
REFERENCE_DEFAULTS = {"proj_name": "lin768", "min_pred_size": 3, "max_pred_size": None, "beam_size": 10, "nhead": 8, "d_model": 256,
                      "num_decoder_layers": 6, "dim_feedforward": 2048, "acti_name": "gelu"}


def test_constructor_defaults_are_the_references():
    """a BaselinePLM built without decode hyper-parameters must search like the reference's (beam_size 10, not 2)."""
    import ast
    import inspect
    from conette_amd.baseline import BaselinePLM
    sig = inspect.signature(BaselinePLM.__init__)
    ours = {k: sig.parameters[k].default for k in REFERENCE_DEFAULTS}
    assert ours == REFERENCE_DEFAULTS
    ref_file = "/root/reference/src/conette/pl_modules/baseline.py"
    if os.path.isfile(ref_file):      # (this container only: the golden list above against the reference's own source)
        tree = ast.parse(open(ref_file).read())
        init = next(f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "BaselinePLM"
                    for f in c.body if isinstance(f, ast.FunctionDef) and f.name == "__init__")
        names = [a.arg for a in init.args.args][1:]
        vals = [ast.literal_eval(d) for d in init.args.defaults]
        ref = dict(zip(names[len(names) - len(vals):], vals))
        assert {k: ref[k] for k in REFERENCE_DEFAULTS} == REFERENCE_DEFAULTS


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["exact", "certified:f16"])
def test_default_constructed_baseline_plm_searches_with_beam_10(prec, baseline_sd, fixture):
    """No decode hyper-parameters given: beam 10 (the generic search-step kernel: the register-resident one holds 8 rows),
    max_pred_size from the tokenizer -- ids equal the oracle's run with the reference's defaults."""
    from conette_amd.baseline import BaselinePLM
    g, fe, shape = fixture
    plm = BaselinePLM(baseline_sd, precision=prec)
    assert plm.hp["beam_size"] == 10 and plm.hp["min_pred_size"] == 3
    out = plm({"audio": fe[:, None], "audio_shape": shape}, "generate")
    w, forbid = _oracle_weights(baseline_sd)
    v = w["model.decoder.classifier.weight"].shape[0]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        mem, mask = O.encode_audio(w, fe, shape)
        preds, lprobs, mult_preds, mult_lprobs = O.generate(w, mem, mask, torch.ones(fe.shape[0], dtype=torch.long), vocab_size=v, beam_size=10,
                                                            min_pred_size=3, max_pred_size=plm.hp["max_pred_size"], forbid_rep_mask=forbid)
    assert out["preds"].cpu().tolist() == preds.tolist()
    assert out["mult_preds"].cpu().tolist() == mult_preds.tolist()
    np.testing.assert_allclose(out["lprobs"].cpu().numpy(), lprobs.numpy(), atol=2e-4 if prec == "exact" else 0.05)

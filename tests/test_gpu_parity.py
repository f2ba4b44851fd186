"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(conette_amd.engine -> libconette_hip.so); the checker is the committed golden fixtures
(outputs of the imported reference) plus the CPU oracle on small cases.

Tolerances:
  fp32 mode : intermediates rtol 1e-3 / atol 2e-4 of an O(1) activation; token ids bit-exact
              (every golden candidate gap is > 6e-4); averaged log-probs abs 1e-4.
  exact mode: the fp32 tolerances (operands are fp16 hi + lo pairs = 22 bits, products of three fp16 MFMAs).
  bf16 mode : intermediates abs 0.1 (observed max 0.035 at rms ~1.1); every beam decision whose
              reference top-(k+1) candidates are separated by > 0.25 must match for as long as the
              search state is still the reference's; best averaged log-prob abs 0.15.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu

# 16-bit operand precisions: every tolerance made of operand rounding scales with it (fp16: 11 significant bits against 8)
R16 = {"bf16": 1.0, "f16": 0.125}
# running log-prob sums of the search against the fp32 reference, per step: the logits of the 16-bit-operand ORACLE sit 0.21 (bf16) /
# 0.024 (fp16) from the reference's at the worst element of the forcing fixture
SUM_TOL = {"bf16": 0.12, "f16": 0.03}
# precisions held to the fp32 tolerances and to bit-exact ids: the fp32-MFMA mode and the "exact" mode (fp16 hi/lo operand
# pairs, three MFMAs per product: include/conette_hip.h CONETTE_PREC_F16X2)
EXACT = ("fp32", "exact")
# measured on MI355X (round 4, profiles/r04_topk_rates.txt): per scenario and 16-bit precision [calls above the margin verified,
# calls identical to the reference's (parents, tokens, order), calls].  The test holds a build to these minus a margin of
# max(2, 8 % of the calls): last-bit changes of a kernel move a few near-ties, a regression that halves the agreement fails.
TOPK_MEASURED = {
    "b1_30s_beam3": {"bf16": [2, 18, 20], "f16": [16, 20, 20]},
    "b2_1s_beam3": {"bf16": [7, 20, 24], "f16": [21, 23, 24]},
    "b2_4s_beam8_content_words": {"bf16": [0, 11, 21], "f16": [7, 17, 21]},
    "b3_mixed_beam2_tasks": {"bf16": [25, 56, 60], "f16": [53, 60, 60]},
    "b3_mixed_beam3_none": {"bf16": [8, 52, 60], "f16": [50, 59, 60]},
    "b4_10s_beam1_audiocaps": {"bf16": [59, 71, 71], "f16": [70, 71, 71]},
    "b4_10s_beam3_clotho": {"bf16": [17, 72, 80], "f16": [68, 78, 80]},
    "b4_odd_beam5_minpred0": {"bf16": [22, 90, 111], "f16": [79, 109, 111]},
    "b8_10s_beam3_all": {"bf16": [25, 122, 146], "f16": [111, 144, 146]},
}
# share of (clip, step) pairs of the greedy fixtures reached with the reference's arg-max chain intact: measured 26 / 36 and 31 / 36
# (bf16), 36 / 36 and 36 / 36 (f16) -- held to those rates minus a margin
GREEDY_FLOOR = {"bf16": 0.6, "f16": 0.9}
NCHW_TAPS = ["stem", "stage0_block0", "stage0", "down1", "stage1", "down2", "stage2", "down3", "stage3"]


@pytest.fixture(scope="module")
def engines(synth_weights):
    from conette_amd.engine import Engine
    return {"fp32": Engine(synth_weights, precision="fp32"), "bf16": Engine(synth_weights, precision="bf16"),
            "exact": Engine(synth_weights, precision="exact"), "f16": Engine(synth_weights, precision="f16")}


def _padded_wave(g):
    from conette_amd import synth
    n = [int(v) for v in g["lengths"]]
    return torch.from_numpy(synth.synth_waveforms(len(n), max(n), int(g["seed0"]), lengths=n)), n


@pytest.mark.parametrize("prec", ["fp32", "exact", "bf16", "f16"])
@pytest.mark.parametrize("name", G.SCENARIOS)
def test_encode_matches_reference_fixture(name, prec, engines):
    g = G.load(name)
    wave, n = _padded_wave(g)
    fe, clip, taps = engines[prec].encode(wave.cuda(), taps=True)
    torch.cuda.synchronize()
    rtol, atol = (1e-3, 2e-4) if prec in EXACT else (0.0, 0.1 * R16[prec])
    np.testing.assert_allclose(G.sub(taps["logmel"]), g["sub_logmel"], rtol=1e-3, atol=2e-3, err_msg="logmel")
    for k in NCHW_TAPS:
        got = G.sub(taps[k].permute(0, 3, 1, 2).contiguous())
        np.testing.assert_allclose(got, g["sub_" + k], rtol=rtol, atol=atol, err_msg=f"{name}/{prec}/{k}")
        if prec in R16:  # the bulk must be much closer than the worst element
            assert float(np.abs(got - g["sub_" + k]).mean()) < 0.02 * R16[prec], k
    np.testing.assert_allclose(fe.cpu().numpy(), g["frame_embs"], rtol=rtol, atol=atol if prec in EXACT else 0.06 * R16[prec])
    # accumulated drift of the whole encoder (18 blocks + 3 downsample layers) in relative-rms terms: the element-wise bounds
    # above are worst cases; this is the number the id agreement depends on (bench.py reports it next to the agreement:
    # 4.4e-3 in bf16 with 72 % of the beam-3 captions identical, 1e-6 in the exact modes with 100 %)
    ref_fe = torch.from_numpy(g["frame_embs"])
    rel = float((fe.cpu() - ref_fe).pow(2).mean().sqrt() / ref_fe.pow(2).mean().sqrt())
    assert rel < (1e-5 if prec in EXACT else 8e-3 * R16[prec]), (name, prec, rel)
    np.testing.assert_allclose(clip.cpu().numpy(), g["tags_probs"], rtol=rtol, atol=1e-4 if prec in EXACT else 0.03 * R16[prec])


def _forbid_mask(synth_weights, synth_cfg, mode):
    """the mask of a forbid_rep_mode as the reference builds it (pl_modules/common.py:222-299, restated in the oracle)"""
    from oracle import cpu_ref as O
    itos = {int(k): v for k, v in synth_cfg["tokenizer_state"]["tokenizer"]["itos"].items()}
    return O.forbid_mask_for_mode(synth_weights, mode, itos)


def _ref_calls(g, beam):
    """Golden per-call trace -> list of (step, clip, parents, tokens, sums, margin)."""
    par, tok, sums, margin = G.trace_of(g)
    kw = json.loads(str(g["kw"]))
    bsz = len(g["lengths"])
    max_pred = kw.get("max_pred_size", 20)
    k = [beam] * bsz
    calls, ci = [], 0
    for step in range(max_pred):
        for clip in range(bsz):
            if k[clip] == 0:
                continue
            assert len(par[ci]) == k[clip]
            calls.append((step, clip, par[ci], tok[ci], sums[ci], float(margin[ci])))
            n_fin = k[clip] if step == max_pred - 1 else sum(1 for t in tok[ci] if t == 2)
            k[clip] -= n_fin
            ci += 1
        if ci == len(par):
            break
    assert ci == len(par)
    return calls


@pytest.mark.parametrize("prec", ["fp32", "exact", "bf16", "f16"])
@pytest.mark.parametrize("name", G.SCENARIOS)
def test_decode_matches_reference_fixture(name, prec, engines, synth_weights, synth_cfg):
    """Beam search from the reference's own frame embeddings: per-step decisions and outputs."""
    g = G.load(name)
    kw = json.loads(str(g["kw"]))
    eng = engines[prec]
    bsz = len(g["lengths"])
    beam = kw.get("beam_size", synth_cfg["beam_size"])
    min_pred = kw.get("min_pred_size", synth_cfg["min_pred_size"])
    max_pred = kw.get("max_pred_size", synth_cfg["max_pred_size"])
    tasks = json.loads(str(g["tasks"]))
    task_names = list(synth_cfg["task_names"])
    bos = synth_weights["model.task_id_to_token_id"][torch.as_tensor([task_names.index(t) for t in tasks])]
    mode = kw.get("forbid_rep_mode")
    v = eng.vocab_size
    forbid = _forbid_mask(synth_weights, synth_cfg, mode)
    fe = torch.from_numpy(g["frame_embs"]).cuda()
    lens = torch.from_numpy(g["audio_shape"][:, 1].astype(np.int32))
    out = eng.decode(fe, lens, bos, forbid, beam, min_pred, max_pred, want_trace=True)
    torch.cuda.synchronize()
    sel = out["trace_sel"].cpu().numpy()
    val = out["trace_val"].cpu().numpy()
    # Effective margin of a call = smallest gap among its top-(k+1) candidates: the gap to the first
    # rejected candidate (recorded) and the gaps between consecutive picks (order decides row slots).
    tol = 5e-4 if prec in EXACT else 0.25 * R16[prec]
    diverged = set()
    n_checked = n_tie = 0
    for step, clip, par, tok, sums, margin in _ref_calls(g, beam):
        if clip in diverged:
            continue
        k = len(par)
        eff = min([margin] + [sums[i] - sums[i + 1] for i in range(k - 1)])
        same = sel[step, clip, :k, 0].tolist() == par and sel[step, clip, :k, 1].tolist() == tok
        if eff <= tol:
            # a near-tie at this precision: either outcome is acceptable, but once the GPU path takes
            # the other branch its later decisions are no longer comparable for this clip
            n_tie += 1
            if not same:
                diverged.add(clip)
            continue
        assert same, (name, prec, step, clip, sel[step, clip, :k].tolist(), par, tok)
        np.testing.assert_allclose(val[step, clip, :k], sums, atol=2e-4 * (step + 1) if prec in EXACT else SUM_TOL[prec] * (step + 1))
        n_checked += 1
    n_calls = len(_ref_calls(g, beam))
    print(f"decode {name}/{prec}: {n_checked} of {n_calls} top-k calls checked, diverged clips {sorted(diverged)}")
    # fp32: every call.  bf16: this sequential comparison stops at a clip's first near-tie that falls the other way, so
    # its count is not a coverage measure -- test_topk_decisions_at_reference_states checks every call independently.
    if prec in EXACT:  # every call is either checked or a tie within 5e-4 (the beam-8 fixture has two) that fell the reference's way
        assert n_checked + n_tie == n_calls and n_tie <= 2 and not diverged
    ps, bm = (int(x) for x in out["sizes"].tolist())
    if not diverged:
        assert out["mult_preds"][:, :, :ps].cpu().tolist() == g["mult_preds"].tolist()
        assert out["best_preds"][:, :bm].cpu().tolist() == g["preds"].tolist()
        np.testing.assert_allclose(out["mult_lprobs"].cpu().numpy(), g["mult_lprobs"], atol=1e-4 if prec in EXACT else 0.05 * R16[prec])
    # beam log-prob tolerance (north_star): clips whose search never left the reference's state must
    # agree to 1e-4 (fp32) / 0.05 (bf16); a clip that took the other side of a near-tie decodes a
    # different caption, whose length-normalised score is only sanity-bounded
    keep = [b for b in range(bsz) if b not in diverged]
    got_lp = out["best_lprobs"].cpu().numpy()
    np.testing.assert_allclose(got_lp[keep], g["lprobs"][keep], atol=1e-4 if prec in EXACT else 0.05 * R16[prec])
    assert np.all(np.abs(got_lp - g["lprobs"]) < 0.5) and np.all(got_lp < 0)
    if prec in EXACT:
        assert not diverged  # every golden candidate gap exceeds the fp32 tolerance


@pytest.mark.parametrize("prec", ["fp32", "exact", "bf16", "f16"])
@pytest.mark.parametrize("name", G.SCENARIOS)
def test_topk_decisions_at_reference_states(name, prec, engines, synth_weights, synth_cfg):
    """Every _select_k_next_toks call of the reference's beam search (beam.py:230-269), checked INDEPENDENTLY: the
    decoder kernels are fed the reference's own prefixes (teacher forcing, KV-cached step kernels), the step's masking /
    log-softmax / running sums / flat top-k are restated here, and the picks must equal the reference's wherever its
    top-(k+1) margin exceeds the precision's tolerance.  Unlike the sequential trace comparison above, a near-tie that
    falls the other way in bf16 does not hide the later calls of that clip."""
    g = G.load(name)
    kw = json.loads(str(g["kw"]))
    beam = kw.get("beam_size", synth_cfg["beam_size"])
    calls = _ref_calls(g, beam)
    tol = 5e-4 if prec in EXACT else 0.25 * R16[prec]
    forbid = _forbid_mask(synth_weights, synth_cfg, kw.get("forbid_rep_mode"))
    n_checked, n_same, n_eligible = G.topk_at_reference_states(
        g, calls, engines[prec], synth_weights, synth_cfg, forbid, tol,
        sum_atol=(lambda step: 2e-4 * (step + 1)) if prec in EXACT else (lambda step: 0.2 * R16[prec]), tag=(name, prec))
    print(f"top-k at reference states {name}/{prec}: {n_checked} of {len(calls)} calls above the margin verified, "
          f"{n_same} of {len(calls)} calls identical (parents, tokens and their order)")
    # exact precisions: every call above the 5e-4 margin (all but the two near-ties of the beam-8 fixture) verified, and ALL calls,
    # ties included, identical to the reference's
    if prec in EXACT:
        assert n_checked >= len(calls) - 2 and n_same == len(calls), (n_checked, n_same, len(calls))
    else:
        # 16-bit operands: pinned to what this scenario measured (TOPK_MEASURED), not to a global floor ("floors, not
        # pins": 0.75 x calls and `>= min(1, eligible)` let a build that halves the agreement pass)
        m_checked, m_same, m_calls = TOPK_MEASURED[name][prec]
        assert m_calls == len(calls)
        slack = max(2, (8 * len(calls) + 99) // 100)
        assert n_same >= m_same - slack, (n_same, m_same, len(calls))
        assert n_checked >= m_checked - slack, (n_checked, m_checked, n_eligible)
        assert n_checked >= 1 or m_checked <= slack, (n_checked, n_eligible)


def test_one_pass_forcing_is_faster_than_stepwise(engines):
    """the one-pass teacher forcing must beat the cap_len dependent steps by a wide margin at cap_len 20."""
    eng = engines["bf16"]
    B, T, cap = 64, 31, 20
    g = torch.Generator().manual_seed(3)
    fe = torch.randn((B, T, 768), generator=g).cuda()
    lens = torch.full((B,), T, dtype=torch.int32)
    caps = torch.randint(4, 5000, (B, cap), generator=g)
    times = {}
    for mode in ("onepass", "stepwise"):
        eng.set_forcing_stepwise(mode == "stepwise")
        try:
            for _ in range(2):
                out = eng.forcing(fe, lens, caps)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = eng.forcing(fe, lens, caps)
            e1.record()
            torch.cuda.synchronize()
            times[mode] = e0.elapsed_time(e1) / 5
        finally:
            eng.set_forcing_stepwise(False)
    print(f"teacher forcing B=64 cap_len=20: one pass {times['onepass']:.3f} ms, stepwise {times['stepwise']:.3f} ms")
    assert times["onepass"] * 3 < times["stepwise"], times


def test_frontend_against_oracle_small(engines, synth_weights):
    """Direct oracle check (DFT-as-conv restatement) on odd lengths incl. the reflect-padded edges."""
    from conette_amd import synth
    from oracle import cpu_ref as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for L in (32000, 48001, 16383):
        wave = torch.from_numpy(synth.synth_waveforms(2, L, 99, lengths=[L, L // 3]))
        got = engines["fp32"].frontend_logmel(wave.cuda()).cpu()
        with torch.no_grad():
            ref = O.logmel_bn0(synth_weights, wave)[:, 0]
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=2e-3)


def test_resample_against_oracle(engines):
    from oracle import thirdparty as tp
    x = torch.randn(3, 44100, generator=torch.Generator().manual_seed(0))
    for o, n in ((44100, 32000), (16000, 32000), (48000, 32000)):
        got = engines["fp32"].resample(x.cuda(), o, n).cpu()
        ref = tp.resample(x, o, n)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.fixture(params=["onepass", "stepwise"])
def forcing_mode(request, engines):
    """conette_forcing as one causal pass (default) or through the KV-cached step kernels (CONETTE_OPT_FORCING_STEPWISE)."""
    for e in engines.values():
        e.set_forcing_stepwise(request.param == "stepwise")
    yield request.param
    for e in engines.values():
        e.set_forcing_stepwise(False)


@pytest.mark.parametrize("prec", ["fp32", "exact", "bf16", "f16"])
def test_teacher_forcing_matches_reference_fixture(prec, engines, forcing_mode):
    """conette_forcing against logits produced by the reference itself (all caption positions,
    padded ones included: padded positions are masked as keys exactly like tensor_to_pad_mask does)."""
    g = np.load(os.path.join(G.GOLDEN, "forcing", "forcing_ragged.npz"))
    eng = engines[prec]
    fe = torch.from_numpy(g["frame_embs"]).cuda()
    lens = torch.from_numpy(g["audio_shape"][:, 1].astype(np.int32))
    caps = torch.from_numpy(g["caps_in"])
    got = eng.forcing(fe, lens, caps).permute(0, 2, 1).cpu().numpy()  # (B, V, cap_len) like the reference
    ref = g["logits"]
    assert got.shape == ref.shape
    if prec in EXACT:
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-3)
    else:
        # bf16 operands: logits span +-40; compare log-probabilities of the reference's top candidates
        # (f16: every bound made of operand rounding is an eighth of the bf16 one, R16)
        r16 = R16[prec]
        err = np.abs(got - ref)
        assert err.max() < 0.6 * r16 and err.mean() < 0.06 * r16, (err.max(), err.mean())
        lp_got = torch.log_softmax(torch.from_numpy(got), dim=1)
        lp_ref = torch.log_softmax(torch.from_numpy(ref), dim=1)
        top = lp_ref.argmax(dim=1, keepdim=True)
        d = (lp_got.gather(1, top) - lp_ref.gather(1, top))[:, 0].numpy()      # (B, cap_len)
        assert np.abs(d).max() < 0.12 * r16, np.abs(d).max()                    # per token
        valid = g["caps_in"] != 0
        per_cap = np.abs((d * valid).sum(axis=1) / valid.sum(axis=1))           # length-averaged, like the beam score
        assert per_cap.max() < 0.05 * r16, per_cap                              # north_star bf16 tolerance
    # twice the same call: deterministic; a batch of one: row independent
    again = eng.forcing(fe, lens, caps).permute(0, 2, 1).cpu().numpy()
    assert np.array_equal(got, again)
    one = eng.forcing(fe[1:2].contiguous(), lens[1:2], caps[1:2]).permute(0, 2, 1).cpu().numpy()
    np.testing.assert_allclose(one[0], got[1], atol=1e-5 if prec in EXACT else 1e-3)


@pytest.mark.parametrize("prec", ["fp32", "exact", "bf16", "f16"])
@pytest.mark.parametrize("name", ["greedy_bos", "greedy_task"])
def test_greedy_search_matches_reference_fixture(name, prec, engines, synth_weights):
    """conette_greedy against the masked step logits produced by the reference's greedy_search."""
    g = np.load(os.path.join(G.GOLDEN, "forcing", name + ".npz"))
    eng = engines[prec]
    fe = torch.from_numpy(g["frame_embs"]).cuda()
    lens = torch.from_numpy(g["audio_shape"][:, 1].astype(np.int32))
    bos = torch.full((fe.shape[0],), int(g["bos_id"]), dtype=torch.int32)
    fm = synth_weights["model.forbid_rep_mask"] if int(g["use_forbid"]) else None
    out = eng.greedy(fe, lens, bos, fm, int(g["min_pred"]), int(g["max_pred"]))
    got = out["logits"].permute(0, 2, 1).cpu()                      # (B, V, pred_size) like the reference
    ref = torch.from_numpy(g["logits"])
    if prec in EXACT:
        assert tuple(got.shape) == tuple(ref.shape)
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin)                # EOS floor, forbid-repeat, finished-clip fill
        np.testing.assert_allclose(got[fin].numpy(), ref[fin].numpy(), rtol=1e-3, atol=2e-3)
        assert torch.equal(got.argmax(dim=1), ref.argmax(dim=1))
        assert torch.equal(out["preds"].cpu().long(), got.argmax(dim=1))  # the chain itself; pad (= arg-max of the fill) after <eos>
    else:
        # bf16: the arg-max chain may leave the reference's at a near-tie; compare the common prefix of every clip
        steps = min(got.shape[2], ref.shape[2])
        ga, ra = got.argmax(dim=1), ref.argmax(dim=1)
        n_cmp = 0
        for b in range(ga.shape[0]):
            for i in range(steps):
                fin_r = torch.isfinite(ref[b, :, i])
                if not torch.equal(torch.isfinite(got[b, :, i]), fin_r):
                    break
                np.testing.assert_allclose(got[b, :, i][fin_r].numpy(), ref[b, :, i][fin_r].numpy(), atol=0.8 * R16[prec])
                n_cmp += 1
                if ga[b, i] != ra[b, i]:
                    break
        print(f"greedy {name}/{prec}: {n_cmp} of {ga.shape[0] * steps} clip-steps compared before a clip's first flipped arg-max")
        assert n_cmp >= GREEDY_FLOOR[prec] * ga.shape[0] * steps, (n_cmp, ga.shape[0] * steps)


def test_fused_decoder_kernels_repeatable_under_load(engines, synth_weights):
    """The fused decoder kernels count their outstanding loads by hand (dec_block.h, dec_ffn.h): a wrong count would
    show up as run-to-run differences once timing changes.  Eager (non-graph) decodes while another stream runs encoder
    work of varying shape must reproduce the first result bit for bit."""
    from conette_amd import synth
    eng = engines["bf16"]
    B = 48
    wave = torch.from_numpy(synth.synth_waveforms(B, 160000, 77)).cuda()
    fe, _ = eng.encode(wave)
    lens = torch.full((B,), fe.shape[1], dtype=torch.int32)
    bos = synth_weights["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)]
    forbid = synth_weights["model.forbid_rep_mask"]
    eng.set_decode_graph(False)
    try:
        ref = eng.decode(fe, lens, bos, forbid, 3, 3, 20, want_trace=True)
        s2 = torch.cuda.Stream()
        for it in range(40):
            with torch.cuda.stream(s2):
                eng.encode(wave[: 8 + (it % 5) * 8].contiguous())
            out = eng.decode(fe, lens, bos, forbid, 3, 3, 20, want_trace=True)
            for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs", "trace_val"):
                assert torch.equal(out[k], ref[k]), (it, k)
        torch.cuda.synchronize()
    finally:
        eng.set_decode_graph(True)


def test_c_abi_error_behaviour(engines):
    """Status codes + conette_last_error instead of exceptions / aborts: bad arguments, unsupported sizes and short
    workspaces are refused before anything is launched (header: 0 ok, 1 argument, 4 workspace)."""
    import ctypes as C
    eng = engines["fp32"]
    lib, ctx = eng.lib, eng._ctx
    b, t = 2, 7
    fe = torch.zeros((b, t, 768), device="cuda")
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    bos = torch.full((b,), 5624, dtype=torch.int32, device="cuda")
    out_i = torch.zeros((b * 3 * 64,), dtype=torch.int32, device="cuda")
    out_f = torch.zeros((b * 3 * 64,), dtype=torch.float32, device="cuda")
    sizes = torch.zeros((2,), dtype=torch.int32, device="cuda")
    need = lib.conette_decode_workspace_bytes(ctx, b, t, 3, 20)
    ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
    p = lambda x: C.c_void_p(x.data_ptr())

    def dec(beam, max_pred, ws_bytes, fe_ptr=p(fe)):
        return lib.conette_decode(ctx, fe_ptr, p(lens), p(bos), None, b, t, beam, 3, max_pred, p(out_i), p(out_f), p(out_i),
                                  p(out_f), p(sizes), None, None, None, None, p(ws), ws_bytes, None)

    assert dec(3, 20, need) == 0
    assert dec(17, 20, need) == 1 and b"beam" in lib.conette_last_error()         # CN_MAX_BEAM = 16
    assert dec(3, 65, need) == 1                                                     # CN_MAX_PRED = 64
    assert dec(3, 20, need - 1) == 4 and b"workspace" in lib.conette_last_error()
    assert dec(3, 20, need, fe_ptr=None) == 1
    assert lib.conette_forcing(ctx, p(fe), p(lens), p(bos), b, t, 0, p(out_f), p(ws), need, None) == 1
    torch.cuda.synchronize()


def test_encode_short_and_odd_lengths(engines):
    """Edge geometry: very short clips (a single frame row in the last stage) and lengths that leave partial tiles in
    every stage; bf16 must track fp32 and everything stays finite."""
    from conette_amd import synth
    with pytest.raises(RuntimeError, match="too short"):          # 0.2 s: no audio frame survives the downsamplings
        engines["fp32"].encode(torch.zeros((1, 6400)).cuda())
    for L in (7680, 9600, 16000, 20481, 33333, 47999):
        wave = torch.from_numpy(synth.synth_waveforms(3, L, 31 + L, lengths=[L, max(L // 2, 3200), max(L - 777, 3200)])).cuda()
        f32, c32 = engines["fp32"].encode(wave)
        f16, c16 = engines["bf16"].encode(wave)
        assert f32.shape == f16.shape and f32.shape[1] == engines["fp32"].lib.conette_num_audio_frames(L)
        assert torch.isfinite(f32).all() and torch.isfinite(f16).all()
        fx, cx = engines["exact"].encode(wave)   # (ragged tiles of the fused hi / lo blocks and the split GEMMs)
        np.testing.assert_allclose(fx.cpu().numpy(), f32.cpu().numpy(), rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(cx.cpu().numpy(), c32.cpu().numpy(), rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(f16.float().cpu().numpy(), f32.cpu().numpy(), atol=0.1, rtol=0.1)
        np.testing.assert_allclose(c16.cpu().numpy(), c32.cpu().numpy(), atol=0.05)

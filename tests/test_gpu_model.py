"""GPU: the drop-in Python API (CoNeTTEModel) end to end on the golden scenarios + API cases,
and size-independent properties at the benchmark batch (B=64)."""
import json
import os

import numpy as np
import pytest
import torch

from conette_amd import synth
from tests import golden_util as G

pytestmark = pytest.mark.gpu
TAGS = {i: f"tag{i}" for i in range(527)}


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    from conette_amd import synth
    return synth.write_pretrained_dir(str(tmp_path_factory.mktemp("conette_synth")))


@pytest.fixture(scope="module")
def model_fp32(model_dir):
    from conette_amd import CoNeTTEConfig, CoNeTTEModel
    config = CoNeTTEConfig.from_pretrained(model_dir)
    # (stop-words of the synthetic vocabulary: what the golden generator's NLTK stand-in hands the reference)
    return CoNeTTEModel.from_pretrained(model_dir, config=config, precision="fp32", offline=True,
                                        audioset_idx_to_name=TAGS, stopwords=synth.synth_stopwords())


@pytest.fixture(scope="module")
def model_exact(model_dir):
    from conette_amd import CoNeTTEModel
    return CoNeTTEModel.from_pretrained(model_dir, precision="exact", offline=True, audioset_idx_to_name=TAGS,
                                        stopwords=synth.synth_stopwords())


@pytest.fixture(scope="module")
def model_bf16(model_dir):
    from conette_amd import CoNeTTEModel
    return CoNeTTEModel.from_pretrained(model_dir, precision="bf16", offline=True, audioset_idx_to_name=TAGS)


@pytest.mark.parametrize("prec", ["fp32", "exact"])
@pytest.mark.parametrize("name", G.SCENARIOS)
def test_model_forward_matches_reference_fixture(name, prec, model_fp32, model_exact):
    """the whole operator API in both precisions whose ids must equal the reference's: fp32 MFMA and "exact" (fp16 hi / lo pairs)"""
    g = G.load(name)
    x, kw = G.inputs(g)
    out = (model_fp32 if prec == "fp32" else model_exact)(x, sr=32000, **kw)
    assert out["preds"].dtype == torch.long and out["mult_preds"].dtype == torch.long
    assert out["preds"].cpu().tolist() == g["preds"].tolist()
    assert out["mult_preds"].cpu().tolist() == g["mult_preds"].tolist()
    np.testing.assert_allclose(out["lprobs"].cpu().numpy(), g["lprobs"], atol=2e-4)
    np.testing.assert_allclose(out["mult_lprobs"].cpu().numpy(), g["mult_lprobs"], atol=2e-4)
    assert out["cands"] == json.loads(str(g["cands"]))
    assert out["mult_cands"] == json.loads(str(g["mult_cands"]))
    assert out["tasks"] == json.loads(str(g["tasks"]))
    np.testing.assert_allclose(out["tags_probs"].cpu().numpy(), g["tags_probs"], rtol=1e-3, atol=1e-4)
    # tags: identical wherever the probability is not within 1e-3 of the 0.3 threshold
    ref_tags = json.loads(str(g["tags"]))
    for b, row in enumerate(g["tags_probs"]):
        safe = {f"tag{i}" for i, p in enumerate(row) if abs(p - 0.3) > 1e-3}
        assert set(out["tags"][b]) & safe == set(ref_tags[b]) & safe


def test_api_cases(model_fp32):
    from conette_amd import synth
    with open(os.path.join(G.GOLDEN, "api_cases.json")) as f:
        api = json.load(f)
    m = model_fp32
    wav = torch.from_numpy(synth.synth_waveforms(2, 64000, 7000))
    assert m.default_task == api["default_task"] and m.tasks == api["tasks"]
    assert m(wav[0], sr=32000)["preds"].cpu().tolist() == api["rank1_preds"]
    assert m(wav[0][None], sr=32000)["preds"].cpu().tolist() == api["rank2_preds"]
    assert m(wav[0:1, None], sr=32000)["preds"].cpu().tolist() == api["rank3_preds"]
    o4 = m(torch.stack([wav[0], wav[1]], 0), sr=32000)
    assert o4["preds"].cpu().tolist() == api["stereo_preds"]
    o5 = m([wav[0][None], wav[1][None]], sr=[32000, 32000], task=["clotho", "audiocaps"])
    assert o5["preds"].cpu().tolist() == api["list_preds"]
    assert o5["tasks"] == api["list_tasks"] and o5["cands"] == api["list_cands"]
    pre = m.preprocessor(wav[:, None], 32000, None)
    o6 = m(pre["audio"], x_shapes=pre["audio_shape"], preprocess=False, task="clotho")
    assert o6["preds"].cpu().tolist() == api["nopre_preds"] and sorted(o6.keys()) == api["nopre_keys"]
    o7 = m(wav[:, None], sr=32000, task="clotho")
    assert o7["preds"].cpu().tolist() == api["pre_preds"] and sorted(o7.keys()) == api["full_keys"]
    with pytest.raises(ValueError) as e:
        m(wav[0], sr=32000, task="not_a_task")
    assert str(e.value) == api["bad_task_error"]
    with pytest.raises(ValueError) as e:
        m(wav[:, None], sr=32000, task=["clotho"])
    assert str(e.value) == api["bad_ntasks_error"]
    with pytest.raises(ValueError):
        m(wav[0], sr=16000, x_shapes=torch.tensor([[64000]]))
    # 16 kHz input through the HIP resampler
    assert m(wav[0][::2].contiguous(), sr=16000)["preds"].cpu().tolist() == api["sr16k_preds"]
    # test_inference.py semantics: str / list types, two tasks, forbid_rep_mode="none", beam 1 + tags
    out = m(wav[0], sr=32000, task="clotho", forbid_rep_mode="none", beam_size=1)
    assert isinstance(out["cands"][0], str) and isinstance(out["tags"], list)


def test_wav_file_input(model_fp32, tmp_path):
    """A path input goes through the PCM reader; 16-bit quantisation only perturbs the input."""
    import wave
    from conette_amd import synth
    wav = synth.synth_waveforms(1, 48000, 31)[0]
    pcm = np.clip(np.round(wav * 32768.0), -32768, 32767).astype("<i2")
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(32000)
        w.writeframes(pcm.tobytes())
    out_path = model_fp32(p)
    out_tensor = model_fp32(torch.from_numpy(pcm.astype(np.float32) / 32768.0), sr=32000)
    assert out_path["preds"].cpu().tolist() == out_tensor["preds"].cpu().tolist()
    assert isinstance(out_path["cands"][0], str)


def test_properties_at_benchmark_batch(model_bf16):
    """BASELINE configs[1]/[2] sizes (B=64, 10 s): size-independent properties of the path."""
    from conette_amd import synth
    eng = model_bf16.engine
    B = 64
    wave = torch.from_numpy(synth.synth_waveforms(B, 320000, 1234)).cuda()
    fe, clip = eng.encode(wave)
    fe2, clip2 = eng.encode(wave)
    assert torch.equal(fe, fe2) and torch.equal(clip, clip2)                  # deterministic
    assert torch.isfinite(fe).all() and ((clip >= 0) & (clip <= 1)).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    fe_p, _ = eng.encode(wave[perm].contiguous())
    assert torch.equal(fe_p, fe[perm])                                         # clips are independent
    fe_s, _ = eng.encode(wave[:7].contiguous())
    assert torch.equal(fe_s, fe[:7])                                           # and batch-size invariant
    lens = torch.full((B,), fe.shape[1], dtype=torch.int32)
    bos = model_bf16.task_id_to_token_id[torch.zeros(B, dtype=torch.long)]
    o3 = eng.decode(fe, lens, bos, model_bf16.forbid_rep_mask, 3, 3, 20)
    o3b = eng.decode(fe, lens, bos, model_bf16.forbid_rep_mask, 3, 3, 20)     # 2nd/3rd call: graph replay
    o3c = eng.decode(fe, lens, bos, model_bf16.forbid_rep_mask, 3, 3, 20)
    for k in ("best_preds", "best_lprobs", "mult_preds", "mult_lprobs", "sizes"):
        assert torch.equal(o3[k], o3b[k]) and torch.equal(o3[k], o3c[k]), k
    ps, bm = (int(v) for v in o3["sizes"].tolist())
    mp, ml = o3["mult_preds"].cpu(), o3["mult_lprobs"].cpu()
    bp, bl = o3["best_preds"].cpu(), o3["best_lprobs"].cpu()
    assert torch.equal(bl, ml.max(dim=1).values)                               # best = max averaged log-prob
    assert torch.equal(bp, mp[torch.arange(B), ml.argmax(dim=1)])
    assert (mp[:, :, ps:] == 0).all() and 1 <= bm <= ps <= 20
    for row in mp.reshape(-1, 20).tolist():                                    # EOS floor, pad after EOS
        if 2 in row:
            e = row.index(2)
            assert e >= 3 and all(t == 0 for t in row[e + 1:])
        assert all(t not in (1, 3) and t < 5624 for t in row)                 # never <bos>/<unk>/task tokens
    forbid = model_bf16.forbid_rep_mask.cpu()
    for row in mp.reshape(-1, 20).tolist():                                    # content words never repeat
        seen = [t for t in row if t > 2 and forbid[t]]
        assert len(seen) == len(set(seen))
    o1 = eng.decode(fe, lens, bos, model_bf16.forbid_rep_mask, 1, 3, 20)
    assert torch.equal(o1["best_preds"], o1["mult_preds"][:, 0])               # beam 1 == greedy chain
    assert torch.equal(o1["best_lprobs"], o1["mult_lprobs"][:, 0])


def test_predict_cli_and_offline_features(model_dir, model_fp32, tmp_path):
    """conette-predict row format / CSV, and the offline per-file feature producer
    feeding the preprocess=False entry."""
    import csv
    import wave
    import conette_amd
    from conette_amd import synth
    from conette_amd.predict import main_predict
    paths = []
    for i in range(2):
        wav = synth.synth_waveforms(1, 40000 + 8000 * i, 77 + i)[0]
        pcm = np.clip(np.round(wav * 32768.0), -32768, 32767).astype("<i2")
        p = str(tmp_path / f"clip{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1), w.setsampwidth(2), w.setframerate(32000)
            w.writeframes(pcm.tobytes())
        paths.append(p)
    # the CLI needs the AudioSet CSV like the reference (class_labels_indices.csv in its cache directory)
    cache = tmp_path / "audioset_mapping"
    cache.mkdir()
    with open(cache / "class_labels_indices.csv", "w") as f:
        f.write("index,mid,display_name\n" + "".join(f"{i},/m/{i},tag{i}\n" for i in range(527)))
    os.environ["CONETTE_AUDIOSET_CACHE"] = str(cache)
    try:
        out_csv = str(tmp_path / "out.csv")
        res = main_predict(["--audio", *paths, "--task", "audiocaps", "--model_name", model_dir, "--precision", "fp32",
                            "--csv_export", out_csv, "--verbose", "0"])
    finally:
        os.environ.pop("CONETTE_AUDIOSET_CACHE", None)
    ref = model_fp32(paths, task="audiocaps")
    assert [r["candidate"] for r in res] == ref["cands"]
    rows = list(csv.DictReader(open(out_csv)))
    assert [r["audio"] for r in rows] == ["clip0.wav", "clip1.wav"] and rows[0]["task"] == "audiocaps"
    assert list(rows[0].keys()) == ["audio", "task", "candidate"]
    # offline producer (batch = 1 per file) -> preprocess=False consumer == online path on the same single clip
    tr = conette_amd.get_offline_transform(model_fp32)
    wav0 = torch.from_numpy(synth.synth_waveforms(1, 48000, 5))
    feats = tr(wav0, 32000)
    assert feats["audio"].shape[1] == 768 and feats["audio_shape"].tolist() == [768, feats["audio"].shape[0]]
    off = model_fp32(feats["audio"][None], x_shapes=feats["audio_shape"][None], preprocess=False, task="clotho")
    on = model_fp32(wav0, sr=32000, task="clotho")
    assert off["preds"].cpu().tolist() == on["preds"].cpu().tolist()


def test_long_clips_long_captions_block_path(model_bf16, model_fp32):
    """Clips longer than 32 audio frames and captions longer than the prefetched self-attention window: the
    remainder loops of the fused decoder kernels (dec_block.h, dec_ffn.h).  Cross-checked against the same bf16
    engine with CONETTE_OPT_DECODE_FUSION = 0 (one launch per sub-layer: the kernels the fp32 golden tests pin),
    step by step while both searches are on the same branch, and against the fp32 engine on the scores."""
    from conette_amd import synth
    B, L = 12, 15 * 32000
    lengths = [L - 7000 * i for i in range(B)]
    wave = torch.from_numpy(synth.synth_waveforms(B, L, 4321, lengths=lengths)).cuda()
    e16, e32 = model_bf16.engine, model_fp32.engine
    fe32, _ = e32.encode(wave)
    fe16, _ = e16.encode(wave)
    assert fe32.shape[1] > 32
    np.testing.assert_allclose(fe16.float().cpu().numpy(), fe32.cpu().numpy(), atol=0.08, rtol=0.08)
    lens = torch.tensor([e32.lib.conette_num_audio_frames(n) for n in lengths], dtype=torch.int32)
    bos = model_fp32.task_id_to_token_id[torch.zeros(B, dtype=torch.long)]
    forbid = model_fp32.forbid_rep_mask
    beam, min_pred, max_pred = 3, 27, 30
    fused = e16.decode(fe32, lens, bos, forbid, beam, min_pred, max_pred, want_trace=True)
    e16.set_decode_fusion(False)
    try:
        plain = e16.decode(fe32, lens, bos, forbid, beam, min_pred, max_pred, want_trace=True)
    finally:
        e16.set_decode_fusion(True)
    sel_f, sel_p = fused["trace_sel"].cpu().numpy(), plain["trace_sel"].cpu().numpy()
    val_f, val_p = fused["trace_val"].cpu().numpy(), plain["trace_val"].cpu().numpy()
    n_same_steps, n_steps = 0, 0
    for clip in range(B):
        for step in range(max_pred):
            if sel_p[step, clip, 0, 0] < 0:
                break
            n_steps += 1
            if not np.array_equal(sel_f[step, clip], sel_p[step, clip]):
                break  # a near-tie resolved the other way: later steps are no longer comparable
            live = sel_p[step, clip, :, 0] >= 0
            np.testing.assert_allclose(val_f[step, clip][live], val_p[step, clip][live], atol=0.08 * (step + 1))  # two bf16 roundings of the same sums
            n_same_steps += 1
    assert n_same_steps >= 0.8 * n_steps, (n_same_steps, n_steps)
    p16 = fused["best_preds"].cpu()
    assert all(int((row != 0).sum()) >= 28 for row in p16)                       # EOS floor held until step 27
    o32 = e32.decode(fe32, lens, bos, forbid, beam, min_pred, max_pred)
    same = [b for b in range(B) if torch.equal(o32["best_preds"].cpu()[b], p16[b])]
    np.testing.assert_allclose(fused["best_lprobs"].cpu().numpy()[same], o32["best_lprobs"].cpu().numpy()[same], atol=0.05)
    assert np.all(np.abs(fused["best_lprobs"].cpu().numpy() - o32["best_lprobs"].cpu().numpy()) < 0.5)
    # batch invariance of the fused path (4-row blocks, padded rows, frame masks)
    sub = e16.decode(fe32[:5].contiguous(), lens[:5], bos[:5], forbid, beam, min_pred, max_pred)
    assert torch.equal(sub["best_preds"].cpu()[:, :28], p16[:5, :28])
    np.testing.assert_allclose(sub["best_lprobs"].cpu().numpy(), fused["best_lprobs"].cpu().numpy()[:5], atol=1e-5)


def test_decode_at_256_clips(model_bf16):
    """BASELINE configs[2] size: 768 decoder rows.  Tiling the 64-clip batch four times must reproduce its
    captions four times (rows are independent in every decode kernel)."""
    from conette_amd import synth
    eng = model_bf16.engine
    wave = torch.from_numpy(synth.synth_waveforms(64, 320000, 1234)).cuda()
    fe, _ = eng.encode(wave)
    lens = torch.full((64,), fe.shape[1], dtype=torch.int32)
    bos = model_bf16.task_id_to_token_id[torch.zeros(64, dtype=torch.long)]
    o64 = eng.decode(fe, lens, bos, model_bf16.forbid_rep_mask, 3, 3, 20)
    fe4 = fe.repeat(4, 1, 1).contiguous()
    o256 = eng.decode(fe4, lens.repeat(4), bos.repeat(4), model_bf16.forbid_rep_mask, 3, 3, 20)
    w = min(o64["best_preds"].shape[1], o256["best_preds"].shape[1])
    for q in range(4):
        assert torch.equal(o256["best_preds"][64 * q:64 * (q + 1), :w], o64["best_preds"][:, :w]), q
        assert torch.equal(o256["best_lprobs"][64 * q:64 * (q + 1)], o64["best_lprobs"]), q


def test_model_teacher_forcing_api(model_fp32):
    """CoNeTTEModel.teacher_forcing mirrors CoNeTTEPLM.decode_audio(..., "forcing", caps_in=...): waveform input,
    (B, vocab, cap_len) output, the reference's BOS check."""
    from conette_amd import synth
    g = np.load(os.path.join(G.GOLDEN, "forcing", "forcing_ragged.npz"))
    n = [int(v) for v in g["lengths"]]
    wav = synth.synth_waveforms(len(n), max(n), int(g["seed0"]), lengths=n)
    x = [torch.from_numpy(wav[i, : n[i]].copy())[None, :] for i in range(len(n))]
    caps = torch.from_numpy(g["caps_in"])
    logits = model_fp32.teacher_forcing(x, caps, sr=32000)
    assert tuple(logits.shape) == tuple(g["logits"].shape)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=2e-3, atol=5e-3)
    pre = {"audio": torch.from_numpy(g["frame_embs"]), "audio_shape": torch.from_numpy(g["audio_shape"])}
    logits2 = model_fp32.teacher_forcing(pre, caps, preprocess=False)
    np.testing.assert_allclose(logits2.cpu().numpy(), g["logits"], rtol=1e-3, atol=2e-3)
    bad = caps.clone()
    bad[0, 0] = 1
    with pytest.raises(ValueError, match="BOS was not replaced"):
        model_fp32.teacher_forcing(pre, bad, preprocess=False)


def test_model_greedy_search_api(model_fp32):
    """CoNeTTEModel.greedy_search mirrors nn/decoding/greedy.py: (B, vocab, pred_size) masked logits."""
    g = np.load(os.path.join(G.GOLDEN, "forcing", "greedy_task.npz"))
    pre = {"audio": torch.from_numpy(g["frame_embs"]), "audio_shape": torch.from_numpy(g["audio_shape"])}
    lg = model_fp32.greedy_search(pre, preprocess=False, bos_id=int(g["bos_id"]), min_pred_size=int(g["min_pred"]),
                                  max_pred_size=int(g["max_pred"])).cpu()
    ref = torch.from_numpy(g["logits"])
    assert tuple(lg.shape) == tuple(ref.shape)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(lg), fin)
    np.testing.assert_allclose(lg[fin].numpy(), ref[fin].numpy(), rtol=1e-3, atol=2e-3)


def test_mixed_length_1s_to_30s_against_oracle(model_fp32, model_bf16):
    """BASELINE configs[4] shape: a ragged batch from 1 s to 30 s (3 ... 94 audio frames, zero-padded to the longest,
    frame masks in the decoder).  fp32 engine against the CPU oracle run on the same inputs: identical ids unless the
    oracle itself reports a near-tie, scores to 1e-3; bf16: valid captions, scores within the bf16 tolerance on the
    clips whose ids agree."""
    from conette_amd import synth
    from oracle import cpu_ref as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    lengths = [32000, 394321, 960000]
    wav = synth.synth_waveforms(len(lengths), max(lengths), 5150, lengths=lengths)
    x = [torch.from_numpy(wav[i, : lengths[i]].copy())[None, :] for i in range(len(lengths))]
    w = O.to_torch(synth.synth_state_dict())
    trace = []
    with torch.no_grad():
        ref = O.model_forward(w, synth.synth_config_dict(), x, sr=32000, task="clotho", trace=trace)
    out = model_fp32(x, sr=32000, task="clotho")
    min_margin = min(d["margin"] for step in trace for d in step)
    if min_margin > 1e-3:
        assert out["preds"].cpu().tolist() == ref["preds"].tolist()
        np.testing.assert_allclose(out["lprobs"].cpu().numpy(), ref["lprobs"].numpy(), atol=1e-3)
    else:  # the oracle's own top-k was a near-tie somewhere: only the scores are comparable
        assert np.all(np.abs(out["lprobs"].cpu().numpy() - ref["lprobs"].numpy()) < 0.5)
    out16 = model_bf16(x, sr=32000, task="clotho")
    p16, p32 = out16["preds"].cpu(), out["preds"].cpu()
    w_ = min(p16.shape[1], p32.shape[1])
    same = [b for b in range(len(lengths)) if torch.equal(p16[b, :w_], p32[b, :w_])]
    np.testing.assert_allclose(out16["lprobs"].cpu().numpy()[same], out["lprobs"].cpu().numpy()[same], atol=0.05)
    assert all(len(c) > 0 for c in out16["cands"])


def test_decode_audio_surface_baseline_style(model_fp32):
    """decode_audio(encoder_outs, method): the surface shared by CoNeTTEPLM and BaselinePLM.  "generate" with the plain
    <bos> first token (BaselinePLM: no task token) against the CPU oracle's beam search; "forcing" / "greedy" route
    to the paths pinned by the reference fixtures; unknown methods raise like the reference."""
    from conette_amd import synth
    from oracle import cpu_ref as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = np.load(os.path.join(G.GOLDEN, "forcing", "greedy_bos.npz"))
    pre = {"audio": torch.from_numpy(g["frame_embs"]), "audio_shape": torch.from_numpy(g["audio_shape"])}
    w = O.to_torch(synth.synth_state_dict())
    mem, mask = O.encode_audio(w, pre["audio"], pre["audio_shape"])
    bos = torch.full((mem.shape[0],), int(g["bos_id"]), dtype=torch.long)
    trace = []
    ref = O.generate(w, mem, mask, bos, vocab_size=w["model.decoder.classifier.weight"].shape[0], beam_size=2,
                     min_pred_size=3, max_pred_size=10, forbid_rep_mask=w["model.forbid_rep_mask"], trace=trace)
    preds, lprobs, mult_preds, mult_lprobs = model_fp32.decode_audio(pre, "generate", beam_size=2, min_pred_size=3,
                                                                    max_pred_size=10)
    if min(d["margin"] for step in trace for d in step) > 1e-3:
        assert preds.cpu().tolist() == ref[0].tolist() and mult_preds.cpu().tolist() == ref[2].tolist()
    np.testing.assert_allclose(lprobs.cpu().numpy(), ref[1].numpy(), atol=1e-3)
    lg = model_fp32.decode_audio(pre, "greedy", min_pred_size=int(g["min_pred"]), max_pred_size=int(g["max_pred"])).cpu()
    fin = torch.isfinite(torch.from_numpy(g["logits"]))
    np.testing.assert_allclose(lg[fin].numpy(), g["logits"][fin.numpy()], rtol=1e-3, atol=2e-3)
    with pytest.raises(ValueError, match="caps_in"):
        model_fp32.decode_audio(pre, "forcing")
    with pytest.raises(ValueError, match="Unknown argument"):
        model_fp32.decode_audio(pre, "sampling")


def _write_wav(path, data, sr):
    """data: (channels, samples) float in [-1, 1) -> 16-bit PCM WAV."""
    import wave
    pcm = np.clip(np.round(data.T * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(data.shape[0]), w.setsampwidth(2), w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def test_predict_cli_on_44k_stereo_against_oracle(model_dir, tmp_path, synth_weights, synth_cfg):
    """SURVEY 8f-1 end to end the way the reference's smoke test runs it (tests/test_inference.py:20-27: a 44.1 kHz
    file through the model): stereo 44.1 kHz + mono 48 kHz WAVs -> conette-predict --csv_export -> candidates equal the
    CPU oracle's on the same files (load, 441:320 / 3:2 sinc resampling on the GPU, channel mean, pad, captions)."""
    import csv
    from conette_amd import synth
    from conette_amd.predict import main_predict
    from oracle import cpu_ref as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = synth.synth_waveforms(2, 44100 * 4, 808)
    b = synth.synth_waveforms(1, 48000 * 3, 809)
    pa, pb = str(tmp_path / "stereo_44k.wav"), str(tmp_path / "mono_48k.wav")
    _write_wav(pa, np.stack([a[0], 0.5 * a[1]]), 44100)
    _write_wav(pb, b, 48000)
    cache = tmp_path / "audioset_mapping"
    cache.mkdir()
    with open(cache / "class_labels_indices.csv", "w") as f:
        f.write("index,mid,display_name\n" + "".join(f"{i},/m/{i},tag{i}\n" for i in range(527)))
    os.environ["CONETTE_AUDIOSET_CACHE"] = str(cache)
    try:
        out_csv = str(tmp_path / "out.csv")
        res = main_predict(["--audio", pa, pb, "--task", "clotho", "--model_name", model_dir, "--precision", "fp32",
                            "--csv_export", out_csv, "--verbose", "0"])
    finally:
        os.environ.pop("CONETTE_AUDIOSET_CACHE", None)
    with torch.no_grad():
        ref = O.model_forward(synth_weights, synth_cfg, [pa, pb], task="clotho")
    assert [r["candidate"] for r in res] == ref["cands"]
    with open(out_csv) as f:
        rows = list(csv.DictReader(f))
    assert [r["candidate"] for r in rows] == ref["cands"] and [r["audio"] for r in rows] == ["stereo_44k.wav", "mono_48k.wav"]
    assert [r["task"] for r in rows] == ["clotho", "clotho"]


def test_offline_feature_files_against_oracle(model_fp32, tmp_path, synth_weights, synth_cfg):
    """per-file (batch of one) features on disk -- audio (T, 768), audio_shape, clip_probs -- equal the
    oracle's preprocessor on each single file, and the stored rows fed back through preprocess=False decode to the
    captions the oracle gets from the same features (transforms/get.py:240-310, conf/dm/hdf.yaml:12-14, model.py:205-212)."""
    from conette_amd import offline, synth
    from oracle import cpu_ref as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    files = []
    for i, (sr, secs, ch) in enumerate([(44100, 3.0, 2), (32000, 5.5, 1), (16000, 2.0, 1)]):
        w = synth.synth_waveforms(ch, int(sr * secs), 300 + i)
        p = str(tmp_path / f"f{i}.wav")
        _write_wav(p, w, sr)
        files.append(p)
    paths = offline.write_features(model_fp32, files, str(tmp_path / "feats"))
    assert sorted(os.listdir(tmp_path / "feats")) == ["f0.npz", "f1.npz", "f2.npz", "index.json"]
    for f, p in zip(files, paths):
        row = np.load(p)
        with torch.no_grad():
            ref = O.preprocessor_forward(synth_weights, f)
        assert row["audio"].shape == tuple(ref["audio"][0].shape) and row["audio"].shape[1] == 768
        assert row["audio_shape"].tolist() == ref["audio_shape"][0].tolist() == [768, row["audio"].shape[0]]
        np.testing.assert_allclose(row["audio"], ref["audio"][0].numpy(), rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(row["clip_probs"], ref["clip_probs"][0].numpy(), rtol=1e-3, atol=1e-4)
    audio, shapes = offline.load_features(paths)
    assert audio.shape[0] == 3 and audio.shape[2] == 768 and shapes.shape == (3, 2)
    out = model_fp32(audio, x_shapes=shapes, preprocess=False, task="clotho")
    with torch.no_grad():
        ref = O.model_forward(synth_weights, synth_cfg, audio, x_shapes=shapes, preprocess=False, task="clotho")
    assert out["preds"].cpu().tolist() == ref["preds"].tolist() and out["cands"] == ref["cands"]
    np.testing.assert_allclose(out["lprobs"].cpu().numpy(), ref["lprobs"].numpy(), atol=2e-4)

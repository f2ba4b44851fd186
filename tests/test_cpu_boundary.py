"""CPU: the C-ABI library loads and exports every symbol include/conette_hip.h declares; host
logic (config / tokenizer / input normalisation helpers / synthetic checkpoint recipe)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from conette_amd import engine
    return engine.load_library()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "conette_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(conette_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 16
    from conette_amd import engine
    assert declared == set(engine.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.conette_abi_version() == 3


def test_geometry_helpers(lib):
    # 10 s -> 1001 STFT frames -> 31 encoder frames; 30 s -> 94; 5 s -> 15; 1 s -> 3
    assert lib.conette_num_frames(320000) == 1001
    assert [lib.conette_num_audio_frames(int(s * 32000)) for s in (10, 30, 5, 1)] == [31, 94, 15, 3]
    assert lib.conette_resample_len(44100, 44100, 32000) == 32000
    assert lib.conette_resample_len(1000, 16000, 32000) == 2000


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from conette_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine({"model.decoder.classifier.weight": torch.zeros(8, 256)})


def test_config_roundtrip(tmp_path):
    from conette_amd import CoNeTTEConfig, synth
    cfg = CoNeTTEConfig(**synth.synth_config_dict(n_words=50))
    cfg.save_pretrained(str(tmp_path))
    cfg2 = CoNeTTEConfig.from_pretrained(str(tmp_path))
    assert cfg2.beam_size == 3 and cfg2.min_pred_size == 3 and cfg2.max_pred_size == 20
    assert list(cfg2.task_names) == list(synth.TASK_NAMES)
    assert cfg2.tokenizer_state["tokenizer"]["itos"]["4"] == "w4"  # JSON turns int keys into str


def test_tokenizer_decode_and_normalisers():
    from conette_amd import synth
    from conette_amd.tokenizer import AACTokenizer, unpickle_extra_state
    st = synth.synth_tokenizer_state(n_words=20)
    st["tokenizer"]["itos"][10] = ","
    st["tokenizer"]["itos"][11] = "-"
    st["tokenizer"]["itos"][12] = "Rain"
    tok = AACTokenizer.from_txt_state(json.loads(json.dumps(st)))  # through JSON: str keys
    assert tok.is_fit() and tok.get_vocab_size() == 4 + 20 + 7
    assert tok.decode_rec(torch.tensor([[1, 4, 10, 5, 2, 0, 0]])) == ["w4, w5"]
    assert tok.decode_rec(torch.tensor([12, 11, 5, 2])) == "rain-w5"
    assert tok.decode_rec(torch.tensor([[[4, 2], [5, 2]]])) == [["w4", "w5"]]
    assert tok.token_to_id("<bos_clotho>") == 24
    extra = unpickle_extra_state(torch.from_numpy(synth.extra_state_tensor(20)))
    assert extra["model.tokenizers.0._extra_state"]["tokenizer"]["itos"][4] == "w4"


def test_extra_state_unpickler_refuses_code():
    import pickle
    from conette_amd.tokenizer import unpickle_extra_state
    evil = torch.frombuffer(bytearray(pickle.dumps(os.system)), dtype=torch.uint8)
    with pytest.raises(pickle.UnpicklingError):
        unpickle_extra_state(evil)


def test_frame_embs_lens_matches_reference_rule():
    from conette_amd.preprocessor import frame_embs_lens
    # 5 s in a 10 s batch -> 16; torch round-half-even on fp32
    lens = frame_embs_lens(torch.tensor([320000, 160000, 64000]), 320000, 31)
    assert lens.tolist() == [31, 16, 6] and lens.dtype == torch.int32


def test_synth_recipe_is_deterministic():
    from conette_amd import synth
    a = synth.synth_state_dict(n_words=30)
    b = synth.synth_state_dict(n_words=30)
    assert len(a) == 306 and all(np.array_equal(a[k], b[k]) for k in a)
    w1 = synth.synth_waveforms(2, 4000, 5, lengths=[4000, 1000])
    assert np.array_equal(w1, synth.synth_waveforms(2, 4000, 5, lengths=[4000, 1000]))
    assert (w1[1, 1000:] == 0).all() and w1.dtype == np.float32


def test_wav_reader_formats(tmp_path):
    """load_audio stands in for torchaudio.load: PCM 8 / 16 / 24 / 32 bit and IEEE float WAV files, plain and
    WAVE_FORMAT_EXTENSIBLE headers, mono and stereo; other containers are an explicit error without a back-end."""
    import struct
    import numpy as np
    import pytest
    import conette_amd  # noqa: F401
    from conette_amd.preprocessor import load_audio
    rng = np.random.default_rng(0)
    x = (rng.random((1000, 2)) * 1.8 - 0.9).astype(np.float32)

    def wav(path, tag, bits, payload, nch=2, sr=44100, extensible=False):
        block = nch * bits // 8
        if extensible:
            fmt = struct.pack("<HHIIHH", 0xFFFE, nch, sr, sr * block, block, bits) + struct.pack("<HHI", 22, bits, 3) + \
                struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        else:
            fmt = struct.pack("<HHIIHH", tag, nch, sr, sr * block, block, bits)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
            b"data" + struct.pack("<I", len(payload)) + payload
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", len(body)) + body)

    cases = {
        "pcm16": (1, 16, np.round(x * 32767).astype("<i2").tobytes(), 2 / 32768),
        "pcm32": (1, 32, np.round(x.astype(np.float64) * 2147483647).astype("<i4").tobytes(), 1e-6),
        "pcm8": (1, 8, (np.round(x * 127) + 128).astype(np.uint8).tobytes(), 2 / 127),
        "f32": (3, 32, x.astype("<f4").tobytes(), 0.0),
        "f64": (3, 64, x.astype("<f8").tobytes(), 1e-7),
    }
    v24 = np.round(x.astype(np.float64) * 8388607).astype(np.int32)
    b24 = np.stack([(v24 >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8).tobytes()
    cases["pcm24"] = (1, 24, b24, 1e-6)
    for name, (tag, bits, payload, tol) in cases.items():
        for ext in (False, True):
            p = str(tmp_path / f"{name}{int(ext)}.wav")
            wav(p, tag, bits, payload, extensible=ext)
            got, sr = load_audio(p)
            assert sr == 44100 and tuple(got.shape) == (2, 1000) and got.dtype.is_floating_point, name
            assert float(np.abs(got.numpy().T - x).max()) <= tol + 1e-7, (name, ext)
    p = str(tmp_path / "x.flac")
    with open(p, "wb") as f:
        f.write(b"fLaC" + b"\x00" * 64)
    try:
        import soundfile  # noqa: F401
    except ImportError:
        try:
            import torchaudio  # noqa: F401
        except ImportError:
            with pytest.raises(ValueError, match="only WAV"):
                load_audio(p)


def test_unsupported_architecture_config_is_refused():
    import pytest
    import conette_amd  # noqa: F401
    from conette_amd import CoNeTTEConfig
    from conette_amd.model import CoNeTTEModel
    for kw in ({"acti_name": "relu"}, {"proj_name": "lin512"}):
        with pytest.raises((ValueError, RuntimeError)) as e:
            CoNeTTEModel(CoNeTTEConfig(**kw), state_dict={})
        assert "Unsupported config" in str(e.value) or "GPU" in str(e.value) or "missing" in str(e.value).lower()


def test_certificate_mask_logic():
    """engine.uncertified_mask (the host half of precision "certified", round 6): tolerances per plane, the final choice, +inf for steps a
    clip no longer takes, NaN / non-finite scores read as "not certified", and the relaxed policy ignoring the pick-order plane."""
    from conette_amd.engine import CERT_DEFAULT_BASE, CERT_TOL, uncertified_mask
    inf, nan = float("inf"), float("nan")
    max_pred = 4
    m = torch.full((6, 2, max_pred + 1), inf)
    lp = torch.zeros(6)
    m[0, 0, :3] = torch.tensor([0.5, 0.2, 0.3])                       # clip 0: every margin wide
    m[0, 1, :3] = torch.tensor([0.4, 0.6, 0.2])
    m[0, 0, max_pred] = 0.1
    m[1] = m[0]; m[1, 0, 1] = 0.01                                   # clip 1: a membership margin below a
    m[2] = m[0]; m[2, 1, 2] = 0.01                                   # clip 2: only a pick-ORDER margin below a
    m[3] = m[0]; m[3, 0, max_pred] = 0.001                           # clip 3: the final choice below c
    m[4] = m[0]; m[4, 0, 0] = nan                                    # clip 4: NaN margin
    m[5] = m[0]; lp[5] = -inf                                        # clip 5: non-finite score
    tol = (0.05, 0.0, 0.004)
    assert uncertified_mask(m, lp, tol, order=True).tolist() == [False, True, True, True, True, True]
    assert uncertified_mask(m, lp, tol, order=False).tolist() == [False, True, False, True, True, True]
    # a tolerance that grows with the step: b * (i + 1)
    assert uncertified_mask(m[:1], lp[:1], (0.0, 0.11, 0.0), order=True).tolist() == [True]      # step 1 needs 0.22 > 0.2
    assert uncertified_mask(m[:1], lp[:1], (0.0, 0.06, 0.0), order=True).tolist() == [False]
    # zero tolerance certifies everything finite, an infinite one nothing that decided anything
    assert uncertified_mask(m[:4], lp[:4], (0.0, 0.0, 0.0)).tolist() == [False] * 4
    assert uncertified_mask(m[:4], lp[:4], (inf, 0.0, 0.0)).tolist() == [True] * 4
    # the shipped table: every base has both kinds of search, greedy tolerances below beam tolerances, and a default base
    assert CERT_DEFAULT_BASE in CERT_TOL
    from conette_amd.engine import cert_kind
    assert [cert_kind(k) for k in (1, 2, 3, 5, 6, 8, 10, 16)] == ["greedy", "beam", "beam", "beam", "wide", "wide", "wide", "wide"]
    for base, t in CERT_TOL.items():
        assert set(t) == {"greedy", "beam", "wide"} and t["greedy"][0] <= t["beam"][0] <= t["wide"][0], base
        assert t["greedy"][2] == 0.0 and 0.0 < t["beam"][2] <= t["wide"][2], base


def test_caption_sizes_of_merged_searches():
    """engine.caption_sizes recomputes [pred_size, best_maxlen] (beam.py:192-194,207-211,222-225) from full-width ids: on every committed
    fixture it must return the widths the reference's own outputs have."""
    from conette_amd.engine import caption_sizes
    from tests import golden_util as G
    for name in G.SCENARIOS:
        g = G.load(name)
        preds, mult = torch.from_numpy(g["preds"]), torch.from_numpy(g["mult_preds"])
        max_pred = 30
        full_b = torch.zeros((preds.shape[0], max_pred), dtype=torch.long); full_b[:, : preds.shape[1]] = preds
        full_m = torch.zeros((mult.shape[0], mult.shape[1], max_pred), dtype=torch.long); full_m[:, :, : mult.shape[2]] = mult
        ps, ml = caption_sizes(full_b, full_m, eos_id=2).tolist()
        kw = __import__("json").loads(str(g["kw"]))
        if kw.get("max_pred_size", 20) <= max_pred and mult.shape[2] < kw.get("max_pred_size", 20):
            assert (ps, ml) == (mult.shape[2], preds.shape[1]), name
        else:   # (a hypothesis that ran to max_pred_size without <eos>: the padded table cannot tell -- the merge keeps the searches' width)
            assert ml <= ps

"""The PreTrainedModel surface of CoNeTTEModel and ``conette-predict --model_path`` (VERDICT r05 missing 3-4 / next 8).

Reference: huggingface/model.py:126-183 (``state_dict()`` packs the non-tensor states -- the fitted tokenizer -- into the pickled
uint8 tensor ``_extra_state_``; ``from_pretrained`` / ``save_pretrained`` / ``.to()`` come from transformers' PreTrainedModel),
predict.py:123-178 (a training log directory: hydra/config.yaml + checkpoints/best.ckpt).

CPU: the error behaviour of the model_path loader.  GPU: save -> from_pretrained -> identical ids; load_state_dict; .to(); the CLI
on a model_path directory that carries the encoder.
"""
import os
import pickle

import numpy as np
import pytest
import torch

TAGS = {i: f"tag{i}" for i in range(527)}


def _write_model_path(root, sd_hf, cfg, with_encoder=True, target="conette.pl_modules.conette.CoNeTTEPLM", encoder_file=False):
    """A training log directory of the reference's layout from an HF-layout synthetic state dict."""
    import yaml
    from conette_amd import synth
    os.makedirs(os.path.join(root, "hydra"))
    os.makedirs(os.path.join(root, "checkpoints"))
    pl = {"_target_": target}
    for k in ("task_mode", "task_names", "gen_test_cands", "label_smoothing", "gen_val_cands", "mixup_alpha", "proj_name", "min_pred_size",
              "max_pred_size", "beam_size", "nhead", "d_model", "num_decoder_layers", "decoder_dropout_p", "dim_feedforward", "acti_name",
              "optim_name", "lr", "weight_decay", "betas", "eps", "use_custom_wd", "sched_name", "sched_n_steps", "sched_interval",
              "sched_freq", "verbose"):
        if k in cfg:
            v = cfg[k]
            pl[k] = list(v) if isinstance(v, tuple) else v
    with open(os.path.join(root, "hydra", "config.yaml"), "w") as f:
        yaml.safe_dump({"pl": pl, "seed": 1234}, f)
    plm, enc = {}, {}
    for k, v in sd_hf.items():
        if k.startswith("model."):
            plm[k[len("model."):]] = v
        elif k.startswith("preprocessor.encoder."):
            enc[k] = v
    plm["tokenizers.0._extra_state"] = synth.synth_tokenizer_state()
    if with_encoder and not encoder_file:
        plm.update(enc)
    torch.save({"state_dict": plm, "hyper_parameters": pl}, os.path.join(root, "checkpoints", "best.ckpt"))
    if encoder_file:
        torch.save({k[len("preprocessor."):]: v for k, v in enc.items()}, os.path.join(root, "checkpoints", "encoder.ckpt"))
    return root


@pytest.fixture(scope="module")
def small_sd():
    """(CPU tests) a tiny stand-in with the key layout only -- the loader never looks at shapes."""
    return {"model.decoder.classifier.weight": torch.zeros(8, 4), "model.projection.2.weight": torch.zeros(4, 4),
            "preprocessor.encoder.stages.0.0.dwconv.weight": torch.zeros(2, 1, 7, 7)}


def test_model_path_loader_error_behaviour(tmp_path, small_sd, synth_cfg):
    from conette_amd.predict import model_path_state_dict
    with pytest.raises(FileNotFoundError, match="Cannot find model_path directory"):
        model_path_state_dict(str(tmp_path / "nope"))
    d = tmp_path / "a"
    d.mkdir()
    with pytest.raises(FileNotFoundError, match="Cannot find config file"):
        model_path_state_dict(str(d))
    (d / "hydra").mkdir()
    (d / "hydra" / "config.yaml").write_text("pl: {_target_: conette.pl_modules.conette.CoNeTTEPLM}\n")
    with pytest.raises(FileNotFoundError, match="Cannot find checkpoint file"):
        model_path_state_dict(str(d))
    # no encoder weights anywhere: refused with the reason (the reference would caption with a random ConvNeXt)
    p = _write_model_path(str(tmp_path / "noenc"), small_sd, synth_cfg, with_encoder=False)
    with pytest.raises(ValueError, match="no audio-encoder weights"):
        model_path_state_dict(p)
    p = _write_model_path(str(tmp_path / "base"), small_sd, synth_cfg, target="conette.pl_modules.baseline.BaselinePLM")
    with pytest.raises(NotImplementedError, match="BaselinePLM"):
        model_path_state_dict(p)
    p = _write_model_path(str(tmp_path / "other"), small_sd, synth_cfg, target="something.Else")
    with pytest.raises(NotImplementedError, match="Unsupported pretrained model type"):
        model_path_state_dict(p)


@pytest.mark.parametrize("encoder_file", [False, True])
def test_model_path_loader_maps_keys(tmp_path, small_sd, synth_cfg, encoder_file):
    from conette_amd import synth
    from conette_amd.predict import model_path_state_dict
    p = _write_model_path(str(tmp_path / "ok"), small_sd, synth_cfg, encoder_file=encoder_file)
    config, sd = model_path_state_dict(p)
    assert set(k for k in sd if isinstance(sd[k], torch.Tensor)) == set(small_sd)
    assert sd["model.tokenizers.0._extra_state"] == synth.synth_tokenizer_state()
    assert config.beam_size == synth_cfg["beam_size"] and list(config.task_names) == list(synth_cfg["task_names"])
    assert config.task_mode == synth_cfg["task_mode"]


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    from conette_amd import synth
    return synth.write_pretrained_dir(str(tmp_path_factory.mktemp("conette_rt")))


def _wave(n=3, length=48000):
    from conette_amd import synth
    return torch.from_numpy(synth.synth_waveforms(n, length, 4242, lengths=[length, length - 9000, length - 20000][:n]))


@pytest.mark.gpu
@pytest.mark.parametrize("safe", [True, False])
def test_save_pretrained_round_trip_identical_ids(model_dir, tmp_path, safe):
    from conette_amd import CoNeTTEModel
    m = CoNeTTEModel.from_pretrained(model_dir, precision="exact", offline=True, audioset_idx_to_name=TAGS)
    x = _wave()
    a = m(x, sr=32000, task="audiocaps")
    sd = m.state_dict()
    # the reference's layout: tensors by reference key + the pickled non-tensor states (model.py:163-183)
    assert sd["_extra_state_"].dtype == torch.uint8 and all(v.is_contiguous() for v in sd.values())
    extra = pickle.loads(bytes(sd["_extra_state_"].tolist()))
    assert list(extra) == ["model.tokenizers.0._extra_state"] and extra["model.tokenizers.0._extra_state"]["tokenizer"]["itos"][4] == "w4"
    assert "preprocessor.encoder.stages.2.8.pwconv2.weight" in sd and "model.decoder.classifier.weight" in sd
    out = str(tmp_path / "saved")
    m.save_pretrained(out, safe_serialization=safe)
    assert os.path.isfile(os.path.join(out, "config.json"))
    assert os.path.isfile(os.path.join(out, "model.safetensors" if safe else "pytorch_model.bin"))
    m2 = CoNeTTEModel.from_pretrained(out, precision="exact", offline=True, audioset_idx_to_name=TAGS)
    b = m2(x, sr=32000, task="audiocaps")
    assert a["preds"].cpu().tolist() == b["preds"].cpu().tolist() and a["cands"] == b["cands"]
    assert a["mult_preds"].cpu().tolist() == b["mult_preds"].cpu().tolist()
    assert torch.equal(a["lprobs"].cpu(), b["lprobs"].cpu())          # the same weights, the same kernels: the same bits
    assert m2.tokenizer.get_vocab_size() == m.tokenizer.get_vocab_size()


@pytest.mark.gpu
def test_load_state_dict_and_to(model_dir):
    from conette_amd import CoNeTTEModel, synth
    m = CoNeTTEModel.from_pretrained(model_dir, precision="exact", offline=True, audioset_idx_to_name=TAGS)
    x = _wave()
    a = m(x, sr=32000)
    assert m.to("cuda:0") is m and m.to(torch.device("cuda")) is m and m.cuda() is m and m.to(torch.float16) is m
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.to("cpu")
    # another checkpoint of the same layout (another seed): other captions; loading the first one back restores them
    other = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(seed=3).items()}
    other["_extra_state_"] = torch.from_numpy(synth.extra_state_tensor())
    first = m.state_dict()
    m.load_state_dict(other)
    b = m(x, sr=32000)
    assert b["preds"].cpu().tolist() != a["preds"].cpu().tolist()
    m.load_state_dict(first)
    c = m(x, sr=32000)
    assert c["preds"].cpu().tolist() == a["preds"].cpu().tolist() and torch.equal(c["lprobs"].cpu(), a["lprobs"].cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("encoder_file", [False, True])
def test_predict_cli_model_path(model_dir, tmp_path, synth_cfg, encoder_file):
    import wave
    from conette_amd import synth
    from conette_amd.model import _read_state_dict
    from conette_amd.predict import main_predict
    sd = {k: v for k, v in _read_state_dict(model_dir).items() if k != "_extra_state_"}
    mp = _write_model_path(str(tmp_path / "run"), sd, synth_cfg, encoder_file=encoder_file)
    paths = []
    for i in range(2):
        wav = synth.synth_waveforms(1, 44000 + 6000 * i, 311 + i)[0]
        pcm = np.clip(np.round(wav * 32768.0), -32768, 32767).astype("<i2")
        p = str(tmp_path / f"c{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1), w.setsampwidth(2), w.setframerate(32000)
            w.writeframes(pcm.tobytes())
        paths.append(p)
    cache = tmp_path / "audioset_mapping"
    cache.mkdir()
    with open(cache / "class_labels_indices.csv", "w") as f:
        f.write("index,mid,display_name\n" + "".join(f"{i},/m/{i},tag{i}\n" for i in range(527)))
    os.environ["CONETTE_AUDIOSET_CACHE"] = str(cache)
    try:
        by_path = main_predict(["--audio", *paths, "--task", "clotho", "--model_path", mp, "--precision", "exact", "--verbose", "0"])
        by_name = main_predict(["--audio", *paths, "--task", "clotho", "--model_name", model_dir, "--precision", "exact", "--verbose", "0"])
    finally:
        os.environ.pop("CONETTE_AUDIOSET_CACHE", None)
    assert by_path == by_name and all(len(r["candidate"]) > 0 for r in by_path)

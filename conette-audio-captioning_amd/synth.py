"""Deterministic synthetic CoNeTTE checkpoint + waveforms (numpy only).

The published ``Labbeti/conette`` weights are unreachable offline, so every
parity test and the benchmark run on a seeded checkpoint that is regenerated bit-identically
on any machine from this recipe.  Key names / shapes are the reference's state-dict layout
(reference src/conette/huggingface/model.py:165-183).

Only ``Generator.random`` (uniform doubles from PCG64 raw words) is used: its bit stream is
frozen across numpy versions, unlike ``Generator.normal``.

Weights are scaled so that the network behaves like a trained one:
LayerScale 0.1..0.5 (the reference's 1e-6 init makes every block an identity,
convnext.py:38,50-54), peaked logits, an EOS bias that ends captions after a few steps, and
cross-attention strong enough that different audio gives different captions.
"""
from __future__ import annotations

import hashlib
import json
import math
import os
import pickle
from typing import Any, Dict, List, Optional

import numpy as np

TASK_NAMES = (
    "clotho",
    "audiocaps",
    "macs",
    "wavcaps_audioset_sl",
    "wavcaps_bbc_sound_effects",
    "wavcaps_freesound",
    "wavcaps_soundbible",
)
SPECIAL_TOKENS = ("<pad>", "<bos>", "<eos>", "<unk>")
N_WORDS_DEFAULT = 5620  # 4 specials + 5620 words + 7 task tokens = 5631
N_STOPWORDS = 147  # ids 4..150 are "stop-words": free to repeat

SA_QK, SA_V, CA_QK, CA_V = 0.08, 0.03, 0.12, 0.03
DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)
SAMPLE_RATE = 32000


def _rng(key: str, seed: int = 0) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def _uni(key: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> np.ndarray:
    """mean + uniform(-a, a) with a = std*sqrt(3)."""
    u = _rng(key, seed).random(size=shape) * 2.0 - 1.0
    return (mean + u * (std * math.sqrt(3.0))).astype(np.float32)


def _rangeu(key: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    u = _rng(key, seed).random(size=shape)
    return (lo + u * (hi - lo)).astype(np.float32)


def synth_words(n_words: int = N_WORDS_DEFAULT) -> List[str]:
    return [f"w{i}" for i in range(4, 4 + n_words)]


def synth_stopwords() -> List[str]:
    return [f"w{i}" for i in range(4, 4 + N_STOPWORDS)]


def synth_tokenizer_state(n_words: int = N_WORDS_DEFAULT, with_task_tokens: bool = True) -> Dict[str, Any]:
    """Text state in the schema of reference tokenization/aac_tokenizer.py:819-837."""
    itos = {i: t for i, t in enumerate(SPECIAL_TOKENS)}
    for i, w in enumerate(synth_words(n_words)):
        itos[4 + i] = w
    added = []
    if with_task_tokens:
        base = 4 + n_words
        for i, name in enumerate(TASK_NAMES):
            tok = f"<bos_{name}>"
            itos[base + i] = tok
            added.append(tok)
    stoi = {t: i for i, t in itos.items()}
    vocab = {t: 1 for t in stoi}
    for t in added:
        vocab[t] = 0
    return {
        "_target_": "conette.tokenization.aac_tokenizer.AACTokenizer",
        "_version_": "2.2.0",
        "_type_": "txt",
        "tokenizer": {
            "hparams": {"level": "word", "lowercase": True, "punctuation_mode": "remove", "normalize": True},
            "normalize": True,
            "added_special_tokens": added,
            "max_sentence_size": 20,
            "min_sentence_size": 3,
            "n_sentences_fit": 1000,
            "itos": itos,
            "stoi": stoi,
            "vocab": vocab,
        },
    }


def synth_config_dict(n_words: int = N_WORDS_DEFAULT) -> Dict[str, Any]:
    """Constructor kwargs of CoNeTTEConfig (reference huggingface/config.py:13-88)."""
    return dict(
        task_mode="ds_src",
        task_names=list(TASK_NAMES),
        gen_test_cands="generate",
        label_smoothing=0.2,
        gen_val_cands="generate",
        mixup_alpha=0.4,
        proj_name="lin768",
        min_pred_size=3,
        max_pred_size=20,
        beam_size=3,
        nhead=8,
        d_model=256,
        num_decoder_layers=6,
        decoder_dropout_p=0.2,
        dim_feedforward=2048,
        acti_name="gelu",
        verbose=0,
        tokenizer_state=synth_tokenizer_state(n_words, with_task_tokens=True),
    )


def _dft_kernels(n_fft: int = 1024):
    n = np.arange(n_fft, dtype=np.float64)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)
    kn = (np.outer(k, n).astype(np.int64) % n_fft).astype(np.float64)
    ang = -2.0 * np.pi * kn / n_fft
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    real = (np.cos(ang) * win[None, :]).astype(np.float32)[:, None, :]
    imag = (np.sin(ang) * win[None, :]).astype(np.float32)[:, None, :]
    return real, imag


def _mel_filterbank(sr=32000, n_fft=1024, n_mels=224, fmin=50.0, fmax=14000.0) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filters (librosa.filters.mel) -> (513, 224)."""
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def hz2mel(f):
        return f / f_sp if f < min_log_hz else min_log_mel + math.log(f / min_log_hz) / logstep

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        out = f_sp * m
        big = m >= min_log_mel
        out[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
        return out

    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels]))[:, None]
    return np.ascontiguousarray(w.T.astype(np.float32))


def _pos_embedding(maxlen: int = 5000, d: int = 256) -> np.ndarray:
    """reference nn/modules/positional_encoding.py:22-30, persisted buffer (maxlen, 1, d)."""
    den = np.exp(-np.arange(0, d, 2, dtype=np.float64) * math.log(10000.0) / d).astype(np.float32)
    pos = np.arange(maxlen, dtype=np.float32)[:, None]
    arg = (pos * den[None, :]).astype(np.float32).astype(np.float64)
    pe = np.zeros((maxlen, d), dtype=np.float32)
    pe[:, 0::2] = np.sin(arg).astype(np.float32)
    pe[:, 1::2] = np.cos(arg).astype(np.float32)
    return pe[:, None, :]


def synth_state_dict(n_words: int = N_WORDS_DEFAULT, seed: int = 0, eos_bias: float = 14.0,
                     d_model: int = 256, n_layers: int = 6, d_ff: int = 2048, recipe: str = "default") -> Dict[str, np.ndarray]:
    """All tensors of the reference state dict (except the pickled ``_extra_state_``).

    recipe = "peaked" (round 5): the same encoder, a decoder whose next-token distribution looks like a trained model's --
    see ``_peaked_decoder``."""
    sd: Dict[str, np.ndarray] = {}
    vocab = 4 + n_words + len(TASK_NAMES)
    E = "preprocessor.encoder."

    def U(key, shape, std=1.0, mean=0.0):
        sd[key] = _uni(key, shape, std, mean, seed)

    real, imag = _dft_kernels()
    sd[E + "spectrogram_extractor.stft.conv_real.weight"] = real
    sd[E + "spectrogram_extractor.stft.conv_imag.weight"] = imag
    sd[E + "logmel_extractor.melW"] = _mel_filterbank()
    U(E + "bn0.weight", (224,), 0.06, 1.0)
    U(E + "bn0.bias", (224,), 0.06)
    U(E + "bn0.running_mean", (224,), 3.0, -10.0)
    sd[E + "bn0.running_var"] = _rangeu(E + "bn0.running_var", (224,), 150.0, 260.0, seed)
    sd[E + "bn0.num_batches_tracked"] = np.asarray(1000, dtype=np.int64)

    U(E + "downsample_layers.0.0.weight", (96, 1, 4, 4), 0.25)
    U(E + "downsample_layers.0.0.bias", (96,), 0.06)
    U(E + "downsample_layers.0.1.weight", (96,), 0.06, 1.0)
    U(E + "downsample_layers.0.1.bias", (96,), 0.06)
    for i in range(3):
        c, c2 = DIMS[i], DIMS[i + 1]
        p = E + f"downsample_layers.{i + 1}."
        U(p + "0.weight", (c,), 0.06, 1.0)
        U(p + "0.bias", (c,), 0.06)
        U(p + "1.weight", (c2, c, 2, 2), 1.0 / math.sqrt(4 * c))
        U(p + "1.bias", (c2,), 0.06)
    for s in range(4):
        c = DIMS[s]
        for b in range(DEPTHS[s]):
            p = E + f"stages.{s}.{b}."
            sd[p + "scale_layer"] = _rangeu(p + "scale_layer", (c,), 0.1, 0.5, seed)
            U(p + "dwconv.weight", (c, 1, 7, 7), 0.1)
            U(p + "dwconv.bias", (c,), 0.03)
            U(p + "norm.weight", (c,), 0.06, 1.0)
            U(p + "norm.bias", (c,), 0.06)
            U(p + "pwconv1.weight", (4 * c, c), 1.0 / math.sqrt(c))
            U(p + "pwconv1.bias", (4 * c,), 0.12)
            U(p + "pwconv2.weight", (c, 4 * c), 1.5 / math.sqrt(4 * c))
            U(p + "pwconv2.bias", (c,), 0.06)
    U(E + "norm.weight", (768,), 0.06, 1.0)
    U(E + "norm.bias", (768,), 0.06)
    U(E + "head_audioset.weight", (527, 768), 2.0 / math.sqrt(768))
    U(E + "head_audioset.bias", (527,), 0.6, -1.0)

    M = "model."
    base = 4 + n_words
    sd[M + "task_id_to_token_id"] = np.arange(base, base + len(TASK_NAMES), dtype=np.int64)
    frm = np.ones((vocab,), dtype=np.bool_)
    frm[4 : 4 + N_STOPWORDS] = False
    sd[M + "forbid_rep_mask"] = frm
    U(M + "projection.2.weight", (d_model, 768), 2.0 / math.sqrt(768))
    U(M + "projection.2.bias", (d_model,), 0.06)
    D = M + "decoder."
    for l in range(n_layers):
        p = D + f"layers.{l}."
        # q/k rows strong (peaked, input-dependent attention), v rows weak so that the attention
        # outputs do not swamp the token stream under the post-norm residuals
        for name, qk_std, v_std in (("self_attn", SA_QK, SA_V), ("multihead_attn", CA_QK, CA_V)):
            blocks = [_uni(p + f"{name}.in_proj_weight.{part}", (d_model, d_model), std, 0.0, seed)
                      for part, std in (("q", qk_std), ("k", qk_std), ("v", v_std))]
            sd[p + f"{name}.in_proj_weight"] = np.concatenate(blocks, axis=0)
            U(p + f"{name}.in_proj_bias", (3 * d_model,), 0.03)
            U(p + f"{name}.out_proj.weight", (d_model, d_model), 1.0 / math.sqrt(d_model))
            U(p + f"{name}.out_proj.bias", (d_model,), 0.03)
        U(p + "linear1.weight", (d_ff, d_model), 1.0 / math.sqrt(d_model))
        U(p + "linear1.bias", (d_ff,), 0.06)
        U(p + "linear2.weight", (d_model, d_ff), 1.5 / math.sqrt(d_ff))
        U(p + "linear2.bias", (d_model,), 0.03)
        for n in ("norm1", "norm2", "norm3"):
            U(p + n + ".weight", (d_model,), 0.06, 1.0)
            U(p + n + ".bias", (d_model,), 0.06)
    U(D + "emb_layer.weight", (vocab, d_model), 1.0 / math.sqrt(d_model))
    sd[D + "emb_layer.weight"][0] = 0.0  # padding_idx row (aac_tfmer.py:40-45)
    sd[D + "pos_encoding.pos_embedding"] = _pos_embedding(5000, d_model)
    U(D + "classifier.weight", (vocab, d_model), 0.4)
    U(D + "classifier.bias", (vocab,), 0.3)
    sd[D + "classifier.bias"][2] = np.float32(eos_bias)
    # specials other than EOS and the task tokens are never wanted as output words
    for t in (0, 1, 3):
        sd[D + "classifier.bias"][t] = np.float32(-30.0)
    sd[D + "classifier.bias"][base:] = np.float32(-30.0)
    if recipe == "peaked":
        _peaked_decoder(sd, n_words, seed, d_model, n_layers)
    elif recipe != "default":
        raise ValueError(f"unknown recipe {recipe!r}")
    return sd


def synth_baseline_state_dict(n_words: int = N_WORDS_DEFAULT, seed: int = 0) -> Dict[str, Any]:
    """A BaselinePLM-layout state dict (reference pl_modules/baseline.py:84-140: no audio encoder, no task tokens): the decoder
    and projection of ``synth_state_dict`` at the vocabulary of a tokenizer WITHOUT task tokens (4 + n_words ids) under the keys
    a Lightning checkpoint of that module holds -- ``projection.2.*``, ``decoder.*``, ``forbid_rep_mask`` (content_words: the
    synthetic stop-words may repeat) and the tokenizer's ``tokenizers.0._extra_state`` dict."""
    full = synth_state_dict(n_words, seed)
    vocab = 4 + n_words
    out: Dict[str, Any] = {}
    for k, v in full.items():
        if k.startswith("model.projection.") or k.startswith("model.decoder."):
            kk = k[len("model."):]
            if kk in ("decoder.emb_layer.weight", "decoder.classifier.weight", "decoder.classifier.bias"):
                v = v[:vocab].copy()          # the task tokens' rows (appended behind the words) do not exist
            out[kk] = v
    frm = np.ones((vocab,), dtype=np.bool_)
    frm[4: 4 + N_STOPWORDS] = False
    out["forbid_rep_mask"] = frm
    out["tokenizers.0._extra_state"] = synth_tokenizer_state(n_words, with_task_tokens=False)
    return out


# ---- recipe "peaked" -------------------------------------------------------------------------------------------------
# The default recipe's logits are Gaussian over the vocabulary: the gap between neighbouring top candidates is ~ sigma / 4
# whatever the scale, i.e. the same few percent of a logit that 16-bit operand rounding moves -- scaling the classifier scales
# both (VERDICT r04 asked for "classifier / embedding scale up": that alone cannot help).  A trained captioner is peaked in
# another way: FEW candidates far above a noise floor.  This recipe builds that by construction:
#   * the token stream survives the six post-norm layers (attention values and the FFN's second matrix are small, embeddings
#     twice as long as the positional encoding), so the final state is ~0.9 aligned with the current token's embedding;
#   * every token u has a SUCCESSOR GROUP g(u) of two words; the classifier row of word v is
#         s x (sum of the embeddings of the tokens whose successor group holds v)  +  s2 x (a random direction R_v),
#     so the two successors of the current token sit ~20 above the ~N(0, s2^2) floor of the other 5 629 logits and are told
#     apart by R_v . x -- which depends on position and audio;
#   * a share of the tokens is "terminal": their embedding carries a component along a fixed direction q, and the <eos> row
#     is s_e x q -- captions end after a terminal token (lengths 5-13 on the fixtures' clips).
# Measured on the reference with the 8 clips of b8_10s (oracle/gen_golden.py): greedy calls with top-2 margin > 0.25: 93 %;
# beam 3 calls with top-(k+1) margin > 0.25: 68 % (hypotheses of a beam search ARE near each other; default recipe: 14-17 %).
PEAKED = dict(s=1.8, s2=7.0, s_e=4.5, tau=2.4, term_frac=0.3, gsize=2, emb_scale=2.0, ffn_scale=0.1, ca_v=0.025, sa_v_scale=0.2)


def _peaked_decoder(sd: Dict[str, np.ndarray], n_words: int, seed: int, d: int, n_layers: int) -> None:
    P = PEAKED
    vocab = 4 + n_words + len(TASK_NAMES)
    D = "model.decoder."
    q = _uni("peaked:q", (d,), 1.0 / math.sqrt(d), 0.0, seed).astype(np.float64)
    q = q / math.sqrt(float((q * q).sum()))
    E0 = sd[D + "emb_layer.weight"].astype(np.float64) * P["emb_scale"]
    term = _rng("peaked:term", seed).random(vocab) < P["term_frac"]
    term[:4] = False
    E = E0 + P["tau"] * term[:, None] * q[None, :]
    E[0] = 0.0                                              # padding_idx row
    sd[D + "emb_layer.weight"] = E.astype(np.float32)
    gsize = int(P["gsize"])
    n_groups = n_words // gsize
    gam = (_rng("peaked:gamma", seed).random(vocab) * n_groups).astype(np.int64)   # successor group of every token
    S = np.zeros((n_groups, d))
    np.add.at(S, gam[1:], E0[1:])                           # per group: sum of its predecessors' (plain) embeddings
    R = _uni("peaked:R", (vocab, d), 1.0 / math.sqrt(d), 0.0, seed).astype(np.float64)
    Wc = np.zeros((vocab, d))
    words = np.arange(4, 4 + n_groups * gsize)
    Wc[words] = P["s"] * S[(words - 4) // gsize] + P["s2"] * R[words]
    Wc[2] = P["s_e"] * q
    sd[D + "classifier.weight"] = Wc.astype(np.float32)
    b = sd[D + "classifier.bias"].copy()
    b[2] = np.float32(0.0)
    b[4 + n_groups * gsize: 4 + n_words] = np.float32(-30.0)
    sd[D + "classifier.bias"] = b
    for l in range(n_layers):
        p = D + f"layers.{l}."
        sd[p + "linear2.weight"] = (sd[p + "linear2.weight"] * np.float32(P["ffn_scale"])).astype(np.float32)
        w = sd[p + "multihead_attn.in_proj_weight"].copy()
        w[2 * d:] *= np.float32(P["ca_v"] / CA_V)
        sd[p + "multihead_attn.in_proj_weight"] = w
        w = sd[p + "self_attn.in_proj_weight"].copy()
        w[2 * d:] *= np.float32(P["sa_v_scale"])
        sd[p + "self_attn.in_proj_weight"] = w


def synth_waveforms(batch: int, n_samples: int = 10 * SAMPLE_RATE, seed0: int = 1234,
                    lengths: Optional[List[int]] = None) -> np.ndarray:
    """(batch, n_samples) float32 mono @ 32 kHz: broadband noise (std 0.1) + 3 gated sinusoids
    (SURVEY.md section 8d).  Clip i uses generator seed ``seed0 + i``; if ``lengths`` is given,
    samples past ``lengths[i]`` are zero (the reference zero-pads to the batch max, pad.py:11-17)."""
    out = np.zeros((batch, n_samples), dtype=np.float32)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    for i in range(batch):
        g = np.random.Generator(np.random.PCG64(seed0 + i))
        n_i = n_samples if lengths is None else int(lengths[i])
        x = (g.random(n_samples) * 2.0 - 1.0) * (0.1 * math.sqrt(3.0))
        par = g.random((3, 4))
        dur = n_i / SAMPLE_RATE
        for s in range(3):
            f = 100.0 + par[s, 0] * 7900.0
            a = 0.1 + 0.3 * par[s, 1]
            on = par[s, 2] * 0.6 * dur
            off = on + (0.2 + 0.8 * par[s, 3]) * (dur - on)
            gate = (t >= on) & (t < off)
            x = x + gate * (a * np.sin(2.0 * np.pi * f * t))
        x[n_i:] = 0.0
        out[i] = x.astype(np.float32)
    return out


def extra_state_tensor(n_words: int = N_WORDS_DEFAULT) -> np.ndarray:
    """The pickled uint8 ``_extra_state_`` entry (reference huggingface/model.py:165-183)."""
    payload = {"model.tokenizers.0._extra_state": synth_tokenizer_state(n_words, True)}
    return np.frombuffer(pickle.dumps(payload), dtype=np.uint8).copy()


def write_pretrained_dir(path: str, n_words: int = N_WORDS_DEFAULT, seed: int = 0) -> str:
    """Write an HF-layout directory (config.json + pytorch_model.bin) from the recipe."""
    import torch

    os.makedirs(path, exist_ok=True)
    cfg = synth_config_dict(n_words)
    cfg["model_type"] = "conette"
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth_state_dict(n_words, seed).items()}
    sd["_extra_state_"] = torch.from_numpy(extra_state_tensor(n_words))
    torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    return path

"""ORACLE -- test infrastructure only.  NOT the product, never shipped, never the thing measured.

CPU (stock PyTorch, fp32) restatement of the reference's inference hot path
``waveform -> log-mel -> ConvNeXt-tiny -> projection -> Transformer decoder under beam search``
(SURVEY.md section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file, and only as the checker / reported CPU baseline.

Pinning: the reference has NO golden vector for this path.  The
oracle is pinned against outputs of the reference itself: ``oracle/gen_golden.py`` imports the
reference's own files from /root/reference (third-party leaves stood in by
``oracle/thirdparty.py``), runs them on the seeded synthetic checkpoint and commits the
results under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against
those fixtures.  The third-party leaves themselves (torchlibrosa / torchaudio / torchoutil)
are "parity unpinned" (see oracle/thirdparty.py).

Every function cites the reference lines it follows (paths relative to
/root/reference/src/conette/).
"""
from __future__ import annotations

import math
import re
import struct
import wave as _wave
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor
from torch.nn import functional as F

from . import thirdparty as tp

DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)
TARGET_SR = 32000
SYNTH_STOPWORDS = [f"w{i}" for i in range(4, 151)]

Weights = Dict[str, Tensor]


def to_torch(sd_np: Dict[str, np.ndarray]) -> Weights:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------------------
# a1  input normalisation -- huggingface/preprocessor.py:82-154, nn/functional/pad.py:11-17
# ----------------------------------------------------------------------------------------
def load_wav(path: str) -> Tuple[Tensor, int]:
    """PCM WAV -> ((C, L) float32 in [-1, 1), sr); stands in for torchaudio.load (preprocessor.py:79-80)."""
    with _wave.open(path, "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    data = data.reshape(-1, nch).T
    return torch.from_numpy(np.ascontiguousarray(data)), sr


def load_resample(x, sr=None, x_shapes=None) -> Tuple[Tensor, Tensor]:
    """preprocessor.py:82-154: accept path(s) / Tensor (T,), (C,T), (B,C,T) / list of (C,T);
    resample to 32 kHz if needed; channel mean; zero-pad + stack; lengths (B, 1)."""
    if isinstance(x, str) or (not isinstance(x, Tensor) and all(isinstance(v, str) for v in x)):
        if isinstance(x, str):
            x = [x]
        loaded = [load_wav(p) for p in x]
        x = [a for a, _ in loaded]
        sr = [s for _, s in loaded]
    else:
        if isinstance(x, Tensor):
            if x.ndim == 1:
                x = x[None, None]
            elif x.ndim == 2:
                x = x[None]
            elif x.ndim != 3:
                raise ValueError(f"Invalid argument shape {x.shape=}.")
        else:
            x = list(x)
        if isinstance(sr, int):
            sr = [sr]
        elif sr is None:
            sr = [TARGET_SR]
        else:
            sr = list(sr)
    if len(sr) == 1 and len(x) != len(sr):
        sr = sr * len(x)
    assert len(x) == len(sr) and len(x) > 0
    if any(s != TARGET_SR for s in sr):
        if x_shapes is not None:
            raise ValueError(f"Invalid argument {x_shapes=}.")
        if tp.all_eq(sr) and isinstance(x, Tensor):
            x = tp.resample(x, sr[0], TARGET_SR)
        else:
            x = [tp.resample(xi, si, TARGET_SR) for xi, si in zip(x, sr)]
    if isinstance(x, Tensor):
        x = x.mean(dim=1)
    else:
        x = [xi.mean(dim=0) for xi in x]
    if x_shapes is None:
        x_shapes = [list(xi.shape) for xi in x]
    x_shapes = torch.as_tensor(x_shapes)
    if not isinstance(x, Tensor):
        max_len = max(xi.shape[-1] for xi in x)
        x = torch.stack([tp.pad_dim(xi, max_len, dim=-1) for xi in x], dim=0)
    return x, x_shapes


# ----------------------------------------------------------------------------------------
# a2  log-mel frontend + bn0 -- nn/encoders/convnext.py:270-292 (+ torchlibrosa restated)
# ----------------------------------------------------------------------------------------
def logmel_bn0(w: Weights, wave: Tensor) -> Tensor:
    """(B, L) -> (B, 1, F, 224): reflect-pad 512, DFT-as-conv1d (hop 320), power, mel matmul,
    10*log10(clamp 1e-10), eval BatchNorm2d over the mel axis (eps 1e-5)."""
    p = "preprocessor.encoder."
    x = wave[:, None, :]
    x = F.pad(x, (512, 512), mode="reflect")
    real = F.conv1d(x, w[p + "spectrogram_extractor.stft.conv_real.weight"], stride=320)
    imag = F.conv1d(x, w[p + "spectrogram_extractor.stft.conv_imag.weight"], stride=320)
    real = real[:, None].transpose(2, 3)
    imag = imag[:, None].transpose(2, 3)
    spec = real ** 2 + imag ** 2  # (B, 1, F, 513)
    mel = torch.matmul(spec, w[p + "logmel_extractor.melW"])
    logmel = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    logmel = logmel - 10.0 * math.log10(max(1e-10, 1.0))
    x = logmel.transpose(1, 3)
    x = F.batch_norm(x, w[p + "bn0.running_mean"], w[p + "bn0.running_var"], w[p + "bn0.weight"],
                     w[p + "bn0.bias"], training=False, eps=1e-5)
    return x.transpose(1, 3)


def _ln_cf(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    """nn/modules/norm.py:35-40 (channels_first LayerNorm, biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


def convnext_block(w: Weights, prefix: str, x: Tensor) -> Tensor:
    """nn/encoders/convnext.py:61-74."""
    c = x.shape[1]
    y = F.conv2d(x, w[prefix + "dwconv.weight"], w[prefix + "dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), w[prefix + "norm.weight"], w[prefix + "norm.bias"], 1e-6)
    y = F.linear(y, w[prefix + "pwconv1.weight"], w[prefix + "pwconv1.bias"])
    y = F.gelu(y)
    y = F.linear(y, w[prefix + "pwconv2.weight"], w[prefix + "pwconv2.bias"])
    y = w[prefix + "scale_layer"] * y
    y = y.permute(0, 3, 1, 2)
    return x + y


def convnext_encode(w: Weights, wave: Tensor, x_shapes: Tensor, taps: Optional[dict] = None) -> Dict[str, Tensor]:
    """nn/encoders/convnext.py:264-336 with the inference flags of huggingface/preprocessor.py:23-33."""
    p = "preprocessor.encoder."
    x = logmel_bn0(w, wave)
    if taps is not None:
        taps["logmel"] = x
    for i in range(4):
        d = p + f"downsample_layers.{i}."
        if i == 0:
            x = F.conv2d(x, w[d + "0.weight"], w[d + "0.bias"], stride=(4, 4), padding=(4, 0))
            x = _ln_cf(x, w[d + "1.weight"], w[d + "1.bias"])
            if taps is not None:
                taps["stem"] = x
        else:
            x = _ln_cf(x, w[d + "0.weight"], w[d + "0.bias"])
            x = F.conv2d(x, w[d + "1.weight"], w[d + "1.bias"], stride=2)
            if taps is not None:
                taps[f"down{i}"] = x
        for b in range(DEPTHS[i]):
            x = convnext_block(w, p + f"stages.{i}.{b}.", x)
            if taps is not None and b == 0:
                taps[f"stage{i}_block0"] = x
        if taps is not None:
            taps[f"stage{i}"] = x
    x = torch.mean(x, dim=3)
    frame_embs = x
    input_lens = x_shapes[:, -1]
    reduction_factor = wave.shape[-1] // frame_embs.shape[-1]
    frame_embs_lens = input_lens.div(reduction_factor).round().int()
    (x1, _) = torch.max(x, dim=2)
    x2 = torch.mean(x, dim=2)
    x = x1 + x2
    x = F.layer_norm(x, (768,), w[p + "norm.weight"], w[p + "norm.bias"], 1e-6)
    x = F.linear(x, w[p + "head_audioset.weight"], w[p + "head_audioset.bias"])
    return {"frame_embs": frame_embs, "frame_embs_lens": frame_embs_lens, "clipwise_output": torch.sigmoid(x)}


def preprocessor_forward(w: Weights, x, sr=None, x_shapes=None, taps=None) -> Dict[str, Tensor]:
    """huggingface/preprocessor.py:50-77."""
    wave, shapes = load_resample(x, sr, x_shapes)
    outs = convnext_encode(w, wave, shapes, taps)
    frame_embs = outs["frame_embs"].transpose(1, 2)
    audio_shape = torch.as_tensor([[768, int(n)] for n in outs["frame_embs_lens"]])
    return {"audio": frame_embs, "audio_shape": audio_shape, "clip_probs": outs["clipwise_output"]}


# ----------------------------------------------------------------------------------------
# a9  encode_audio -- pl_modules/conette.py:452-467, pl_modules/common.py:59-78, encoders/ident.py
# ----------------------------------------------------------------------------------------
def encode_audio(w: Weights, audio: Tensor, audio_shape: Tensor) -> Tuple[Tensor, Tensor]:
    """(B, T, 768) -> memory (B, 256, T) [Linear + ReLU, transposed], pad mask (B, T) True = padded."""
    if audio.ndim == 4:
        audio = audio.squeeze(1)
    lens = audio_shape[:, 1]
    y = F.relu(F.linear(audio, w["model.projection.2.weight"], w["model.projection.2.bias"]))
    y = y.transpose(1, 2)
    max_len = max(int(lens.max()), y.shape[-1])
    mask = tp.lengths_to_pad_mask(lens, max_len)
    return y, mask


# ----------------------------------------------------------------------------------------
# a14  decoder -- nn/decoders/aac_tfmer.py:71-118 + torch post-norm TransformerDecoderLayer
# ----------------------------------------------------------------------------------------
def _mha(x_q: Tensor, x_kv: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor,
         nhead: int, attn_mask: Optional[Tensor], key_padding_mask: Optional[Tensor]) -> Tensor:
    """torch.nn.MultiheadAttention forward (seq-first): q scaled by 1/sqrt(dh) before q.k^T,
    additive float mask (-inf), softmax, .v, out-proj."""
    tq, r, e = x_q.shape
    tk = x_kv.shape[0]
    dh = e // nhead
    q = F.linear(x_q, w_in[:e], b_in[:e])
    k = F.linear(x_kv, w_in[e : 2 * e], b_in[e : 2 * e])
    v = F.linear(x_kv, w_in[2 * e :], b_in[2 * e :])
    q = q.reshape(tq, r * nhead, dh).transpose(0, 1)
    k = k.reshape(tk, r * nhead, dh).transpose(0, 1)
    v = v.reshape(tk, r * nhead, dh).transpose(0, 1)
    q = q * math.sqrt(1.0 / float(dh))
    scores = torch.bmm(q, k.transpose(1, 2))  # (r*h, tq, tk)
    if attn_mask is not None:
        scores = scores + attn_mask[None]
    if key_padding_mask is not None:
        kpm = torch.zeros(key_padding_mask.shape, dtype=scores.dtype).masked_fill(key_padding_mask, -math.inf)
        scores = (scores.view(r, nhead, tq, tk) + kpm[:, None, None, :]).view(r * nhead, tq, tk)
    attn = torch.softmax(scores, dim=-1)
    out = torch.bmm(attn, v)  # (r*h, tq, dh)
    out = out.transpose(0, 1).reshape(tq, r, e)
    return F.linear(out, w_out, b_out)


def decoder_forward(w: Weights, memory: Tensor, mem_pad_mask: Optional[Tensor], caps_in: Tensor,
                    nhead: int = 8, n_layers: int = 6, caps_pad_mask: Optional[Tensor] = None) -> Tensor:
    """memory (T, R, d), mask (R, T) bool, caps_in (t, R) ids [, caps_pad_mask (R, t) bool] -> logits (t, R, V)."""
    D = "model.decoder."
    d = w[D + "emb_layer.weight"].shape[1]
    t = caps_in.shape[0]
    x = F.embedding(caps_in, w[D + "emb_layer.weight"]) * math.sqrt(d)
    x = x + w[D + "pos_encoding.pos_embedding"][:t]
    sq_mask = tp.generate_square_subsequent_mask(t)
    for l in range(n_layers):
        p = D + f"layers.{l}."
        sa = _mha(x, x, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"],
                  w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"], nhead, sq_mask, caps_pad_mask)
        x = F.layer_norm(x + sa, (d,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
        ca = _mha(x, memory, w[p + "multihead_attn.in_proj_weight"], w[p + "multihead_attn.in_proj_bias"],
                  w[p + "multihead_attn.out_proj.weight"], w[p + "multihead_attn.out_proj.bias"], nhead, None,
                  mem_pad_mask)
        x = F.layer_norm(x + ca, (d,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
        ff = F.linear(F.gelu(F.linear(x, w[p + "linear1.weight"], w[p + "linear1.bias"])),
                      w[p + "linear2.weight"], w[p + "linear2.bias"])
        x = F.layer_norm(x + ff, (d,), w[p + "norm3.weight"], w[p + "norm3.bias"], 1e-5)
    return F.linear(x, w[D + "classifier.weight"], w[D + "classifier.bias"])


# ----------------------------------------------------------------------------------------
# 8(f)3  teacher forcing -- nn/decoding/forcing.py:12-71 via pl_modules/conette.py:392-417
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def teacher_forcing(w: Weights, audio: Tensor, audio_shape: Tensor, caps_in: Tensor, *, pad_id: int = 0,
                    bos_id: int = 1, nhead: int = 8, n_layers: int = 6, require_task_token: bool = True) -> Tensor:
    """frame embeddings (B, T, 768) + input captions (B, t) -> logits (B, V, t).

    encode_audio (conette.py:452-470: projection + frame pad mask), then ONE causal decoder pass over caps_in with
    caps_in_pad_mask = tensor_to_pad_mask(caps_in, pad_value=pad_id) (forcing.py:44-49) and the square subsequent
    mask (:51-55); the result is permuted (caps, B, V) -> (B, V, caps) (:68-70).  conette.py:399-404 rejects captions
    whose first token is still <bos> (the task token must have replaced it); BaselinePLM (baseline.py:346-361) has no task
    tokens and no such check: ``require_task_token=False``."""
    if require_task_token and bool(caps_in[:, 0].eq(bos_id).any()):
        raise ValueError("BOS was not replaced in input captions for decode_method='forcing'.")
    memory_bdt, mask = encode_audio(w, audio, audio_shape)
    caps_pad_mask = caps_in.eq(pad_id)
    logits = decoder_forward(w, memory_bdt.permute(2, 0, 1), mask, caps_in.permute(1, 0), nhead, n_layers,
                             caps_pad_mask=caps_pad_mask)
    return logits.permute(1, 2, 0)


# ----------------------------------------------------------------------------------------
# a15 / 8(f)4  greedy search with full-logit output -- nn/decoding/greedy.py:17-131
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_search(w: Weights, memory_bdt: Tensor, mem_pad_mask: Tensor, bos_id: int, *, pad_id: int = 0, eos_id: int = 2,
                  vocab_size: int, min_pred_size: int = 0, max_pred_size: int = 20,
                  forbid_rep_mask: Optional[Tensor] = None, nhead: int = 8, n_layers: int = 6) -> Tensor:
    """memory (B, d, T), pad mask (B, T) -> logits (B, V, pred_size): per step the last-position logits of every
    unfinished clip (whole prefix re-decoded, :82-93) with the EOS floor (:96-97) and the forbid-repeat mask (:99-105)
    applied in place; arg-max continues the prefix (:107-109); a clip leaves the batch after <eos> or at the last
    step (:111-123); untouched entries keep the initial fill: -inf, row pad_id = 0 (:64-69)."""
    bsize = memory_bdt.shape[0]
    memory = memory_bdt.permute(2, 0, 1)
    batch_idxs = torch.arange(bsize)
    preds = torch.full((bsize, max_pred_size + 1), pad_id, dtype=torch.long)
    preds[:, 0] = bos_id
    out = torch.full((bsize, vocab_size, max_pred_size), -math.inf, dtype=memory_bdt.dtype)
    out[:, pad_id, :] = 0
    use_forbid = forbid_rep_mask is not None and bool(forbid_rep_mask.any())
    pred_size = max_pred_size
    mask = mem_pad_mask
    for i in range(max_pred_size):
        logits_i = decoder_forward(w, memory.contiguous(), mask, preds[:, : i + 1].transpose(0, 1), nhead, n_layers)[-1]
        if i < min_pred_size:
            logits_i[:, eos_id] = -math.inf
        if use_forbid:
            seen = torch.zeros((preds.shape[0], vocab_size), dtype=torch.bool)
            seen.scatter_(1, preds[:, : i + 1], True)
            logits_i[seen & forbid_rep_mask[None]] = -math.inf
        nxt = logits_i.argmax(dim=-1)
        preds[:, i + 1] = nxt
        unfinished = (nxt != eos_id) if i < max_pred_size - 1 else torch.zeros_like(nxt, dtype=torch.bool)
        out[batch_idxs, :, i] = logits_i
        preds, batch_idxs, memory, mask = preds[unfinished], batch_idxs[unfinished], memory[:, unfinished], mask[unfinished]
        if preds.nelement() <= 0:
            pred_size = i + 1
            break
    return out[:, :, :pred_size].contiguous() if pred_size < max_pred_size else out


# ----------------------------------------------------------------------------------------
# a12/a13  beam search -- nn/decoding/beam.py:22-269
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def generate(w: Weights, memory_bdt: Tensor, mem_pad_mask: Tensor, bos_ids: Tensor, *, pad_id: int = 0,
             eos_id: int = 2, vocab_size: int, beam_size: int = 3, min_pred_size: int = 3,
             max_pred_size: int = 20, forbid_rep_mask: Optional[Tensor] = None, nhead: int = 8,
             n_layers: int = 6, trace: Optional[list] = None):
    """Per-batch beam search with the reference's exact bookkeeping:

    * every step re-decodes the whole prefix of every active row (no KV cache, beam.py:113-123);
    * EOS is masked while step < min_pred_size (:129-130);
    * per clip j with k_j unfinished rows: forbid-repeat mask (:146-156), log-softmax, running
      sums, flat top-k_j over (k_j * V) -- step 0 uses the first row only (:230-269);
    * finished rows (EOS, or last step) are written to output slot ``beam_idx + clip*beam`` with
      score sum/(step+1) and dropped; the remaining rows continue with smaller k (:164-203);
    * best beam = first max of the averaged log-prob; trimming as :205-227.
    """
    bsize = memory_bdt.shape[0]
    mem = memory_bdt.repeat_interleave(beam_size, dim=0).permute(2, 0, 1).contiguous()  # (T, R, d)
    mask = mem_pad_mask.repeat_interleave(beam_size, dim=0)
    preds = torch.full((bsize * beam_size, max_pred_size + 1), pad_id, dtype=torch.long)
    preds[:, 0] = bos_ids.repeat_interleave(beam_size)
    batch_idxs = torch.arange(bsize).repeat_interleave(beam_size)
    beam_idxs = torch.arange(beam_size).repeat(bsize)
    sum_lprobs = torch.zeros(bsize * beam_size)
    out_preds = torch.full((bsize * beam_size, max_pred_size), pad_id, dtype=torch.long)
    out_done = torch.zeros(bsize * beam_size, dtype=torch.bool)
    out_avg = torch.zeros(bsize * beam_size)
    if forbid_rep_mask is None:
        forbid_rep_mask = torch.zeros(vocab_size, dtype=torch.bool)
    use_forbid = bool(forbid_rep_mask.any())
    pred_size = max_pred_size

    for i in range(max_pred_size):
        logits = decoder_forward(w, mem, mask, preds[:, : i + 1].t().contiguous(), nhead, n_layers)[-1]
        if i < min_pred_size:
            logits[:, eos_id] = -math.inf
        finished = torch.zeros(preds.shape[0], dtype=torch.bool)
        step_trace = []
        for j in torch.unique_consecutive(batch_idxs).tolist():
            rows = torch.nonzero(batch_idxs == j).flatten()
            lg = logits[rows]
            if use_forbid:
                hot = tp.indices_to_multihot(preds[rows, : i + 1], vocab_size)
                lg = lg.masked_fill(hot & forbid_rep_mask[None], -math.inf)
            k = rows.numel()
            if i == 0:
                cand = torch.log_softmax(lg[0:1], dim=1)
            else:
                cand = sum_lprobs[rows][:, None] + torch.log_softmax(lg, dim=1)
            vals, flat = torch.topk(cand.reshape(-1), k)
            parent = torch.div(flat, vocab_size, rounding_mode="trunc")
            token = flat % vocab_size
            sum_lprobs[rows] = vals
            preds[rows, : i + 1] = preds[rows][parent, : i + 1]
            preds[rows, i + 1] = token
            finished[rows] = (token == eos_id) if i < max_pred_size - 1 else True
            if trace is not None:
                top = torch.topk(cand.reshape(-1), min(k + 1, cand.numel())).values
                step_trace.append(dict(clip=j, parent=parent.tolist(), token=token.tolist(), sum_lprob=vals.tolist(),
                                       margin=float(top[k - 1] - top[k]) if top.numel() > k else float("inf")))
        if trace is not None:
            trace.append(step_trace)
        if finished.any():
            slots = beam_idxs[finished] + batch_idxs[finished] * beam_size
            out_preds[slots, : i + 1] = preds[finished, 1 : i + 2]
            out_done[slots] = True
            out_avg[slots] = sum_lprobs[finished] / (i + 1)
            if bool(out_done.all()):
                pred_size = i + 1
                break
        keep = ~finished
        mem, mask, preds = mem[:, keep], mask[keep], preds[keep]
        batch_idxs, beam_idxs, sum_lprobs = batch_idxs[keep], beam_idxs[keep], sum_lprobs[keep]

    out_preds = out_preds.reshape(bsize, beam_size, max_pred_size)[:, :, :pred_size].contiguous()
    out_avg = out_avg.reshape(bsize, beam_size)
    best_avg, best_beam = out_avg.max(dim=1)
    best_preds = out_preds[torch.arange(bsize), best_beam]
    lens = tp.tensor_to_lengths(best_preds, end_value=eos_id)
    best_preds = best_preds[:, : int(lens.max()) + 1].contiguous()
    return best_preds, best_avg, out_preds, out_avg


# ----------------------------------------------------------------------------------------
# a16  ids -> text -- tokenization/aac_tokenizer.py:197-209,327-384,953-963; normalizers.py
# ----------------------------------------------------------------------------------------
_POST = [
    (re.compile("(<pad>|<bos>|<eos>|<unk>)"), ""),
    (re.compile(r'\s+([,.!?;:"\'])'), r"\1"),
    None,  # strip
    (re.compile(" +"), " "),
    (re.compile(r"(\s*)(\-)(\s*)"), r"\2"),
]


def decode_ids(itos: Dict[int, str], ids: Sequence[int], lowercase: bool = True) -> str:
    s = " ".join(itos[int(i)] for i in ids)
    for step in _POST:
        if step is None:
            s = s.strip()
        else:
            s = step[0].sub(step[1], s)
    return s.lower() if lowercase else s


def decode_rec(itos: Dict[int, str], x) -> Any:
    if isinstance(x, Tensor):
        x = x.tolist()
    if len(x) > 0 and isinstance(x[0], (list, tuple)):
        return [decode_rec(itos, xi) for xi in x]
    return decode_ids(itos, x)


# ----------------------------------------------------------------------------------------
# a8/a10/a11  model forward -- huggingface/model.py:185-261, pl_modules/conette.py:352-525
# ----------------------------------------------------------------------------------------
def forbid_mask_for_mode(w: Weights, mode: Optional[str], itos: Dict[int, str]) -> Optional[Tensor]:
    """pl_modules/common.py:222-299."""
    v = w["model.decoder.classifier.weight"].shape[0]
    if mode is None:
        return w["model.forbid_rep_mask"]
    if mode == "none":
        return None
    if mode == "all":
        return torch.ones(v, dtype=torch.bool)
    if mode == "content_words":
        m = torch.ones(v, dtype=torch.bool)
        stop = set(SYNTH_STOPWORDS)
        for i, t in itos.items():
            if t in stop:
                m[int(i)] = False
        return m
    raise ValueError(f"Invalid argument {mode=}. (expected one of ('none', 'all', 'content_words'))")


def model_forward(w: Weights, cfg: Dict[str, Any], x, sr=None, x_shapes=None, preprocess=True, threshold=0.3,
                  task=None, beam_size=None, min_pred_size=None, max_pred_size=None, forbid_rep_mode=None,
                  taps=None, trace=None) -> Dict[str, Any]:
    task_names = list(cfg["task_names"])
    itos = {int(k): v for k, v in cfg["tokenizer_state"]["tokenizer"]["itos"].items()}
    if preprocess:
        batch = preprocessor_forward(w, x, sr, x_shapes, taps)
        clip_probs = batch.pop("clip_probs")
        tags = tp.probs_to_names(clip_probs, threshold, {i: f"tag{i}" for i in range(clip_probs.shape[1])})
    else:
        batch = {"audio": x, "audio_shape": x_shapes}
        clip_probs, tags = None, None
    bsize = len(batch["audio"])
    if task is None:
        tasks = [task_names[0]] * bsize
    elif isinstance(task, str):
        tasks = [task] * bsize
    elif len(task) != bsize:
        raise ValueError(f"Invalid number of tasks with input. (found {len(task)} tasks but {bsize} elements)")
    else:
        tasks = list(task)
    for t in tasks:
        if t not in task_names:
            raise ValueError(f"Invalid argument {tasks=}. (task {t} is not in {task_names})")
    names = []
    for t in tasks:
        parts = t.split("_")
        names.append(parts[0] if len(parts) < 2 else f"{parts[0]}_{'_'.join(parts[1:])}".lower())
    bos_ids = w["model.task_id_to_token_id"][torch.as_tensor([task_names.index(n) for n in names])]
    memory, mask = encode_audio(w, batch["audio"], batch["audio_shape"])
    if taps is not None:
        taps["memory"] = memory
    vocab = w["model.decoder.classifier.weight"].shape[0]
    preds, lprobs, mpreds, mlprobs = generate(
        w, memory, mask, bos_ids, pad_id=0, eos_id=2, vocab_size=vocab,
        beam_size=cfg["beam_size"] if beam_size is None else beam_size,
        min_pred_size=cfg["min_pred_size"] if min_pred_size is None else min_pred_size,
        max_pred_size=cfg["max_pred_size"] if max_pred_size is None else max_pred_size,
        forbid_rep_mask=forbid_mask_for_mode(w, forbid_rep_mode, itos),
        nhead=cfg["nhead"], n_layers=cfg["num_decoder_layers"], trace=trace)
    out = {"cands": decode_rec(itos, preds), "preds": preds, "lprobs": lprobs,
           "mult_cands": decode_rec(itos, mpreds), "mult_preds": mpreds, "mult_lprobs": mlprobs, "tasks": tasks}
    if clip_probs is not None:
        out["tags_probs"] = clip_probs
        out["tags"] = tags
    return out

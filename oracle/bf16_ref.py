"""ORACLE -- test infrastructure only.  NOT the product, never shipped, never the thing measured.

bf16-operand restatement of the kernels the benchmark runs in bf16 mode: the SAME algorithm as ``cpu_ref.py`` (and
therefore as the reference lines it cites), with every tensor that the HIP kernels feed to a bf16 MFMA rounded to bf16
(round-to-nearest-even) at the same point, fp32 accumulation and fp32 everything else.  It exists so that the bf16
kernels can be pinned tightly (rtol 3e-3 on full tensors) instead of against loose "bf16 noise" bounds; the rounding
points are part of the kernels' definition and are listed per function.

Rounding points (conette-audio-captioning_amd/csrc):
  ConvNeXt block, stages 0-2 (mlp_rc2.h):   y = bf16(LN(dwconv(x)));  W1 -> bf16;  b1 = bf16 hi + bf16 lo (exact to 2^-17);
      h = bf16(gelu(.));  W2' = bf16(scale * W2)  (LayerScale folded into the operand);  x' = x + h . W2'^T + scale * b2
  ConvNeXt block, stage 3 (gemm2.h pw1 / pw2): y, W1, h = bf16(gelu(y W1^T + b1)), W2 -> bf16;  x' = x + scale * (h W2^T + b2)
  downsample (encoder.hip cn_ln_patchify + gemm2): patches = bf16(LN(x)), conv weight -> bf16, fp32 bias
  decoder: see ``decoder_forward_bf16``.
  residual stream (round 5): inside ``operands(...)`` every tensor the 16-bit precisions' kernels STORE to the stream -- the
      stem's output, every block's x', every downsample layer's output -- is rounded to IEEE fp16 (``res16``; csrc/common.h XT);
      the depthwise convolution reads those fp16 values as exact operands and -- since the second half of round 5, when it adds two taps
      per v_dot2_f32_f16 -- its 7 x 7 weights as fp16 operands too (``res16`` of dwconv.weight: 2^-12 relative); products exact, sums fp32.
The GPU evaluates GELU through approximations that are exact to <= 5.5e-5 absolute (bf16) / 8.6e-7 (fp16) in the fused MLP
(common.h cn_gelu_e1) and 1.5e-7 elsewhere (A&S 7.1.26, common.h); the oracle uses the exact erf form, so a handful of hidden
values per million round to the neighbouring bf16.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional

import torch
from torch import Tensor
from torch.nn import functional as F

from . import cpu_ref as O

Weights = Dict[str, Tensor]


_OPERAND_DTYPE = torch.bfloat16
_RESIDUAL_DTYPE = None      # None: the fp32 residual stream (fp32 / exact / fp8 precisions, and this module outside operands())


def res16(t: Tensor) -> Tensor:
    """A tensor as the kernels store it to the residual stream: fp16 (round to nearest even) inside ``operands(...)``."""
    return t if _RESIDUAL_DTYPE is None else t.to(_RESIDUAL_DTYPE).to(torch.float32)


def bf16(t: Tensor) -> Tensor:
    """fp32 -> the 16-bit operand type (round to nearest even; bf16 unless inside ``operands("f16")``) -> fp32."""
    if _OPERAND_DTYPE is torch.float16:
        t = t.clamp(-65504.0, 65504.0)      # the kernels' conversions saturate (csrc/common.h cn_from_f32<half_t>)
    return t.to(_OPERAND_DTYPE).to(torch.float32)


@contextlib.contextmanager
def operands(kind: str):
    """Every rounding point of this module at another 16-bit operand type: "bf16" (CONETTE_PREC_BF16) or "f16"
    (CONETTE_PREC_F16: the same kernels instantiated for IEEE fp16, csrc/common.h half_t).  Both precisions keep the encoder's
    residual stream in fp16 since round 5 (``res16``)."""
    global _OPERAND_DTYPE, _RESIDUAL_DTYPE
    keep = (_OPERAND_DTYPE, _RESIDUAL_DTYPE)
    _OPERAND_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16}[kind]
    _RESIDUAL_DTYPE = torch.float16
    try:
        yield
    finally:
        _OPERAND_DTYPE, _RESIDUAL_DTYPE = keep


def convnext_block_bf16(w: Weights, prefix: str, x: Tensor, folded: bool) -> Tensor:
    """nn/encoders/convnext.py:61-74 with bf16 GEMM operands.  x: (B, C, H, W) fp32 values of the residual stream."""
    c = x.shape[1]
    y = F.conv2d(x, res16(w[prefix + "dwconv.weight"]), w[prefix + "dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = bf16(F.layer_norm(y, (c,), w[prefix + "norm.weight"], w[prefix + "norm.bias"], 1e-6))
    h = F.linear(y, bf16(w[prefix + "pwconv1.weight"])) + w[prefix + "pwconv1.bias"]
    h = bf16(F.gelu(h))
    s = w[prefix + "scale_layer"]
    if folded:
        z = F.linear(h, bf16(s[:, None] * w[prefix + "pwconv2.weight"])) + s * w[prefix + "pwconv2.bias"]
    else:
        z = s * (F.linear(h, bf16(w[prefix + "pwconv2.weight"])) + w[prefix + "pwconv2.bias"])
    return res16(x + z.permute(0, 3, 1, 2))


def downsample_bf16(w: Weights, i: int, x: Tensor, folded: bool = False) -> Tensor:
    """downsample_layers[i], i = 1..3 (convnext.py:207-217): LayerNorm(channels_first) + 2x2/2 conv, bf16 operands.

    folded (csrc/down_fused.h, stage 0 -> 1): the LayerNorm affine lives in the packed weights -- the activation operand
    is bf16((x - mean) rstd), the weight operand bf16(W g) (g per input channel) and the bias is b + sum W beta in fp32."""
    d = f"preprocessor.encoder.downsample_layers.{i}."
    if not folded:
        y = bf16(O._ln_cf(x, w[d + "0.weight"], w[d + "0.bias"]))
        return res16(F.conv2d(y, bf16(w[d + "1.weight"]), w[d + "1.bias"], stride=2))
    g, beta, W, b = w[d + "0.weight"].float(), w[d + "0.bias"].float(), w[d + "1.weight"].float(), w[d + "1.bias"].float()
    y = bf16(O._ln_cf(x, torch.ones_like(g), torch.zeros_like(beta)))
    bias = b + (W * beta.view(1, -1, 1, 1)).sum(dim=(1, 2, 3))
    return res16(F.conv2d(y, bf16(W * g.view(1, -1, 1, 1)), bias, stride=2))


def block_prefix(blk: int) -> str:
    i = 0
    for st, depth in enumerate(O.DEPTHS):
        if blk < i + depth:
            return f"preprocessor.encoder.stages.{st}.{blk - i}."
        i += depth
    raise IndexError(blk)


def block_stage(blk: int) -> int:
    i = 0
    for st, depth in enumerate(O.DEPTHS):
        if blk < i + depth:
            return st
        i += depth
    raise IndexError(blk)


# ----------------------------------------------------------------------------------------
# decoder (nn/decoders/aac_tfmer.py:71-118) with the bf16 kernels' rounding points
# (decoder.hip / dec_block.h / dec_ffn.h): every GEMM input activation, every weight matrix, the projected audio
# memory, the cross- and self-attention K / V (cache precision, own entry included), the attention outputs and the
# FFN hidden are bf16; q, scores, softmax, residual stream, LayerNorms, biases, logits are fp32.
# ----------------------------------------------------------------------------------------
def encode_audio_bf16(w: Weights, audio: Tensor) -> Tensor:
    """(B, T, 768) -> projected memory (B, T, 256): bf16(relu(bf16(audio) . bf16(Wp)^T + bp))  (conette.py:457)."""
    return bf16(F.relu(F.linear(bf16(audio), bf16(w["model.projection.2.weight"]), w["model.projection.2.bias"])))


def _attend(q: Tensor, k: Tensor, v: Tensor, nhead: int, mask: Optional[Tensor]) -> Tensor:
    """q (R, tq, d) fp32 (already scaled), k / v (R, tk, d) at cache precision, mask (R, tq, tk) bool True = masked."""
    r, tq, e = q.shape
    tk = k.shape[1]
    dh = e // nhead
    qh = q.view(r, tq, nhead, dh).transpose(1, 2)
    kh = k.view(r, tk, nhead, dh).transpose(1, 2)
    vh = v.view(r, tk, nhead, dh).transpose(1, 2)
    sc = torch.matmul(qh, kh.transpose(2, 3))
    if mask is not None:
        sc = sc.masked_fill(mask[:, None], float("-inf"))
    return torch.matmul(torch.softmax(sc, dim=-1), vh).transpose(1, 2).reshape(r, tq, e)


@torch.no_grad()
def teacher_forcing_bf16(w: Weights, audio: Tensor, audio_shape: Tensor, caps_in: Tensor, *, pad_id: int = 0,
                         nhead: int = 8, n_layers: int = 6) -> Tensor:
    """Same contract as cpu_ref.teacher_forcing (forcing.py:12-71): logits (B, V, t), bf16-operand arithmetic."""
    D = "model.decoder."
    d = w[D + "emb_layer.weight"].shape[1]
    b, t = caps_in.shape
    dh = d // nhead
    scale = 1.0 / (dh ** 0.5)
    mem = encode_audio_bf16(w, audio)                                  # (B, T, d)
    ta = mem.shape[1]
    lens = audio_shape[:, 1].clamp(1, ta)
    mem_mask = (torch.arange(ta)[None, :] >= lens[:, None])[:, None, :].expand(b, t, ta)
    causal = torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)[None].expand(b, t, t)
    self_mask = causal | caps_in.eq(pad_id)[:, None, :]
    x = F.embedding(caps_in, w[D + "emb_layer.weight"]) * (d ** 0.5) + w[D + "pos_encoding.pos_embedding"][:t, 0][None]
    for l in range(n_layers):
        p = D + f"layers.{l}."
        qkv = F.linear(bf16(x), bf16(w[p + "self_attn.in_proj_weight"]), w[p + "self_attn.in_proj_bias"])
        q, k, v = qkv[..., :d] * scale, bf16(qkv[..., d : 2 * d]), bf16(qkv[..., 2 * d :])
        a = bf16(_attend(q, k, v, nhead, self_mask))
        x = F.layer_norm(x + F.linear(a, bf16(w[p + "self_attn.out_proj.weight"]), w[p + "self_attn.out_proj.bias"]),
                         (d,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
        wi, bi = w[p + "multihead_attn.in_proj_weight"], w[p + "multihead_attn.in_proj_bias"]
        q2 = F.linear(bf16(x), bf16(wi[:d]), bi[:d]) * scale
        k2 = bf16(F.linear(mem, bf16(wi[d : 2 * d]), bi[d : 2 * d]))
        v2 = bf16(F.linear(mem, bf16(wi[2 * d :]), bi[2 * d :]))
        c = bf16(_attend(q2, k2, v2, nhead, mem_mask))
        x = F.layer_norm(x + F.linear(c, bf16(w[p + "multihead_attn.out_proj.weight"]), w[p + "multihead_attn.out_proj.bias"]),
                         (d,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
        h = bf16(F.gelu(F.linear(bf16(x), bf16(w[p + "linear1.weight"]), w[p + "linear1.bias"])))
        x = F.layer_norm(x + F.linear(h, bf16(w[p + "linear2.weight"]), w[p + "linear2.bias"]),
                         (d,), w[p + "norm3.weight"], w[p + "norm3.bias"], 1e-5)
    logits = F.linear(bf16(x), bf16(w[D + "classifier.weight"]), w[D + "classifier.bias"])
    return logits.permute(0, 2, 1)

"""ORACLE -- test infrastructure only.  NOT the product, never shipped, never the thing measured.

fp8-operand restatement of the ConvNeXt block as the experimental fp8 precision of rounds 3-5 ran it at stages 0-2 (now
tools/lab/mlp_f8.h: the precision was WITHDRAWN in round 6, profiles/r06_notes.md section 3; BASELINE.json configs[4] "fp8 MFMA
pointwise GEMMs").  Kept as the per-tensor-scaled arm of oracle/mx_study.py: the algorithm of ``cpu_ref.convnext_block`` (nn/encoders/convnext.py:61-74) with
the kernel's quantisation points, fp32 accumulation and fp32 everything else:

  y8   = e4m3(LN(dwconv(x)))                                   (scale 1; the depthwise-conv kernel stores y as e4m3)
  W1q  = e4m3(W1 / is1),   is1 = max|W1| / 448                  (one scale per matrix)
  hid  = (y8 . W1q^T) * is1 + b1                                (b1 fp32)
  h8   = e4m3(gelu(hid))                                        (scale 1)
  W2q  = e4m3(ls[c] W2[c][:] / sc[c]),  sc[c] = max_k |ls[c] W2[c][k]| / 448   (one scale per output channel; absorbs LayerScale)
  x'   = sc * (x / sc + h8 . W2q^T) + ls * b2

e4m3 is the OCP "fn" format (max 448, no infinities), round to nearest even -- torch.float8_e4m3fn on the CPU.  With 3
mantissa bits a value that lands on the other side of a rounding boundary moves by 1/8 of its magnitude, so the GELU is
restated in the kernel's own form -- x sigmoid(x (a + b x^2 + c x^4)), mlp_rc2.h cn_gelu_sig2: 2.5e-5 from erf, which
would flip about one hidden value per position -- and what is left are the y values whose LayerNorm (computed to ~1e-6 by
both sides) straddles a boundary: ~1e-3 of the positions carry one flipped operand, and the test budgets for them.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor
from torch.nn import functional as F

Weights = Dict[str, Tensor]
E4M3_MAX = 448.0


def e4m3(t: Tensor) -> Tensor:
    """fp32 -> OCP e4m3 (round to nearest even, saturating like v_cvt_pk_fp8_f32) -> fp32."""
    return t.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def gelu_sig2(x: Tensor) -> Tensor:
    """csrc/mlp_rc2.h cn_gelu_sig2: x * sigmoid(x (a + b x^2 + c x^4)), x^2 clamped at 64 (minimax fit of the normal CDF's
    logit on [-8, 8]; max |error| against the erf form 2.5e-5)."""
    x2 = (x * x).clamp(max=64.0)
    p = (0.0007030350670982541 * x2 - 0.07401130190658815) * x2 - 1.5950157568571721
    return x / (1.0 + torch.exp(x * p))


def convnext_block_fp8(w: Weights, prefix: str, x: Tensor) -> Tensor:
    """x: (B, C, H, W) fp32 residual stream -> the block's output, e4m3 operands in both pointwise convolutions."""
    c = x.shape[1]
    y = F.conv2d(x, w[prefix + "dwconv.weight"], w[prefix + "dwconv.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y8 = e4m3(F.layer_norm(y, (c,), w[prefix + "norm.weight"], w[prefix + "norm.bias"], 1e-6))
    w1 = w[prefix + "pwconv1.weight"].float()
    is1 = w1.abs().max() / E4M3_MAX
    w1q = e4m3(w1 * (1.0 / is1))
    hid = F.linear(y8, w1q) * is1 + w[prefix + "pwconv1.bias"]
    h8 = e4m3(gelu_sig2(hid))
    ls = w[prefix + "scale_layer"].float()
    w2 = w[prefix + "pwconv2.weight"].float()
    sc = (ls[:, None] * w2).abs().amax(dim=1) / E4M3_MAX
    isc = 1.0 / sc
    w2q = e4m3((ls * isc)[:, None] * w2)
    o = x.permute(0, 2, 3, 1) * isc + F.linear(h8, w2q)
    out = o * sc + ls * w[prefix + "pwconv2.bias"]
    return out.permute(0, 3, 1, 2)

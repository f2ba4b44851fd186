"""ORACLE -- test infrastructure only.  NOT the product, never shipped, never the thing measured.

Host study for BASELINE configs[4]'s "fp8 MFMA pointwise GEMMs" (VERDICT r05 next 6): what do BLOCK-SCALED e4m3 operands -- the
operand form of v_mfma_scale_f32_16x16x128_f8f6f4: 32 consecutive k-elements share one e8m0 (power-of-two) scale, OCP MX --
cost the frame embeddings, before any kernel is written?  The pointwise convolutions of the chosen ConvNeXt stages take MX-e4m3
operands at the four points a fused kernel would quantise (y = LN output, W1, gelu output, LayerScale x W2; fp32 accumulation);
everything else runs as the bf16 throughput mode does (bf16 operands, fp16 residual stream: oracle/bf16_ref.py).
    python -m oracle.mx_study [n_clips]
prints the frame embeddings' rel. rms against the fp32 oracle for: the bf16 mode, MX-e4m3 in stage 2 only, in stages 0-2, and the
per-tensor-scaled e4m3 of the withdrawn fp8 precision (oracle/fp8_ref.py) in stages 0-2.
"""
import importlib
import sys

import torch
from torch.nn import functional as F

from . import bf16_ref as R
from . import cpu_ref as O
from . import fp8_ref as F8

synth = importlib.import_module("conette-audio-captioning_amd.synth")


def mx_e4m3(t: torch.Tensor) -> torch.Tensor:
    """OCP MX: blocks of 32 along the last dim share the scale 2^(floor(log2 amax) - 8) (e4m3: emax 8, max 448); elements e4m3."""
    k = t.shape[-1]
    assert k % 32 == 0
    b = t.reshape(*t.shape[:-1], k // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -120)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (b / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale
    return q.reshape(t.shape)


def block_mx(w, prefix, x):
    """ConvNeXt block (convnext.py:61-74) with MX-e4m3 operands in both pointwise convolutions; x (B, C, H, W) fp16-stream values."""
    c = x.shape[1]
    wd = R.res16(w[prefix + "dwconv.weight"])
    y = F.conv2d(x, wd, w[prefix + "dwconv.bias"], padding=3, groups=c).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), w[prefix + "norm.weight"], w[prefix + "norm.bias"], 1e-6)
    hid = F.linear(mx_e4m3(y), mx_e4m3(w[prefix + "pwconv1.weight"].float())) + w[prefix + "pwconv1.bias"]
    g = F.gelu(hid)
    ls = w[prefix + "scale_layer"].float()
    o = F.linear(mx_e4m3(g), mx_e4m3(ls[:, None] * w[prefix + "pwconv2.weight"].float())) + ls * w[prefix + "pwconv2.bias"]
    return x + o.permute(0, 3, 1, 2)


def encode(w, wave, mx_stages=(), f8_stages=()):
    p = "preprocessor.encoder."
    x = O.logmel_bn0(w, wave)
    for i in range(4):
        d = p + f"downsample_layers.{i}."
        if i == 0:
            x = F.conv2d(x, w[d + "0.weight"], w[d + "0.bias"], stride=(4, 4), padding=(4, 0))
            x = R.res16(O._ln_cf(x, w[d + "1.weight"], w[d + "1.bias"]))
        else:
            x = R.res16(R.downsample_bf16(w, i, x, folded=(i < 3)))
        for b in range(O.DEPTHS[i]):
            pre = p + f"stages.{i}.{b}."
            if i in mx_stages:
                x = R.res16(block_mx(w, pre, x))
            elif i in f8_stages:
                x = R.res16(F8.convnext_block_fp8(w, pre, x))
            else:
                x = R.res16(R.convnext_block_bf16(w, pre, x, folded=(i < 3)))
    return torch.mean(x, dim=3).transpose(1, 2).contiguous()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(8)
    sd = O.to_torch(synth.synth_state_dict())
    wave = torch.from_numpy(synth.synth_waveforms(n, 320000, 1234))
    with torch.no_grad():
        shapes = torch.as_tensor([[1, wave.shape[1]]] * n)
        ref = O.convnext_encode(sd, wave, shapes)["frame_embs"].transpose(1, 2).contiguous()   # (B, T, 768)
        rows = [("bf16 mode (bf16 operands, fp16 stream)", encode(sd, wave)),
                ("MX-e4m3 pointwise, stage 2 only", encode(sd, wave, mx_stages=(2,))),
                ("MX-e4m3 pointwise, stages 0-2", encode(sd, wave, mx_stages=(0, 1, 2))),
                ("per-tensor e4m3 pointwise (the withdrawn fp8 precision's points), stages 0-2", encode(sd, wave, f8_stages=(0, 1, 2)))]
    for name, fe in rows:
        rel = float((fe - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"{name:70s} frame_embs rel rms {rel:.3e}", flush=True)


if __name__ == "__main__":
    main()

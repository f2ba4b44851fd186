"""ORACLE -- test infrastructure only.  NOT the product, never shipped, never the thing measured.

Study, on the host: how many greedy / beam captions of the fp32 oracle survive when the encoder's MFMA operands are rounded
to bf16 or to fp16 (same rounding points as bf16_ref.py), the decoder staying fp32.  Usage:
    python -m oracle.rounding_study [n_clips]
"""
import importlib
import sys
import time

import torch
from torch.nn import functional as F

from . import bf16_ref as R
from . import cpu_ref as O

synth = importlib.import_module("conette-audio-captioning_amd.synth")


def encode_rounded(w, wave, rnd, res=lambda t: t):
    """convnext_encode (cpu_ref.py) with the 16-bit kernels' operand rounding points, `rnd` in place of bf16(); `res` rounds the
    residual stream wherever the kernels store it (stem, every block, every downsample layer: round 5's fp16 stream)."""
    keep = R.bf16
    R.bf16 = rnd
    try:
        p = "preprocessor.encoder."
        x = O.logmel_bn0(w, wave)
        blk = 0
        for i in range(4):
            d = p + f"downsample_layers.{i}."
            if i == 0:
                x = F.conv2d(x, w[d + "0.weight"], w[d + "0.bias"], stride=(4, 4), padding=(4, 0))
                x = res(O._ln_cf(x, w[d + "1.weight"], w[d + "1.bias"]))
            else:
                x = res(R.downsample_bf16(w, i, x, folded=(i < 3)))
            for b in range(O.DEPTHS[i]):
                x = res(R.convnext_block_bf16(w, p + f"stages.{i}.{b}.", x, folded=(i < 3)))
                blk += 1
        return torch.mean(x, dim=3).transpose(1, 2).contiguous()
    finally:
        R.bf16 = keep


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(8)
    sd = O.to_torch(synth.synth_state_dict())
    wave = torch.from_numpy(synth.synth_waveforms(n, 320000, 1234))
    lens = torch.full((n,), 320000)
    shapes = torch.stack([torch.ones(n, dtype=torch.long), lens], 1)
    bos = sd["model.task_id_to_token_id"][torch.zeros(n, dtype=torch.long)]
    vocab = sd["model.decoder.classifier.weight"].shape[0]
    b16, f16 = (lambda t: t.to(torch.bfloat16).float()), (lambda t: t.to(torch.float16).float())
    ident = lambda t: t
    rounders = {"fp32": (ident, ident), "bf16": (b16, ident), "f16": (f16, ident), "bf16+res16": (b16, f16), "f16+res16": (f16, f16),
                "bf16+resb16": (b16, b16)}
    out = {}
    with torch.no_grad():
        for name, (rnd, res) in rounders.items():
            t0 = time.time()
            fe = encode_rounded(sd, wave, rnd, res)
            ashape = torch.as_tensor([[768, fe.shape[1]]] * n)
            mem, mask = O.encode_audio(sd, fe, ashape)
            res = {}
            for beam in (1, 3):
                g = O.generate(sd, mem, mask, bos, vocab_size=vocab, beam_size=beam, min_pred_size=3, max_pred_size=20,
                               forbid_rep_mask=sd["model.forbid_rep_mask"])
                res[beam] = g[0]
            out[name] = (fe, res)
            print(name, f"{time.time() - t0:.1f}s", flush=True)
    fe0, r0 = out["fp32"]
    for name in list(rounders)[1:]:
        fe, r = out[name]
        rel = float((fe - fe0).pow(2).mean().sqrt() / fe0.pow(2).mean().sqrt())
        same = {b: sum(int(torch.equal(torch.as_tensor(a), torch.as_tensor(c))) for a, c in zip(r[b], r0[b])) for b in (1, 3)}
        print(f"{name}: frame_embs rel rms {rel:.2e}; greedy identical {same[1]}/{n}; beam3 identical {same[3]}/{n}")


if __name__ == "__main__":
    main()

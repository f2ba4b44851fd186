"""CPU experiment (oracle arithmetic, no GPU): which bf16 rounding points of the decoder cost the logits how much?
Teacher forcing on the committed ragged fixture; every group of rounding points of oracle/bf16_ref.teacher_forcing_bf16
switched on alone, and all but that group; error of the logits against the fp32 oracle (cpu_ref.teacher_forcing)."""
import os, sys
import numpy as np, torch
from torch.nn import functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import conette_amd  # noqa
from conette_amd import synth
from oracle import cpu_ref as O
from oracle.bf16_ref import _attend

GROUPS = ["weights", "gemm_in", "kv", "attn_out", "ffn_hidden", "memory", "classifier"]


def run(w, audio, audio_shape, caps_in, on):
    r = lambda g, t: t.to(torch.bfloat16).float() if g in on else t
    D = "model.decoder."; d = 256; nhead = 8; b, t = caps_in.shape; scale = 1.0 / (32 ** 0.5)
    mem = r("memory", F.relu(F.linear(r("memory", audio), r("memory", w["model.projection.2.weight"]), w["model.projection.2.bias"])))
    ta = mem.shape[1]; lens = audio_shape[:, 1].clamp(1, ta)
    mem_mask = (torch.arange(ta)[None, :] >= lens[:, None])[:, None, :].expand(b, t, ta)
    causal = torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)[None].expand(b, t, t)
    self_mask = causal | caps_in.eq(0)[:, None, :]
    x = F.embedding(caps_in, w[D + "emb_layer.weight"]) * 16.0 + w[D + "pos_encoding.pos_embedding"][:t, 0][None]
    for l in range(6):
        p = D + f"layers.{l}."
        qkv = F.linear(r("gemm_in", x), r("weights", w[p + "self_attn.in_proj_weight"]), w[p + "self_attn.in_proj_bias"])
        q, k, v = qkv[..., :d] * scale, r("kv", qkv[..., d:2 * d]), r("kv", qkv[..., 2 * d:])
        a = r("attn_out", _attend(q, k, v, nhead, self_mask))
        x = F.layer_norm(x + F.linear(a, r("weights", w[p + "self_attn.out_proj.weight"]), w[p + "self_attn.out_proj.bias"]), (d,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
        wi, bi = w[p + "multihead_attn.in_proj_weight"], w[p + "multihead_attn.in_proj_bias"]
        q2 = F.linear(r("gemm_in", x), r("weights", wi[:d]), bi[:d]) * scale
        k2 = r("kv", F.linear(mem, r("weights", wi[d:2 * d]), bi[d:2 * d]))
        v2 = r("kv", F.linear(mem, r("weights", wi[2 * d:]), bi[2 * d:]))
        c = r("attn_out", _attend(q2, k2, v2, nhead, mem_mask))
        x = F.layer_norm(x + F.linear(c, r("weights", w[p + "multihead_attn.out_proj.weight"]), w[p + "multihead_attn.out_proj.bias"]), (d,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
        h = r("ffn_hidden", F.gelu(F.linear(r("gemm_in", x), r("weights", w[p + "linear1.weight"]), w[p + "linear1.bias"])))
        x = F.layer_norm(x + F.linear(h, r("weights", w[p + "linear2.weight"]), w[p + "linear2.bias"]), (d,), w[p + "norm3.weight"], w[p + "norm3.bias"], 1e-5)
    return F.linear(r("classifier", x), r("classifier", w[D + "classifier.weight"]), w[D + "classifier.bias"])


def main():
    torch.set_num_threads(8)
    w = O.to_torch(synth.synth_state_dict())
    g = np.load(os.path.join(ROOT, "tests", "golden", "forcing", "forcing_ragged.npz"))
    audio, shp, caps = torch.from_numpy(g["frame_embs"]), torch.from_numpy(g["audio_shape"]), torch.from_numpy(g["caps_in"]).long()
    valid = (caps != 0)[:, :, None]
    with torch.no_grad():
        ref = run(w, audio, shp, caps, set())
        # difference of the top-2 logits is what a decision sees: error of (logit - top logit) for the reference's top 5
        top = ref.topk(5, dim=-1).indices
        def err(lg):
            e = (lg - ref) * valid
            rel = ((lg.gather(-1, top) - lg.gather(-1, top[..., :1])) - (ref.gather(-1, top) - ref.gather(-1, top[..., :1]))) * valid
            return float(e.abs().max()), float(e.abs().mean()), float(rel.abs().max()), float(rel.abs().mean())
        print(f"{'rounded to bf16':40s} max|dlogit| mean|dlogit|  max|d(gap)| mean|d(gap)| (gap = logit - top logit, reference's top 5)")
        for name, on in [("everything (the bf16 mode)", set(GROUPS))] + [(f"only {x}", {x}) for x in GROUPS] + [(f"all but {x}", set(GROUPS) - {x}) for x in GROUPS]:
            print(f"{name:40s} " + "  ".join(f"{v:10.4f}" for v in err(run(w, audio, shp, caps, on))))


if __name__ == "__main__":
    main()

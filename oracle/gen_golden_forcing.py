"""Golden vectors for the teacher-forcing path, produced by the reference itself.

Imports the reference from /root/reference (oracle/refshim.py), loads the seeded synthetic checkpoint, encodes three
ragged clips with the reference's preprocessor, builds input captions from the reference's own beam-search output
(task token + words, right-padded) and records ``CoNeTTEPLM.decode_audio(encoder_outs, "forcing", caps_in=...)``
(pl_modules/conette.py:392-417 -> nn/decoding/forcing.py:12-71).  Data only: tests/golden/forcing/*.npz.

    python oracle/gen_golden_forcing.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conette_amd  # noqa: E402,F401
from conette_amd import synth  # noqa: E402
from oracle import refshim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "forcing")
SR = 32000


def main() -> None:
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    R = refshim.ref()
    import conette.huggingface.model as hm

    hm.load_audioset_idx_to_name = lambda offline=False, verbose=0: {i: f"tag{i}" for i in range(527)}
    cfg = R.CoNeTTEConfig(**synth.synth_config_dict())
    model = R.CoNeTTEModel(cfg, device="cpu", offline=True)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict().items()}
    sd["_extra_state_"] = torch.from_numpy(synth.extra_state_tensor())
    model.load_state_dict(sd, strict=True)
    plm = model.model

    lengths = [3 * SR, 2 * SR + 1234, SR]
    seed0 = 8100
    wav = synth.synth_waveforms(len(lengths), max(lengths), seed0, lengths=lengths)
    x = [torch.from_numpy(wav[i, : lengths[i]].copy())[None, :] for i in range(len(lengths))]
    tasks = ["clotho", "audiocaps", "clotho"]
    with torch.no_grad():
        pre = model.preprocessor(x, SR, None)
        out = model(x, sr=SR, task=tasks, max_pred_size=12 - 2 * 0)
        bos = plm.batch_to_task_token_ids({"dataset": [t.split("+")[0] for t in tasks], "source": [None] * 3})
        preds = out["preds"]
        rows = []
        for b in range(len(lengths)):
            words = [int(t) for t in preds[b].tolist() if t not in (plm.pad_id, plm.bos_id, plm.eos_id)]
            words = words[: 9 - 3 * b]  # different caption lengths -> padded positions
            rows.append([int(bos[b])] + words)
        cap_len = max(len(r) for r in rows)
        caps_in = torch.full((len(rows), cap_len), plm.pad_id, dtype=torch.long)
        for b, r in enumerate(rows):
            caps_in[b, : len(r)] = torch.as_tensor(r)
        enc = plm.encode_audio(pre["audio"], pre["audio_shape"])
        logits = plm.decode_audio(enc, "forcing", caps_in=caps_in)  # (B, V, cap_len)
    rec = dict(lengths=np.asarray(lengths, dtype=np.int64), seed0=np.int64(seed0), tasks=np.asarray(json.dumps(tasks)),
               caps_in=caps_in.numpy(), frame_embs=pre["audio"].numpy(), audio_shape=pre["audio_shape"].numpy(),
               logits=logits.numpy().astype(np.float32))
    np.savez_compressed(os.path.join(GOLD, "forcing_ragged.npz"), **rec)
    print("caps_in", caps_in.tolist(), "logits", tuple(logits.shape), float(logits.abs().max()))

    # ---- greedy_search (nn/decoding/greedy.py:17-131): full masked logits of the arg-max chain
    from conette.nn.decoding.greedy import greedy_search
    with torch.no_grad():
        # plain <bos> (BaselinePLM has no task token; the synthetic decoder then never stops: all 12 steps), and the
        # audiocaps task token as first token (one caption stops early: the finished-clip fill pattern)
        for name, bos_tok, fmask, maxp in (("greedy_bos", plm.bos_id, plm.forbid_rep_mask, 12),
                                           ("greedy_task", int(bos[1]), plm.forbid_rep_mask, 12)):
            glog = greedy_search(plm.decoder, plm.pad_id, bos_tok, plm.eos_id, plm.tokenizer.get_vocab_size(),
                                 enc["frame_embs"], enc["frame_embs_pad_mask"], min_pred_size=3, max_pred_size=maxp,
                                 forbid_rep_mask=fmask)
            grec = dict(lengths=np.asarray(lengths, dtype=np.int64), seed0=np.int64(seed0),
                        frame_embs=pre["audio"].numpy(), audio_shape=pre["audio_shape"].numpy(),
                        bos_id=np.int64(bos_tok), min_pred=np.int64(3), max_pred=np.int64(maxp),
                        use_forbid=np.int64(fmask is not None), logits=glog.numpy().astype(np.float32))
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), **grec)
            print(name, tuple(glog.shape), glog.argmax(dim=1).tolist())


if __name__ == "__main__":
    main()

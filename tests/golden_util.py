"""Helpers shared by the oracle and GPU parity tests: rebuild a golden scenario's inputs."""
import json
import os

import numpy as np
import torch

from conette_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENARIOS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))
TAPS = ["logmel", "stem", "stage0_block0", "stage0", "down1", "stage1", "down2", "stage2", "down3", "stage3"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def inputs(g):
    n = [int(v) for v in g["lengths"]]
    wav = synth.synth_waveforms(len(n), max(n), int(g["seed0"]), lengths=n)
    form = str(g["form"])
    kw = json.loads(str(g["kw"]))
    if form == "tensor3":
        x = torch.from_numpy(wav)[:, None, :]
    elif form == "tensor1":
        x = torch.from_numpy(wav[0])
    else:
        x = [torch.from_numpy(wav[i, : n[i]].copy())[None, :] for i in range(len(n))]
    return x, kw


def sub(x, maxel=20000):
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // maxel)
    return f[::step].cpu().float().numpy()


def trace_of(g):
    return (json.loads(str(g["trace_parent"])), json.loads(str(g["trace_token"])),
            json.loads(str(g["trace_sum"])), g["trace_margin"])


def first_risky_step(g, tol):
    """Number of leading _select_k_next_toks calls whose top-k margin exceeds tol."""
    m = g["trace_margin"]
    bad = np.nonzero(m <= tol)[0]
    return int(bad[0]) if bad.size else len(m)


def topk_at_reference_states(g, calls, eng, weights, cfg, forbid, tol, sum_atol, tag=""):
    """Every _select_k_next_toks call of the reference's beam search (beam.py:230-269), checked INDEPENDENTLY of the others: the
    decoder kernels are fed the reference's own prefixes (teacher forcing through the KV-cached step kernels the search itself
    runs), the step's masking / log-softmax / running sums / flat top-k are restated here, and the picks must equal the
    reference's wherever its effective top-(k+1) margin exceeds `tol`.  calls: [(step, clip, parents, tokens, sums, margin)] in
    the reference's call order.  Returns (calls above the margin verified, calls identical incl. near-ties, calls above the margin)."""
    kw = json.loads(str(g["kw"]))
    bsz = len(g["lengths"])
    beam = kw.get("beam_size", cfg["beam_size"])
    min_pred = kw.get("min_pred_size", cfg["min_pred_size"])
    max_pred = kw.get("max_pred_size", cfg["max_pred_size"])
    tasks = json.loads(str(g["tasks"]))
    task_names = list(cfg["task_names"])
    bos = weights["model.task_id_to_token_id"][torch.as_tensor([task_names.index(t) for t in tasks])].tolist()
    v = eng.vocab_size
    forbid = torch.zeros(v, dtype=torch.bool) if forbid is None else forbid.bool()
    state = {j: ([[bos[j]] for _ in range(beam)], [0.0] * beam) for j in range(bsz)}
    items, rows_clip, rows_caps = [], [], []
    for step, clip, par, tok, sums, margin in calls:
        pre, sm = state[clip]
        use = pre[:1] if step == 0 else pre
        items.append((len(rows_clip), len(use), list(sm)))
        for p_ in use:
            rows_clip.append(clip)
            rows_caps.append(p_ + [0] * (max_pred - len(p_)))
        newp = [pre[p_] + [t] for p_, t in zip(par, tok)]
        keep = [i for i, t in enumerate(tok) if not (t == 2 or step == max_pred - 1)]
        state[clip] = ([newp[i] for i in keep], [sums[i] for i in keep])
    fe = torch.from_numpy(g["frame_embs"])[rows_clip].cuda()
    lens = torch.from_numpy(g["audio_shape"][:, 1].astype(np.int32))[rows_clip]
    caps = torch.as_tensor(rows_caps, dtype=torch.int64)
    eng.set_forcing_stepwise(True)   # the KV-cached step kernels the search itself runs (fused block / FFN kernels in bf16)
    try:
        logits = eng.forcing(fe, lens, caps).cpu()                    # (rows, max_pred, V)
    finally:
        eng.set_forcing_stepwise(False)
    n_checked = n_same = n_eligible = 0
    for (r0, n_rows, sm), (step, clip, par, tok, sums, margin) in zip(items, calls):
        lg = logits[r0 : r0 + n_rows, step].clone()
        if step < min_pred:
            lg[:, 2] = -float("inf")
        for i in range(n_rows):
            seen = torch.zeros(v, dtype=torch.bool)
            seen[torch.as_tensor(rows_caps[r0 + i][: step + 1])] = True
            lg[i, seen & forbid] = -float("inf")
        cand = torch.log_softmax(lg, dim=1)
        if step > 0:
            cand = cand + torch.as_tensor(sm[:n_rows])[:, None]
        k = len(par)
        vals, flat = torch.topk(cand.reshape(-1), k)
        eff = min([margin] + [sums[i] - sums[i + 1] for i in range(k - 1)])
        np.testing.assert_allclose(vals.numpy(), sums, atol=sum_atol(step))
        same = (flat // v).tolist() == par and (flat % v).tolist() == tok
        n_same += same
        if eff <= tol:
            continue
        n_eligible += 1
        assert same, (tag, step, clip)
        n_checked += 1
    return n_checked, n_same, n_eligible

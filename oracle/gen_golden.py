"""ORACLE tooling: generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container: imports the reference's own Python from /root/reference
(oracle/refshim.py), loads the seeded synthetic checkpoint (conette_amd.synth) into the
reference's ``CoNeTTEModel`` and records inputs + outputs of the hot path per scenario.
The fixtures are data (numbers / strings), never reference source.

    python -m oracle.gen_golden            # writes tests/golden/*.npz + api_cases.json
    python -m oracle.gen_golden --only=a,b # only the named scenarios
"""
from __future__ import annotations

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

GOLD = os.path.join(ROOT, "tests", "golden")
SR = 32000

# name -> (lengths in seconds, seed0, call kwargs, input form)
SCENARIOS = {
    "b4_10s_beam3_clotho": dict(secs=[10, 10, 10, 10], seed0=1234, kw=dict(task="clotho"), form="tensor3"),
    "b4_10s_beam1_audiocaps": dict(secs=[10, 10, 10, 10], seed0=1234, kw=dict(task="audiocaps", beam_size=1), form="tensor3"),
    "b3_mixed_beam3_none": dict(secs=[2, 5, 10], seed0=2000, kw=dict(task="clotho", forbid_rep_mode="none"), form="list"),
    "b3_mixed_beam2_tasks": dict(secs=[2, 5, 10], seed0=2000, kw=dict(task=["clotho", "audiocaps", "wavcaps_bbc_sound_effects"], beam_size=2), form="list"),
    "b2_1s_beam3": dict(secs=[1, 1], seed0=3000, kw=dict(task="macs", min_pred_size=1, max_pred_size=12), form="tensor3"),
    "b1_30s_beam3": dict(secs=[30], seed0=4000, kw=dict(task="wavcaps_freesound"), form="tensor1"),
    "b8_10s_beam3_all": dict(secs=[10] * 8, seed0=5000, kw=dict(task="clotho", forbid_rep_mode="all"), form="tensor3"),
    # round 3: odd lengths (no multiple of the hop or of the 32x reduction), wide beams, min_pred 0, long max_pred, the
    # content_words mask; beam 8 is the library's CN_MAX_BEAM
    "b4_odd_beam5_minpred0": dict(secs=[7.013, 3.3, 12.47, 5.5], seed0=6000,
                                  kw=dict(task=["clotho", "macs", "audiocaps", "wavcaps_soundbible"], beam_size=5, min_pred_size=0,
                                          max_pred_size=30), form="list"),
    "b2_4s_beam8_content_words": dict(secs=[4, 4], seed0=7000, kw=dict(task="audiocaps", beam_size=8, forbid_rep_mode="content_words",
                                                                        max_pred_size=16), form="tensor3"),
    # round 5: the "peaked" synthetic checkpoint (synth.py PEAKED: few candidates far above a noise floor, like a trained
    # captioner) -- the fixtures on which the 16-bit precisions' token ids are held to the reference's (tests/test_gpu_peaked.py);
    # written to tests/golden/peaked/
    "pk_b8_10s_beam1_clotho": dict(secs=[10] * 8, seed0=5000, kw=dict(task="clotho", beam_size=1), form="tensor3", recipe="peaked"),
    "pk_b8_10s_beam3_clotho": dict(secs=[10] * 8, seed0=5000, kw=dict(task="clotho"), form="tensor3", recipe="peaked"),
    "pk_b4_mixed_beam1_tasks": dict(secs=[3, 6.5, 10, 14.2], seed0=8000,
                                    kw=dict(task=["clotho", "audiocaps", "macs", "wavcaps_freesound"], beam_size=1), form="list", recipe="peaked"),
    "pk_b4_mixed_beam3_tasks": dict(secs=[3, 6.5, 10, 14.2], seed0=8000,
                                    kw=dict(task=["clotho", "audiocaps", "macs", "wavcaps_freesound"]), form="list", recipe="peaked"),
}


def make_inputs(sc):
    n = [int(s * SR) for s in sc["secs"]]
    wav = synth.synth_waveforms(len(n), max(n), sc["seed0"], lengths=n)
    if sc["form"] == "tensor3":
        return torch.from_numpy(wav)[:, None, :], wav, n
    if sc["form"] == "tensor1":
        return torch.from_numpy(wav[0]), wav, n
    return [torch.from_numpy(wav[i, : n[i]].copy())[None, :] for i in range(len(n))], wav, n


def sub(x: torch.Tensor, maxel: int = 20000) -> np.ndarray:
    """strided sample of a big activation (deterministic)."""
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // maxel)
    return f[::step].numpy().astype(np.float32)


def main() -> None:
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    R = refshim.ref()
    import conette.huggingface.model as hm
    import conette.nn.decoding.beam as beam_mod

    hm.load_audioset_idx_to_name = lambda offline=False, verbose=0: {i: f"tag{i}" for i in range(527)}
    cfg = R.CoNeTTEConfig(**synth.synth_config_dict())
    model = R.CoNeTTEModel(cfg, device="cpu", offline=True)
    loaded = {"recipe": None}

    def load_recipe(recipe):
        if loaded["recipe"] == recipe:
            return
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(recipe=recipe).items()}
        sd["_extra_state_"] = torch.from_numpy(synth.extra_state_tensor())
        print(recipe, model.load_state_dict(sd, strict=True))
        loaded["recipe"] = recipe

    load_recipe("default")
    enc = model.preprocessor.encoder

    # ---- record per-step selections of the reference's own _select_k_next_toks ---------------
    trace = []
    orig_select = beam_mod._select_k_next_toks

    def rec_select(logits_i, prev_sum_lprobs, is_first):
        out = orig_select(logits_i=logits_i, prev_sum_lprobs=prev_sum_lprobs, is_first=is_first)
        k, v = logits_i.shape
        lg = logits_i[0:1] if is_first else logits_i
        cand = torch.log_softmax(lg, dim=1) if is_first else prev_sum_lprobs[:, None] + torch.log_softmax(lg, dim=1)
        top = torch.topk(cand.reshape(-1), min(k + 1, cand.numel())).values
        margin = float(top[k - 1] - top[k]) if top.numel() > k else float("inf")
        trace.append((out[0].tolist(), out[1].tolist(), out[2].tolist(), margin))
        return out

    beam_mod._select_k_next_toks = rec_select

    # ---- stage taps through forward hooks on the reference modules ---------------------------
    taps = {}

    def hook(name):
        def f(mod, inp, out):
            taps[name] = out.detach()
        return f

    enc.bn0.register_forward_hook(lambda m, i, o: taps.__setitem__("logmel", o.detach().transpose(1, 3)))
    enc.downsample_layers[0].register_forward_hook(hook("stem"))
    for i in range(4):
        enc.stages[i].register_forward_hook(hook(f"stage{i}"))
        enc.stages[i][0].register_forward_hook(hook(f"stage{i}_block0"))
    for i in range(1, 4):
        enc.downsample_layers[i].register_forward_hook(hook(f"down{i}"))
    model.model.projection.register_forward_hook(hook("memory"))
    model.preprocessor.register_forward_hook(lambda m, i, o: taps.__setitem__("pre", {k: v.detach() for k, v in o.items()}))

    only = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    for name, sc in SCENARIOS.items():
        if only and name not in only[0]:
            continue
        load_recipe(sc.get("recipe", "default"))
        x, wav, n = make_inputs(sc)
        trace.clear()
        taps.clear()
        with torch.no_grad():
            out = model(x, sr=SR, **sc["kw"])
        rec = dict(
            lengths=np.asarray(n, dtype=np.int64), seed0=np.int64(sc["seed0"]),
            kw=np.asarray(json.dumps(sc["kw"])), form=np.asarray(sc["form"]),
            preds=out["preds"].numpy(), lprobs=out["lprobs"].numpy(),
            mult_preds=out["mult_preds"].numpy(), mult_lprobs=out["mult_lprobs"].numpy(),
            cands=np.asarray(json.dumps(out["cands"])), mult_cands=np.asarray(json.dumps(out["mult_cands"])),
            tasks=np.asarray(json.dumps(out["tasks"])), tags=np.asarray(json.dumps(out["tags"])),
            tags_probs=out["tags_probs"].numpy(),
            frame_embs=taps["pre"]["audio"].numpy(), audio_shape=taps["pre"]["audio_shape"].numpy(),
            memory=taps["memory"].numpy(),
            trace_parent=np.asarray(json.dumps([t[0] for t in trace])),
            trace_token=np.asarray(json.dumps([t[1] for t in trace])),
            trace_sum=np.asarray(json.dumps([t[2] for t in trace])),
            trace_margin=np.asarray([t[3] for t in trace], dtype=np.float64),
        )
        for k in ["logmel", "stem", "stage0_block0", "stage0", "down1", "stage1", "down2", "stage2", "down3", "stage3"]:
            rec["sub_" + k] = sub(taps[k])
            rec["sum_" + k] = np.float64(taps[k].double().sum().item())
            rec["abs_" + k] = np.float64(taps[k].double().abs().sum().item())
        out_dir = GOLD if sc.get("recipe", "default") == "default" else os.path.join(GOLD, sc["recipe"])
        os.makedirs(out_dir, exist_ok=True)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, out["preds"].shape, out["lprobs"].numpy().round(4), "min margin %.4g" % min(t[3] for t in trace))

    if only:
        return
    load_recipe("default")
    # ---- API-shape cases (model.py:185-261, preprocessor.py:89-114) ---------------------------
    api = {}
    wav = torch.from_numpy(synth.synth_waveforms(2, 2 * SR, 7000))
    with torch.no_grad():
        o1 = model(wav[0], sr=SR)                       # rank-1 (T,)
        o2 = model(wav[0][None], sr=SR)                 # rank-2 (C,T)
        o3 = model(wav[0:1, None], sr=SR)               # rank-3 (B,C,T)
        stereo = torch.stack([wav[0], wav[1]], 0)       # (C=2,T) -> channel mean
        o4 = model(stereo, sr=SR)
        o5 = model([wav[0][None], wav[1][None]], sr=[SR, SR], task=["clotho", "audiocaps"])
        pre = model.preprocessor(wav[:, None], SR, None)
        o6 = model(pre["audio"], x_shapes=pre["audio_shape"], preprocess=False, task="clotho")
        o7 = model(wav[:, None], sr=SR, task="clotho")
    api["rank1_preds"] = o1["preds"].tolist()
    api["rank2_preds"] = o2["preds"].tolist()
    api["rank3_preds"] = o3["preds"].tolist()
    api["stereo_preds"] = o4["preds"].tolist()
    api["stereo_lprobs"] = o4["lprobs"].tolist()
    api["list_preds"] = o5["preds"].tolist()
    api["list_tasks"] = o5["tasks"]
    api["list_cands"] = o5["cands"]
    api["nopre_preds"] = o6["preds"].tolist()
    api["nopre_keys"] = sorted(o6.keys())
    api["pre_preds"] = o7["preds"].tolist()
    api["full_keys"] = sorted(o7.keys())
    api["default_task"] = model.default_task
    api["tasks"] = model.tasks
    try:
        model(wav[0], sr=SR, task="not_a_task")
    except ValueError as e:
        api["bad_task_error"] = str(e)
    try:
        model(wav[:, None], sr=SR, task=["clotho"])
    except ValueError as e:
        api["bad_ntasks_error"] = str(e)
    try:
        model(wav[0], sr=16000, x_shapes=torch.tensor([[64000]]))
    except ValueError as e:
        api["xshapes_resample_error"] = str(e)
    # 16 kHz input exercises the resampler stand-in (parity unpinned: restated torchaudio)
    with torch.no_grad():
        o8 = model(wav[0][::2].contiguous(), sr=16000)
    api["sr16k_preds"] = o8["preds"].tolist()
    with open(os.path.join(GOLD, "api_cases.json"), "w") as f:
        json.dump(api, f, indent=1)
    print("api cases written")


if __name__ == "__main__":
    main()

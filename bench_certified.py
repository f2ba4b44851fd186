#!/usr/bin/env python
"""Pipelined benchmark of precision "certified" (round 6): the benchmark workload of bench.py -- B = 64 clips of 10 s per step,
beam 3, inputs resident in HBM, four rotating input batches -- with ids that are the exact precision's (= the reference's) on
every clip.  `bench.py` runs it as a child for `also_pipelined["certified"]` / `["certified@peaked"]`.

Per group of G = 4 steps (256 clips):
  encode stream   the base precision's encoder of every step's batch (as in bench.py's grouped schedule);
  decode stream   ONE base-precision beam search over the group's 256 clips with the device-side margins of conette_decode, the
                  certificate (a few torch ops on (256, 21) floats), the flagged clips compacted to the front of an index vector,
                  their count sent to pinned host memory;
  exact stream    ONE group later (the host reads the count of group g after it has enqueued group g + 1: the device never
                  waits for the host): the flagged clips' waveforms gathered, exact encoder + exact beam search over
                  ceil(n / 16) * 16 clips, results scattered over the group's table.  No host round trip per search step.
The timed windows end with a full drain, so every step's re-run lies inside its window.  After the timed region every timed step's
ids and scores are compared with the un-pipelined `Engine.generate_certified` of its batch (bit for bit) and its ids with the
exact precision's.

    python bench_certified.py [--base mixed16|f16|bf16] [--checkpoint default|peaked] [--steps 100] [--repeat 3] [--beam 3]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SR, CLIP_S = 32000, 10


def main(argv=None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default="mixed16", help="base precision: mixed16 (fp16 encoder + exact decoder; engine.CERT_DEFAULT_BASE), f16, bf16, bf16+f16dec")
    ap.add_argument("--checkpoint", default="default", choices=["default", "peaked"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--policy", default="strict", choices=["strict", "best"],
                    help="strict: ids, candidates and their slot order certified; best: the returned caption and the SET of beam hypotheses "
                         "(precision 'certified-best': the pick-order margins are not held to the tolerance)")
    args = ap.parse_args(argv)

    import numpy as np
    import torch
    import conette_amd  # noqa: F401
    from conette_amd import synth
    from conette_amd.engine import CERT_TOL, Engine, cert_kind

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, beam, min_pred, max_pred, G = args.batch, args.beam, 3, 20, args.group
    if args.steps % G:
        raise SystemExit(f"--steps {args.steps} must be a multiple of the decode group {G}")
    NB = 4
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(recipe=args.checkpoint).items()}
    eng = Engine(sd, precision=("certified" if args.policy == "strict" else "certified-best") + f":{args.base}", device=dev)
    eng.set_encode_reserved_cus(24)
    forbid = sd["model.forbid_rep_mask"].to(dev)
    L = CLIP_S * SR
    t_audio = eng.lib.conette_num_audio_frames(L)
    waves = [torch.from_numpy(synth.synth_waveforms(B, L, 1234 + k * B)).to(dev) for k in range(NB)]
    wave_cat = torch.cat(waves)                                  # (NB B, L): what the exact stream gathers flagged clips from
    GB = G * B
    lens_big = torch.full((GB,), t_audio, dtype=torch.int32, device=dev)
    bos_big = sd["model.task_id_to_token_id"][torch.zeros(GB, dtype=torch.long)].to(torch.int32).to(dev)   # task "clotho"
    s_enc, s_dec, s_x = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
    NS = 3
    slots = []
    for sl in range(NS):
        slots.append(dict(
            fe=eng.decode_input_buffer(GB, t_audio, beam, max_pred, slot=100 + sl, margins=True),
            clip=torch.empty((B, 527), dtype=torch.float32, device=dev),
            ids=torch.zeros((GB, max_pred), dtype=torch.int32, device=dev), lp=torch.zeros((GB,), dtype=torch.float32, device=dev),
            flag=torch.zeros((GB,), dtype=torch.bool, device=dev), order=torch.zeros((GB,), dtype=torch.long, device=dev),
            n_host=torch.zeros((1,), dtype=torch.int64).pin_memory(),
            enc_done=[torch.cuda.Event() for _ in range(G)], dec_done=torch.cuda.Event(), x_done=torch.cuda.Event()))
    state = {"i": 0, "pending": [], "keep": None, "flagged": 0, "clips": 0, "rerun_clips": 0}
    # the exact stream's scratch at its largest shape up front (a growing workspace would be another hipGraph key every time)
    eng._workspace("xenc", eng.lib.conette_encode_workspace_bytes(eng._ctx_x, GB, L))
    eng._workspace("xdec200", eng.lib.conette_decode_workspace_bytes(eng._ctx_x, GB, t_audio, beam, max_pred))

    def rerun(g):
        """the exact stream's share of group g: re-run the clips the certificate flagged, merge, keep"""
        gs = slots[g % NS]
        gs["dec_done"].synchronize()          # (group g + 1 is enqueued already: the device has work while the host waits)
        n = int(gs["n_host"][0])
        state["flagged"] += n
        state["clips"] += GB
        with torch.cuda.stream(s_x):
            s_x.wait_event(gs["dec_done"])
            if n > 0:
                n_pad = min(GB, (n + 15) // 16 * 16)     # few distinct shapes: the exact search replays a hipGraph per shape
                state["rerun_clips"] += n_pad
                idx = gs["order"][:n_pad]
                src = ((g * G + idx // B) % NB) * B + idx % B        # row of wave_cat: step i = g G + idx // B encoded batch i % NB
                fe_x, _ = eng.encode(wave_cat.index_select(0, src), exact=True)
                rx = eng.decode(fe_x, lens_big[:n_pad], bos_big[:n_pad], forbid, beam, min_pred, max_pred, clone=False, slot=200,
                                exact=True)
                take = gs["flag"].index_select(0, idx)                # (the padding rows keep their certified 16-bit results)
                gs["ids"].index_copy_(0, idx, torch.where(take[:, None], rx["best_preds"], gs["ids"].index_select(0, idx)))
                gs["lp"].index_copy_(0, idx, torch.where(take, rx["best_lprobs"], gs["lp"].index_select(0, idx)))
            if state["keep"] is not None and g < state["keep"][0].shape[0]:
                kp, kl, kf = state["keep"]
                kp[g].copy_(gs["ids"], non_blocking=True)
                kl[g].copy_(gs["lp"], non_blocking=True)
                kf[g].copy_(gs["flag"], non_blocking=True)
            gs["x_done"].record(s_x)

    def step():
        i = state["i"]
        g, m = i // G, i % G
        gs = slots[g % NS]
        with torch.cuda.stream(s_enc):
            if g >= NS and m == 0:
                s_enc.wait_event(gs["x_done"])      # the slot's tables and frame buffer are free again
            eng.encode(waves[i % NB], out=(gs["fe"][m * B:(m + 1) * B], gs["clip"]), slot=i & 1)
            gs["enc_done"][m].record(s_enc)
        if m == G - 1:
            with torch.cuda.stream(s_dec):
                for e_ in gs["enc_done"]:
                    s_dec.wait_event(e_)
                if g >= NS:
                    s_dec.wait_event(gs["x_done"])
                out = eng.decode(gs["fe"], lens_big, bos_big, forbid, beam, min_pred, max_pred, clone=False, slot=100 + g % NS,
                                 want_margins=True)
                flag = eng.uncertified(out["margins"], out["best_lprobs"], beam=beam)
                gs["flag"].copy_(flag)
                gs["order"].copy_(torch.argsort(flag.to(torch.int8), descending=True, stable=True))
                gs["n_host"].copy_(flag.sum().reshape(1), non_blocking=True)
                gs["ids"].copy_(out["best_preds"])
                gs["lp"].copy_(out["best_lprobs"])
                gs["dec_done"].record(s_dec)
            state["pending"].append(g)
            while state["pending"] and state["pending"][0] < g:      # one group behind
                rerun(state["pending"].pop(0))
        state["i"] = i + 1

    def drain():
        while state["pending"]:
            rerun(state["pending"].pop(0))
        torch.cuda.synchronize(dev)

    warm = max(args.warmup, 3 * NS * G) // G * G + G
    for _ in range(warm):
        step()
    drain()
    n_rep = max(1, args.repeat)
    n_groups = n_rep * args.steps // G
    # (the group index keeps counting through warm-up and timed region: keep tables are indexed from the first timed group)
    g0 = state["i"] // G
    kp = torch.zeros((n_groups, GB, max_pred), dtype=torch.int32, device=dev)
    kl = torch.zeros((n_groups, GB), dtype=torch.float32, device=dev)
    kf = torch.zeros((n_groups, GB), dtype=torch.bool, device=dev)
    # rerun() indexes the keep tables by absolute group number: g0 unused leading rows
    keep_abs = (torch.cat([torch.zeros((g0, GB, max_pred), dtype=kp.dtype, device=dev), kp]),
                torch.cat([torch.zeros((g0, GB), dtype=kl.dtype, device=dev), kl]),
                torch.cat([torch.zeros((g0, GB), dtype=kf.dtype, device=dev), kf]))
    state["keep"] = keep_abs
    state["flagged"] = state["clips"] = state["rerun_clips"] = 0
    win_dt = []
    for _ in range(n_rep):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        win_dt.append(time.perf_counter() - t0)
    kp, kl, kf = (t[g0:] for t in keep_abs)
    recompute_fraction = state["flagged"] / max(1, state["clips"])
    rerun_fraction = state["rerun_clips"] / max(1, state["clips"])

    # ---- checks (untimed): every timed step against the un-pipelined certified search of its batch, and against the exact precision
    solo_p, solo_l, solo_f, ex_p = [], [], [], []
    lens_b, bos_b = lens_big[:B], bos_big[:B]
    for k in range(NB):
        fe_k, _ = eng.encode(waves[k])
        o = eng.generate_certified(waves[k], fe_k.clone(), lens_b, bos_b, forbid, beam, min_pred, max_pred)
        solo_p.append(o["best_preds"].clone()), solo_l.append(o["best_lprobs"].clone()), solo_f.append(o["recomputed"].clone())
        fx, _ = eng.encode(waves[k], exact=True)
        ex_p.append(eng.decode(fx, lens_b, bos_b, forbid, beam, min_pred, max_pred, exact=True)["best_preds"].clone())
    solo_p, solo_l, solo_f, ex_p = torch.stack(solo_p), torch.stack(solo_l), torch.stack(solo_f), torch.stack(ex_p)
    steps_all = torch.arange(n_groups * G, device=dev) + g0 * G
    rot = steps_all % NB
    kp_s, kl_s, kf_s = kp.reshape(n_groups * G, B, max_pred), kl.reshape(n_groups * G, B), kf.reshape(n_groups * G, B)
    same_solo = bool(torch.equal(kp_s, solo_p[rot]) and torch.equal(kl_s, solo_l[rot]) and torch.equal(kf_s, solo_f[rot]))
    same_exact_clips = int((kp_s == ex_p[rot]).all(dim=2).sum())      # (best_preds: certified under both policies)
    torch.cuda.synchronize(dev)

    order = sorted(range(n_rep), key=lambda k: win_dt[k])
    dt = win_dt[order[(n_rep - 1) // 2]]
    res = {
        "metric": "clips_per_sec", "value": round(B * args.steps / dt, 2), "unit": "clips/s", "n_gpus": 1, "steps": args.steps,
        "warmup": warm, "repeat": n_rep, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "windows": {"clips_per_sec": [round(B * args.steps / w, 2) for w in win_dt]},
        "precision": eng.precision_name, "policy": args.policy, "dtype": f"{args.base} + f16x2 re-run of uncertified clips", "checkpoint": args.checkpoint,
        "recompute_fraction": round(recompute_fraction, 4), "rerun_fraction_with_padding": round(rerun_fraction, 4),
        "tolerance": {"a_b_c": list(CERT_TOL[args.base][cert_kind(beam)]),
                      "source": "profiles/r06_margin_calibration.txt"},
        "pipeline_consistent": same_solo, "pipeline_steps_checked": int(n_groups * G),
        "ids_identical_to_exact": [same_exact_clips, int(n_groups * G * B)],
        "config": {"workload": f"B={B}/GPU x 10 s @ 32 kHz clips, beam {beam}, min 3 / max 20 tokens, synthetic {args.checkpoint} checkpoint",
                   "decode_group": G, "input_batches_rotated": NB, "exact_rerun_lag_groups": 1},
        "higher_is_better": True, "data": "synthetic",
    }
    print(json.dumps(res), flush=True)
    if not same_solo or same_exact_clips != n_groups * G * B:
        raise SystemExit("bench_certified: pipelined certified results differ from the un-pipelined ones or from the exact precision's ids")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark of the CoNeTTE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clips already resident in
HBM: waveform (B, 320000) fp32 -> log-mel -> ConvNeXt -> projection -> KV-cached decoder under
beam search -> token ids / scores on device (+ the RCCL all-gather of ids for N > 1).
Workload = BASELINE.json configs[1]/[2] shape: B = 64 clips of 10 s @ 32 kHz per GPU, beam 3,
bf16 operands, seeded synthetic checkpoint (random-init weights of the reference architecture;
the published checkpoint is not reachable offline).  N > 1: one process per GPU, every rank
processes its own 64 clips (weak scaling), no data-path collective except the final all-gather.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import conette_amd  # noqa: E402,F401
from conette_amd import synth  # noqa: E402

SR = 32000
CLIP_S = 10
DIMS = (96, 192, 384, 768)
DEPTHS = (3, 3, 9, 3)
POS = (252 * 56, 126 * 28, 63 * 14, 31 * 7)  # positions per 10 s clip and stage
PEAK_BF16_TFLOPS = 2500.0                    # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                        # HBM3E spec (MI355X_MICROARCH.md)


# C = 96 / 192 / 384: pw1 + GELU + pw2 run as ONE kernel (mlp_fused.h), timed under "pw1_gemm"
FUSED_STAGES = (0, 1) if os.environ.get("CN_MLP384") == "0" else (0, 1, 2)
# kernel-name fragments of each profiling class in the committed PMC table (profiles/*_pmc_hbm_traffic.csv)
PMC_KERNELS = {
    "pw1_gemm": ("cn_mlp_fused_kernel", "EpiBiasActIDF16bLi4"),
    "pw2_gemm": ("EpiResid",),
    "dwconv_ln": ("cn_dwconv_ln_kernel",),
}


def pmc_traffic(cls: str, batch: int, launches_per_step: float):
    """HBM bytes per launch of a kernel class from the newest committed rocprofv3 PMC table (separate FETCH_SIZE and
    WRITE_SIZE passes of this benchmark at B = 64, bf16; FETCH_SIZE doubled: gfx950 tallies wide streaming reads at
    half their size, MI355X_MICROARCH.md).  None when no table matches the configuration."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.csv")))
    if not files or batch != 64:
        return None, None
    tot = 0.0
    n_disp = {}
    for r in csv.DictReader(open(files[-1])):
        keys = list(r.keys())
        name, disp, kb = r[keys[1]], int(r[keys[3]]), float(r[keys[4]])
        if not any(f in name for f in PMC_KERNELS.get(cls, ())):
            continue
        n_disp.setdefault(r[keys[0]], 0)
        n_disp[r[keys[0]]] += disp
        tot += (2.0 if r[keys[0]] == "FETCH_SIZE" else 1.0) * kb * 1024.0 * disp
    if not n_disp:
        return None, None
    passes = max(n_disp.values()) / max(launches_per_step, 1.0)  # profiled batches in the table
    return tot / passes / launches_per_step, os.path.basename(files[-1])


def algorithmic_work(cls: str, batch: int):
    """(flops, bytes) of ALL launches of a kernel class for one batch.

    Classes follow the library's profiling scopes: "pw1_gemm" = the fused MLP launches of stages 0-1
    (both GEMMs) + the pw1 GEMMs of stages 2-3; "pw2_gemm" = the pw2 GEMMs of stages 2-3."""
    fl = by = 0.0
    for st, (c, d, p) in enumerate(zip(DIMS, DEPTHS, POS)):
        n = batch * p
        gemm = 2.0 * n * c * 4 * c
        wbytes = 2.0 * 4 * c * c
        if cls == "pw1_gemm":
            if st in FUSED_STAGES:   # y (bf16) in, x (fp32) in + out, both weight matrices; hidden stays on chip
                fl += d * 2 * gemm
                by += d * (2.0 * n * c + 8.0 * n * c + 2 * wbytes)
            else:                    # (P x C) . (C x 4C) + bias + GELU -> bf16 hidden
                fl += d * gemm
                by += d * (2.0 * n * c + 2.0 * n * 4 * c + wbytes)
        elif cls == "pw2_gemm" and st not in FUSED_STAGES:  # (P x 4C) . (4C x C), LayerScale, fp32 residual in/out
            fl += d * gemm
            by += d * (2.0 * n * 4 * c + 8.0 * n * c + wbytes)
        elif cls == "dwconv_ln":     # 49 MAC per element on the VALU; fp32 in, bf16 out
            fl += d * 2.0 * 49 * n * c
            by += d * (4.0 * n * c + 2.0 * n * c)
    return fl, by


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-clips", type=int, default=32, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from conette_amd.dist import gather_captions
    from conette_amd.engine import Engine

    B, beam, min_pred, max_pred = args.batch, args.beam, 3, 20
    L = CLIP_S * SR
    sd_np = synth.synth_state_dict()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    eng = Engine(sd, precision=args.precision, device=dev)
    if os.environ.get("CN_NO_GRAPH"):
        eng.set_decode_graph(False)
    eng.set_encode_reserved_cus(int(os.environ.get("CN_ENC_RESERVE", "48")))
    wave = torch.from_numpy(synth.synth_waveforms(B, L, 1234 + rank * B)).to(dev)
    t_audio = eng.lib.conette_num_audio_frames(L)
    lens = torch.full((B,), t_audio, dtype=torch.int32, device=dev)
    bos = sd["model.task_id_to_token_id"][torch.zeros(B, dtype=torch.long)].to(dev)  # task "clotho"
    forbid = sd["model.forbid_rep_mask"].to(dev)
    # Two pipeline slots and two HIP streams: the decode of batch i (small latency-bound launches,
    # replayed from a hipGraph) overlaps the encode of batch i+1 (big MFMA / VALU kernels).
    share = int(os.environ.get("CN_DEC_SHARE", "0"))  # CU-masked streams measured no gain on this stack
    if share > 0:   # CU-partitioned streams: decode owns 1/share of the CUs, encode the rest
        from conette_amd.engine import make_partitioned_streams
        s_enc, s_dec = make_partitioned_streams(dev, decode_share=share)
    else:
        prio = int(os.environ.get("CN_DEC_PRIO", "-1"))  # decode stream priority (negative = higher)
        s_enc, s_dec = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=prio)
    fe_buf = [eng.decode_input_buffer(B, t_audio, beam, max_pred, slot=i) for i in range(2)]
    clip_buf = [torch.empty((B, 527), dtype=torch.float32, device=dev) for _ in range(2)]
    enc_done = [torch.cuda.Event() for _ in range(2)]
    dec_done = [torch.cuda.Event() for _ in range(2)]
    state = {"i": 0, "last": None}

    def step():
        i = state["i"]
        sl = i & 1
        with torch.cuda.stream(s_enc):
            if i >= 2:
                s_enc.wait_event(dec_done[sl])          # slot's frame buffer is free again
            eng.encode(wave, out=(fe_buf[sl], clip_buf[sl]))
            enc_done[sl].record(s_enc)
        with torch.cuda.stream(s_dec):
            s_dec.wait_event(enc_done[sl])
            out = eng.decode(fe_buf[sl], lens, bos, forbid, beam, min_pred, max_pred, clone=False, slot=sl)
            res = (out["best_preds"], out["best_lprobs"])
            if world > 1:
                res = gather_captions(res[0], res[1], world * B)
            dec_done[sl].record(s_dec)
        state["i"] = i + 1
        state["last"] = out
        return res

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 6)):  # >= 3 per slot: the third identical decode call replays its hipGraph
        step()
    fence()

    # ---- pre-pass (untimed): which kernel class dominates, encode / decode split --------------------
    enc_classes = ("frontend", "stem", "dwconv_ln", "pw1_gemm", "pw2_gemm", "downsample", "heads")
    eng.profile_enable(enc_classes)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    eng.encode(wave, out=(fe_buf[0], clip_buf[0]))
    ev[1].record()
    out = eng.decode(fe_buf[0], lens, bos, forbid, beam, min_pred, max_pred, clone=False, slot=0)
    ev[2].record()
    torch.cuda.synchronize(dev)
    pre = eng.profile_read()
    eng.profile_enable(())
    encode_ms, decode_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    stage_ms = {k: round(v[0], 4) for k, v in pre.items()}
    dominant = max(("pw1_gemm", "pw2_gemm", "dwconv_ln"), key=lambda k: pre.get(k, (0.0, 0))[0])
    mult = out["mult_preds"]
    gen_tokens = int((mult != 0).sum().item())  # generated row-steps of this batch (EOS included, pad excluded)

    # ---- timed region: exactly K steps, events only around the dominant class --------------------------
    eng.profile_enable((dominant,))
    state["i"] = 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    dom_ms, dom_n = prof[dominant]
    fl, by = algorithmic_work(dominant, B)
    launches_per_step = dom_n / args.steps
    avg_launch_s = dom_ms * 1e-3 / dom_n
    if dominant in ("pw1_gemm", "pw2_gemm"):
        achieved = fl / launches_per_step / avg_launch_s / 1e12
        roof = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None}
    else:
        achieved = by / launches_per_step / avg_launch_s / 1e9
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
    roof["avg_launch_us"] = round(avg_launch_s * 1e6, 2)
    roof["launches_per_step"] = launches_per_step
    if args.precision == "bf16":
        tr, src = pmc_traffic(dominant, B, launches_per_step)
        if tr is not None:
            roof["traffic"] = round(tr)            # HBM bytes per launch (class average), PMC
            roof["traffic_unit"] = "bytes/launch"
            roof["traffic_source"] = "profiles/" + src
            roof["algorithmic_bytes_per_launch"] = round(by / launches_per_step)

    result = None
    if rank == 0:
        clips_per_s = world * B * args.steps / dt
        result = {
            "metric": "clips_per_sec", "value": round(clips_per_s, 2), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"B={B}/GPU x 10 s @ 32 kHz clips, ConvNeXt encoder + beam-{beam} KV-cached decode "
                                   f"(min 3 / max 20 tokens, V=5631), synthetic seeded checkpoint",
                       "batch_per_gpu": B, "global_batch": world * B, "beam_size": beam, "parallelism": f"dp{world}"},
            "decode_tokens_per_sec": round(world * gen_tokens / (decode_ms * 1e-3), 1),
            "encode_ms": round(encode_ms, 3), "decode_ms": round(decode_ms, 3), "stage_ms": stage_ms,
            "roofline": roof,
        }

    # ---- CPU baseline (rank 0, N = 1 only): the oracle restatement on host cores ------------------------
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        from oracle import cpu_ref as O

        n_thr = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n_thr)
        cfg = synth.synth_config_dict()
        xs = wave[: args.cpu_clips].cpu()[:, None, :]
        with torch.no_grad():
            tc = time.perf_counter()
            O.model_forward(sd, cfg, xs, sr=SR, task="clotho", beam_size=beam)
            tc = time.perf_counter() - tc
        result["cpu_baseline"] = {
            "value": round(args.cpu_clips / tc, 4), "unit": "clips/s", "cores": n_thr, "kind": "port",
            "sample": f"{args.cpu_clips} of the same synthetic clips, same checkpoint, beam {beam}: oracle/cpu_ref.py "
                      f"(stock PyTorch fp32, reference algorithm incl. its no-KV-cache decode), {tc:.1f} s"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

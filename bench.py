#!/usr/bin/env python
"""Benchmark of the CoNeTTE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--global-batch G] [--workload mixed]

One "step" = one pass of the hot path over one batch of synthetic clips already resident in
HBM: waveform (B, 320000) fp32 -> log-mel -> ConvNeXt -> projection -> KV-cached decoder under
beam search -> token ids / scores on device (+ the RCCL all-gather of ids for N > 1).
Default workload = BASELINE.json configs[1]/[2] shape: B = 64 clips of 10 s @ 32 kHz per GPU, beam 3,
bf16 operands, seeded synthetic checkpoint (random-init weights of the reference architecture;
the published checkpoint is not reachable offline).

N > 1: one process per GPU.  Started by the driver through torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or directly as `python bench.py --gpus N`, in which case this process
starts the N ranks itself (a child `torch.distributed.run`, before anything here touches a GPU) and
exits with its status.  Default N > 1 mode: weak scaling, every rank processes its own 64 clips, no
data-path collective except the final all-gather of ids + scores.  `--global-batch G` (BASELINE config 4:
G = 2048 over 8 GPUs): strong scaling, rank r takes the contiguous shard `shard_bounds(G, r, N)`.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 32000
CLIP_S = 10
DIMS = (96, 192, 384, 768)
DEPTHS = (3, 3, 9, 3)
POS = (252 * 56, 126 * 28, 63 * 14, 31 * 7)  # positions per 10 s clip and stage
PEAK_BF16_TFLOPS = 2500.0                    # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                        # HBM3E spec (MI355X_MICROARCH.md)
DEC_GROUP_DEFAULT = 16                       # CN_DEC_GROUP: batches whose beam searches run as one chain (see step_grouped)
FUSED_STAGES = (0, 1, 2)                     # pw1 + GELU + pw2 run as ONE kernel (mlp_rc2.h), timed under "pw1_gemm"
# kernel-name fragments of each profiling class in the committed PMC tables (profiles/*_pmc_*.csv)
PMC_KERNELS = {
    "pw1_gemm": ("cn_mlp_rc2_", "cn_mlp_rs_", "cn_mlp_rs16_", "EpiBiasActIDF16bLi4"),
    "pw2_gemm": ("EpiResid",),
    "dwconv_ln": ("cn_dwconv_ln",),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total clips, sharded over the GPUs (strong scaling)")
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "bf16+f16dec", "mixed", "mixed16", "fp32", "exact"])
    ap.add_argument("--repeat", type=int, default=5,
                    help="timed windows of --steps steps each (barrier + synchronize around every window); `value` and "
                         "`ms_per_step` are the MEDIAN window's, every window's clips/s is listed under `windows`")
    ap.add_argument("--workload", default="fixed", choices=["fixed", "mixed"],
                    help="fixed: 10 s clips; mixed: lengths U[1 s, 30 s] (seed 1234), length-bucketed batches")
    ap.add_argument("--cpu-clips", type=int, default=32, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--parity-clips", type=int, default=64, help="clips in the bf16-vs-fp32 agreement leg (0 = skip)")
    ap.add_argument("--parity-batches", type=int, default=4,
                    help="batches of --parity-clips clips (other seeds) in the agreement-with-the-fp32-mode leg: an identical-caption rate "
                         "moves by several clips when a kernel's fp32 rounding changes in the last bit, 64 clips are a noisy sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: CPU self-test of the launcher / sharding / gather path (no GPU work, no timing)")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--also", default="bf16:e2,bf16:g1,bf16+f16dec,f16,mixed16,exact,certified,certified@peaked,certified-best@peaked,certified:f16@peaked",
                    help="N = 1, fixed workload, bf16 only: after the run, the SAME pipelined benchmark at these precisions (one child "
                         "process each, 3 windows, no CPU / parity legs), reported under `also_pipelined` ('' = skip); `PREC:g1` = that "
                         "precision with one beam search per batch (CN_DEC_GROUP=1) instead of the grouped decode")
    return ap.parse_args(argv)


def choose_decode_group(steps: int, batch: int, beam: int, cap: int = 16) -> int:
    """Batches whose beam searches run as one chain (bench.py step_grouped): the largest divisor of ``steps`` -- a timed window must
    end on a group boundary -- with G x batch <= 256 clips, G <= cap and G x batch x beam below the 4 096 rows at which the
    GEMM tile shapes change."""
    g_max = max(1, min(cap, int(os.environ.get("CN_DEC_GROUP_CLIPS", "256")) // max(1, batch), 4095 // max(1, batch * beam)))
    return max(g for g in range(1, g_max + 1) if steps % g == 0)


def launch_ranks(args) -> int:
    """Start `--gpus N` ranks (one process per GPU) as a child torch.distributed.run and return its exit status.
    Runs BEFORE this process imports torch.cuda or the HIP library: a process that has initialised the GPU must not
    start another GPU program in its place, and children inherit nothing from it."""
    import socket
    port = args.master_port
    if port == 0:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pmc_table(suffix: str):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{suffix}.csv")))
    return files[-1] if files else None


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_conette_build", os.path.join(ROOT, "conette-audio-captioning_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pmc_is_current(path: str):
    """True / False: the PMC table was measured on a build of the sources the running library was built from
    (tools/profile_round.sh writes build.py's source hash -- csrc/ + header + flags -- next to the tables, build.py writes the
    same hash next to the library; a binary hash would turn false for anyone who recompiles: VERDICT r03).  None: no record
    (tables of rounds 1-3 carry a binary hash only)."""
    meta = path.rsplit("_pmc_", 1)[0] + "_pmc_meta.json"
    if not os.path.exists(meta):
        return None
    try:
        want = json.load(open(meta)).get("source_sha256")
        have = _build_module().built_source_hash()
    except (OSError, ValueError):
        return None
    if want is None or have is None:
        return None
    return want == have


def pmc_traffic(cls: str, batch: int, launches_per_step: float):
    """HBM bytes per launch of a kernel class from the newest committed rocprofv3 PMC table (separate FETCH_SIZE and
    WRITE_SIZE passes of this benchmark at B = 64, bf16; FETCH_SIZE doubled: gfx950 tallies wide streaming reads at
    half their size, MI355X_MICROARCH.md).  None when no table matches the configuration."""
    import csv
    path = pmc_table("pmc_hbm_traffic")
    if path is None or batch != 64:
        return None, None
    tot = 0.0
    n_disp = {}
    for r in csv.DictReader(open(path)):
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
    return tot / passes / launches_per_step, os.path.basename(path)


def pmc_mfma_busy(cls: str):
    """MFMA-busy share of the class's kernels from the newest committed profiles/*_pmc_mfma.csv
    (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), tools/pmc_mfma_summary.py); None when absent."""
    import csv
    path = pmc_table("pmc_mfma")
    if path is None:
        return None, None
    num = den = 0.0
    for r in csv.DictReader(open(path)):
        if any(f in r["kernel"] for f in PMC_KERNELS.get(cls, ())):
            w = float(r["dispatches"]) * float(r["avg_us"]) if "avg_us" in r and r["avg_us"] else float(r["dispatches"])
            num += w * float(r["mfma_busy"])
            den += w
    return (round(num / den, 4), os.path.basename(path)) if den > 0 else (None, None)


def algorithmic_work(cls: str, batch: int, xbytes: float = 2.0):
    """(flops, bytes) of ALL launches of a kernel class for one batch of 10 s clips.

    Classes follow the library's profiling scopes: "pw1_gemm" = the fused MLP launches of stages 0-2
    (both GEMMs) + the pw1 GEMMs of stage 3; "pw2_gemm" = the pw2 GEMMs of stage 3.  xbytes: bytes per element of the
    residual stream (2: the fp16 stream of the 16-bit precisions since round 5; 4: fp32)."""
    fl = by = 0.0
    for st, (c, d, p) in enumerate(zip(DIMS, DEPTHS, POS)):
        n = batch * p
        gemm = 2.0 * n * c * 4 * c
        wbytes = 2.0 * 4 * c * c
        if cls == "pw1_gemm":
            if st in FUSED_STAGES:   # y (16-bit) in, x in + out, both weight matrices; hidden stays on chip
                fl += d * 2 * gemm
                by += d * (2.0 * n * c + 2 * xbytes * n * c + 2 * wbytes)
            else:                    # (P x C) . (C x 4C) + bias + GELU -> bf16 hidden
                fl += d * gemm
                by += d * (2.0 * n * c + 2.0 * n * 4 * c + wbytes)
        elif cls == "pw2_gemm" and st not in FUSED_STAGES:  # (P x 4C) . (4C x C), LayerScale, residual in/out
            fl += d * gemm
            by += d * (2.0 * n * 4 * c + 2 * xbytes * n * c + wbytes)
        elif cls == "dwconv_ln":     # 49 MAC per element on the VALU; residual stream in, 16-bit y out
            fl += d * 2.0 * 49 * n * c
            by += d * (xbytes * n * c + 2.0 * n * c)
    return fl, by


def gloo_selftest(args, rank: int, world: int) -> None:
    """CPU check of everything around the GPU work in the N > 1 path: rendezvous, shard bounds, the all-gather of ids +
    scores, trimming, the one JSON line on rank 0 (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist
    import conette_amd  # noqa: F401
    from conette_amd.dist import gather_caption_windows, gather_captions, shard_bounds, trim_captions
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = args.global_batch if args.global_batch > 0 else world * args.batch
    lo, hi = shard_bounds(total, rank, world) if args.global_batch > 0 else (rank * args.batch, (rank + 1) * args.batch)
    g = torch.Generator().manual_seed(11)
    full = torch.randint(4, 100, (total, 20), generator=g, dtype=torch.int32)
    full[:, 7] = 2
    full[:, 8:] = 0
    scores = -torch.arange(total, dtype=torch.float32)
    p, l = gather_captions(full[lo:hi].clone(), scores[lo:hi].clone(), total)
    ok = torch.equal(p, full) and torch.equal(l, scores) and trim_captions(p).shape[1] == 8
    # the per-window form the timed loop uses: K steps' tables in one collective
    fk = torch.stack([full + k for k in range(3)])
    sk = torch.stack([scores - k for k in range(3)])
    pk, lk = gather_caption_windows(fk[:, lo:hi].clone(), sk[:, lo:hi].clone(), total)
    ok = ok and torch.equal(pk, fk) and torch.equal(lk, sk)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "clips_per_sec", "value": None, "unit": "clips/s", "n_gpus": world,
                          "world_size_observed": dist.get_world_size(), "selftest": "gloo", "ok": bool(flag.item()),
                          "scaling": "strong" if args.global_batch > 0 else "weak", "global_batch": total,
                          "shard": [lo, hi]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def agreement(a, b, eos: int = 2):
    """(share of identical sequences, share of identical tokens up to and including the longer <eos>) of two (B, P) id matrices."""
    import torch
    w = max(a.shape[1], b.shape[1])
    pa = torch.zeros((a.shape[0], w), dtype=torch.long)
    pb = torch.zeros((b.shape[0], w), dtype=torch.long)
    pa[:, : a.shape[1]] = a.cpu().long()
    pb[:, : b.shape[1]] = b.cpu().long()
    same_seq = float((pa == pb).all(dim=1).float().mean())
    live = (pa != 0) | (pb != 0)
    same_tok = float(((pa == pb) & live).sum() / max(int(live.sum()), 1))
    return round(same_seq, 4), round(same_tok, 4)


def decode_algorithmic_bytes(batch: int, beam: int, t_audio: int, n_steps: int, vocab: int, d: int = 256, n_layers: int = 6,
                             d_ff: int = 2048, feat: int = 768):
    """HBM bytes one beam search has to move if every operand is touched once per use (DESIGN.md section 4, bf16 operands):
    preparation = frame_embs in, projection + cross K/V of all layers out; per step: every weight matrix once (the six
    attention-side d x d matrices + the FFN pair per layer, the classifier), the clip's cross K/V per layer (shared by its
    beams), the self-attention cache rows written so far, the (rows, vocab) fp32 logits written and read once."""
    rows = batch * beam
    w_layer = (6 * d * d + 2 * d * d_ff) * 2
    w_step = n_layers * w_layer + vocab * d * 2
    prep = batch * t_audio * feat * 4 + (feat * d + n_layers * 2 * d * d) * 2 + batch * t_audio * (d + n_layers * 2 * d) * 2
    total = prep
    for t in range(n_steps):
        r = batch if t == 0 else rows                                  # step 0 runs one row per clip
        total += w_step + n_layers * batch * t_audio * 2 * d * 2       # weights + cross K/V
        total += n_layers * r * (t + 1) * 2 * d * 2                    # self K/V cache read (and one row appended)
        total += 2 * r * vocab * 4                                     # logits out + in
    return total


def clip_min_margins(trace, n_clips: int):
    """Smallest top-k margin (candidate k vs candidate k + 1) over all search steps of every clip, from the oracle's
    per-call trace (oracle/cpu_ref.py generate(trace=...)): how close the clip's closest decision was."""
    m = [float("inf")] * n_clips
    for step in trace:
        for call in step:
            m[call["clip"]] = min(m[call["clip"]], call["margin"])
    return m


def _ids(out):
    return out["best_preds"][:, : int(out["sizes"][1].item())].cpu()


def compare_ids(got, got_lp, ref, ref_lp=None, margins=None):
    """Agreement of (n, P) id matrices with the checker's: identical sequences, identical tokens, |score diff| over the
    clips whose ids agree, and -- when the checker's decision margins are known -- the same over the clips whose closest
    decision was wider than 0.05 / 0.25 (a precision can only be blamed for flips of decisions that were not ties)."""
    import torch
    n = min(got.shape[0], ref.shape[0])
    got, ref = got[:n], ref[:n]
    seq, tok = agreement(got, ref)
    w = max(got.shape[1], ref.shape[1])
    pa = torch.zeros((n, w), dtype=torch.long)
    pb = torch.zeros((n, w), dtype=torch.long)
    pa[:, : got.shape[1]] = got.long()
    pb[:, : ref.shape[1]] = ref.long()
    same = (pa == pb).all(dim=1)
    r = {"clips": n, "seq_identical": seq, "token_agree": tok}
    if ref_lp is not None and got_lp is not None:
        d = (got_lp[:n].cpu().float() - ref_lp[:n].cpu().float()).abs()
        r["max_abs_lprob_diff_same_ids"] = round(float(d[same].max()), 6) if bool(same.any()) else None
        r["max_abs_lprob_diff"] = round(float(d.max()), 6)
    if margins is not None:
        mg = torch.tensor(margins[:n])
        for thr in (0.05, 0.25):
            sel = mg > thr
            r[f"seq_identical_margin_gt_{thr}"] = [int((same & sel).sum()), int(sel.sum())]
    return r


def also_pipelined(args, batch):
    """The same encode | decode | decode pipeline at other precisions, each in a child process of its own (this process stays
    idle meanwhile; a child is a plain `python bench.py --precision P`, started, never exec'ed into).  `value` of the line stays
    the bf16 number BASELINE.json's config names; these are the rates of the precisions whose ids are closer to the oracle's
    (`parity.vs_oracle` of the same line)."""
    import subprocess
    out = {}
    for name in [n for n in args.also.split(",") if n]:
        if name.startswith("certified"):       # "certified[-best][:base][@peaked]": the id-certified pipeline (bench_certified.py, round 6)
            spec, _, ckpt = name.partition("@")
            cmd = [sys.executable, os.path.join(ROOT, "bench_certified.py"), "--base", spec.partition(":")[2] or "mixed16", "--checkpoint",
                   ckpt or "default", "--policy", "best" if spec.startswith("certified-best") else "strict", "--steps", str(max(4, args.steps // 4 * 4)), "--repeat", "3", "--batch", str(batch), "--beam", str(args.beam)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                out[name] = {k: d[k] for k in ("value", "ms_per_step", "precision", "policy", "checkpoint", "recompute_fraction",
                                               "rerun_fraction_with_padding", "tolerance", "pipeline_consistent", "pipeline_steps_checked",
                                               "ids_identical_to_exact")}
                out[name]["clips_per_sec"] = out[name].pop("value")
                out[name]["windows_clips_per_sec"] = d["windows"]["clips_per_sec"]
            except Exception as e:
                out[name] = {"error": f"{type(e).__name__}: {e}"[:300], "stderr": (r.stderr[-300:] if "r" in dir() else None)}
            continue
        prec, _, opt = name.partition(":")     # "bf16:g1" = the bf16 pipeline with ONE beam search per batch (round 3's schedule)
        env = dict(os.environ)
        if opt == "g1":
            env["CN_DEC_GROUP"] = "1"
        if opt == "e2":                        # "bf16:e2" = the encodes of consecutive batches on TWO streams (see s_encs in main)
            env["CN_ENC_STREAMS"] = "2"
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", prec, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--repeat", "3", "--batch", str(batch), "--beam", str(args.beam), "--cpu-clips", "0", "--parity-clips", "0", "--also", ""]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[name] = {"clips_per_sec": d["value"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                         "decode_group": d["config"].get("decode_group"), "decode_streams": d["config"].get("decode_streams"),
                         "encode_streams": d["config"].get("encode_streams"),
                         "windows_clips_per_sec": d["windows"]["clips_per_sec"], "pipeline_consistent": d["pipeline_consistent"]}
        except Exception as e:  # a failed child does not void the line: it is reported as such
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def parity_report(args, Engine, eng, sd, dev, w0, lens0, bos0, forbid, t_audio, beam, min_pred, max_pred, ora, result):
    """Ids / scores of every precision of the library against the CPU ORACLE on the clips of the cpu_baseline sample (the
    checker is oracle/cpu_ref.py, run above on the same waveforms), plus -- on all --parity-clips clips -- against the
    library's fp32 mode, with the encoder and the decoder switched to bf16 one at a time: frame_embs (B, T, 768) fp32 is
    the interface between the two, so "bf16 encoder + fp32 decoder" is engine A's encode fed to engine B's decode."""
    import torch
    n = min(args.parity_clips, w0.shape[0])
    wv, ln, bs = w0[:n].contiguous(), lens0[:n].contiguous(), bos0[:n].contiguous()
    engines = {args.precision: eng}
    for name in ("bf16", "f16", "bf16+f16dec", "mixed", "mixed16", "exact", "fp32"):
        if name not in engines:
            try:
                engines[name] = Engine(sd, precision=name, device=dev)
            except (KeyError, RuntimeError, ValueError) as e:  # a precision this build does not have
                print(f"[bench] parity: precision {name} unavailable ({e})", file=sys.stderr)
    fe, outs = {}, {}
    for name, e in engines.items():
        fe[name], _ = e.encode(wv)
        fe[name] = fe[name].clone()
        for bm_ in (1, beam):
            o = e.decode(fe[name], ln, bs, forbid, bm_, min_pred, max_pred)
            outs[(name, name, bm_)] = (_ids(o), o["best_lprobs"].cpu())
    mixes = [(a, b) for a in engines for b in engines if a != b and ({a, b} == {"bf16", "fp32"} or {a, b} == {"f16", "fp32"})]
    for enc_p, dec_p in mixes:
        for bm_ in (1, beam):
            o = engines[dec_p].decode(fe[enc_p], ln, bs, forbid, bm_, min_pred, max_pred)
            outs[(enc_p, dec_p, bm_)] = (_ids(o), o["best_lprobs"].cpu())
    torch.cuda.synchronize(dev)
    par = {"clips": n, "checker": "oracle/cpu_ref.py (CPU, stock PyTorch fp32) on the first clips of the same batch" if ora else
           "library fp32 mode only (no --cpu-clips)"}
    if ora is not None:
        vo = {}
        ofe = ora["frame_embs"]
        for (enc_p, dec_p, bm_), (ids, lp) in outs.items():
            key = enc_p if enc_p == dec_p else f"enc_{enc_p}+dec_{dec_p}"
            d = vo.setdefault(key, {})
            if bm_ == 1:
                d["greedy"] = compare_ids(ids, lp, ora["greedy"], None, ora["greedy_margin"])
            if bm_ == beam:
                d[f"beam{beam}"] = compare_ids(ids, lp, ora["beam"], ora["beam_lp"], ora["beam_margin"])
        for name in engines:
            if ofe is not None:
                g = fe[name][: ora["n"]].cpu()
                err = (g - ofe)
                vo[name]["frame_embs_rel_rms"] = round(float(err.pow(2).mean().sqrt() / ofe.pow(2).mean().sqrt()), 7)
                vo[name]["frame_embs_max_abs"] = round(float(err.abs().max()), 6)
        # the certified precision (round 6): the fp16 pipeline + device-side margins, uncertified clips re-run through the exact
        # context -- its ids must be the oracle's on EVERY clip (tests/test_gpu_certified.py asserts it; here it is reported)
        try:
            ce = Engine(sd, precision="certified", device=dev)
            cfe = ce.encode(wv)[0].clone()
            cd = {"base": ce.base_precision}
            for bm_ in (1, beam):
                o = ce.generate_certified(wv, cfe, ln, bs, forbid, bm_, min_pred, max_pred)
                key = "greedy" if bm_ == 1 else f"beam{beam}"
                cd[key] = compare_ids(_ids(o), o["best_lprobs"].cpu(), ora["greedy" if bm_ == 1 else "beam"],
                                      None if bm_ == 1 else ora["beam_lp"], ora["greedy_margin" if bm_ == 1 else "beam_margin"])
                cd[key]["recomputed_clips"] = [int(o["recomputed"].sum()), n]
            vo["certified"] = cd
            del ce
        except Exception as e_:
            vo["certified"] = {"error": f"{type(e_).__name__}: {e_}"[:300]}
        par["vs_oracle"] = vo
    if "fp32" in engines:  # the library's fp32 mode as the reference (its ids equal the oracle's above), on --parity-batches batches
        from conette_amd import synth as _synth

        def _cat(a, b):  # (ids, lprobs) of two batches: id matrices padded to a common width
            w_ = max(a[0].shape[1], b[0].shape[1])
            pa_ = torch.zeros((a[0].shape[0], w_), dtype=a[0].dtype)
            pb_ = torch.zeros((b[0].shape[0], w_), dtype=b[0].dtype)
            pa_[:, : a[0].shape[1]] = a[0]
            pb_[:, : b[0].shape[1]] = b[0]
            return torch.cat([pa_, pb_]), torch.cat([a[1], b[1]])

        allo = dict(outs)
        for bi in range(1, max(1, args.parity_batches)):
            wv_b = torch.from_numpy(_synth.synth_waveforms(n, wv.shape[1], 1234 + 7919 * bi)).to(dev)
            fe_b = {}
            for name, e in engines.items():
                fe_b[name] = e.encode(wv_b)[0].clone()
                for bm_ in (1, beam):
                    o = e.decode(fe_b[name], ln, bs, forbid, bm_, min_pred, max_pred)
                    allo[(name, name, bm_)] = _cat(allo[(name, name, bm_)], (_ids(o), o["best_lprobs"].cpu()))
            for enc_p, dec_p in mixes:
                for bm_ in (1, beam):
                    o = engines[dec_p].decode(fe_b[enc_p], ln, bs, forbid, bm_, min_pred, max_pred)
                    allo[(enc_p, dec_p, bm_)] = _cat(allo[(enc_p, dec_p, bm_)], (_ids(o), o["best_lprobs"].cpu()))
        torch.cuda.synchronize(dev)
        vf = {}
        r32 = {bm_: allo[("fp32", "fp32", bm_)] for bm_ in (1, beam)}
        for (enc_p, dec_p, bm_), (ids, lp) in allo.items():
            if enc_p == dec_p == "fp32":
                continue
            key = enc_p if enc_p == dec_p else f"enc_{enc_p}+dec_{dec_p}"
            vf.setdefault(key, {})["greedy" if bm_ == 1 else f"beam{beam}"] = compare_ids(ids, lp, r32[bm_][0], r32[bm_][1])
        par["vs_fp32_mode"] = vf
    # The same comparison on the PEAKED synthetic checkpoint (conette_amd.synth PEAKED, round 5: few candidates far above a noise
    # floor, like a trained captioner -- the default checkpoint's Gaussian logits put every decision within a few percent of a
    # logit of its runner-up, which is what the bf16 agreement above measures): the benchmark's clips, the 16-bit precisions
    # against the library's exact precision (whose ids equal the oracle's / the reference's: tests/test_peaked_checkpoint.py).
    try:
        import numpy as np
        from conette_amd import synth as _synth
        sd_pk = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in _synth.synth_state_dict(recipe="peaked").items()}
        pk_eng = {name: Engine(sd_pk, precision=name, device=dev) for name in ("exact", "bf16", "f16", "certified")}
        pk_re = {}
        pk_out = {}
        for bi in range(max(1, args.parity_batches)):
            wv_b = wv if bi == 0 else torch.from_numpy(_synth.synth_waveforms(n, wv.shape[1], 1234 + 7919 * bi)).to(dev)
            for name, e in pk_eng.items():
                fe_pk = e.encode(wv_b)[0].clone()
                for bm_ in (1, beam):
                    if name == "certified":
                        o = e.generate_certified(wv_b, fe_pk, ln, bs, forbid, bm_, min_pred, max_pred)
                        pk_re[bm_] = pk_re.get(bm_, 0) + int(o["recomputed"].sum())
                    else:
                        o = e.decode(fe_pk, ln, bs, forbid, bm_, min_pred, max_pred)
                    cur = (_ids(o), o["best_lprobs"].cpu())
                    if (name, bm_) in pk_out:
                        a = pk_out[(name, bm_)]
                        w_ = max(a[0].shape[1], cur[0].shape[1])
                        pa_ = torch.zeros((a[0].shape[0], w_), dtype=a[0].dtype)
                        pb_ = torch.zeros((cur[0].shape[0], w_), dtype=cur[0].dtype)
                        pa_[:, : a[0].shape[1]] = a[0]
                        pb_[:, : cur[0].shape[1]] = cur[0]
                        cur = (torch.cat([pa_, pb_]), torch.cat([a[1], cur[1]]))
                    pk_out[(name, bm_)] = cur
        torch.cuda.synchronize(dev)
        pk = {"reference": "the library's exact precision on the same clips (ids = the CPU oracle's on the committed peaked fixtures)",
              "checkpoint": "conette_amd.synth.synth_state_dict(recipe='peaked')"}
        for name in ("bf16", "f16", "certified"):
            pk[name] = {("greedy" if bm_ == 1 else f"beam{beam}"): compare_ids(pk_out[(name, bm_)][0], pk_out[(name, bm_)][1],
                                                                                pk_out[("exact", bm_)][0], pk_out[("exact", bm_)][1])
                        for bm_ in (1, beam)}
        for bm_ in (1, beam):
            pk["certified"]["greedy" if bm_ == 1 else f"beam{beam}"]["recomputed_clips"] = [pk_re.get(bm_, 0), pk_out[("exact", bm_)][0].shape[0]]
        par["peaked_checkpoint"] = pk
        del pk_eng
    except Exception as e_:   # (reported, never fatal: the leg is additional evidence)
        par["peaked_checkpoint"] = {"error": f"{type(e_).__name__}: {e_}"[:300]}
    # what the other precisions cost: the same two-slot encode + decode loop, un-pipelined (one stream)
    for name, e in engines.items():
        if name == args.precision:
            continue
        fe_b = [e.decode_input_buffer(n, t_audio, beam, max_pred, slot=i) for i in range(2)]
        cl_b = [torch.empty((n, 527), dtype=torch.float32, device=dev) for _ in range(2)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for i in range(3):
            e.encode(wv, out=(fe_b[i & 1], cl_b[i & 1]))
            e.decode(fe_b[i & 1], ln, bs, forbid, beam, min_pred, max_pred, clone=False, slot=i & 1)
        torch.cuda.synchronize(dev)
        k_ = 4
        enc_ms = dec_ms = 0.0
        t1 = time.perf_counter()
        for i in range(k_):
            ev[0].record()
            e.encode(wv, out=(fe_b[i & 1], cl_b[i & 1]))
            ev[1].record()
            e.decode(fe_b[i & 1], ln, bs, forbid, beam, min_pred, max_pred, clone=False, slot=i & 1)
            ev[2].record()
            torch.cuda.synchronize(dev)
            enc_ms += ev[0].elapsed_time(ev[1]) / k_
            dec_ms += ev[1].elapsed_time(ev[2]) / k_
        result[f"{name}_clips_per_sec_unpipelined"] = round(n * k_ / (time.perf_counter() - t1), 1)   # (encode then decode on one stream; `bench.py --precision NAME` is the pipelined number)
        result[f"{name}_encode_ms"], result[f"{name}_decode_ms"] = round(enc_ms, 3), round(dec_ms, 3)
        e.profile_enable(("frontend", "stem", "dwconv_ln", "pw1_gemm", "pw2_gemm", "downsample", "heads"))
        e.encode(wv, out=(fe_b[0], cl_b[0]))
        torch.cuda.synchronize(dev)
        result[f"{name}_stage_ms"] = {k: round(v[0], 4) for k, v in e.profile_read().items()}
        e.profile_enable(())
    return par


def main() -> None:
    args = parse_args()
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_rank:
        raise SystemExit(launch_ranks(args))       # (nothing above has touched a GPU)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` and let it start the ranks)")
    if args.backend == "gloo":
        return gloo_selftest(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist
    import conette_amd  # noqa: F401
    from conette_amd import synth

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # CN_BENCH_SHARE_GPU=1 (development aid, NOT a measurement): all ranks on cuda:0 with gloo collectives through host memory --
    # the only way to run the N > 1 code path (sharding, per-window timing tables, the gather, the consistency vote) on a
    # one-GPU box; RCCL refuses two ranks on one device
    share = os.environ.get("CN_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share else dev   # where the small collective tensors of this file live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # What the process group itself saw (not the WORLD_SIZE variable): its size, and the sum of one `1` per rank through the
    # collective backend -- the first real N-GPU line proves RCCL joined N ranks.
    world_observed, ones_sum = 1, 1
    if world > 1:
        world_observed = dist.get_world_size()
        one = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(one)
        ones_sum = int(one.item())
        if world_observed != world or ones_sum != world:
            raise SystemExit(f"bench: the process group has {world_observed} ranks and an all-reduce of ones gives {ones_sum}, but --gpus is {world}")

    from conette_amd.dist import gather_caption_windows, shard_bounds, trim_captions
    from conette_amd.engine import Engine

    beam, min_pred, max_pred = args.beam, 3, int(os.environ.get("CN_MAX_PRED", "20"))  # (CN_MAX_PRED: interference experiments only)
    strong = args.global_batch > 0
    if strong:
        lo, hi = shard_bounds(args.global_batch, rank, world)
        B, total_clips, clip0 = hi - lo, args.global_batch, lo
    else:
        B, total_clips, clip0 = args.batch, world * args.batch, rank * args.batch
    if B <= 0:
        raise SystemExit(f"rank {rank}: empty shard of --global-batch {args.global_batch}")
    sd_np = synth.synth_state_dict()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    eng = Engine(sd, precision=args.precision, device=dev)
    if os.environ.get("CN_NO_GRAPH"):
        eng.set_decode_graph(False)
    eng.set_encode_reserved_cus(int(os.environ.get("CN_ENC_RESERVE", "24")))
    bos_all = sd["model.task_id_to_token_id"]
    forbid = sd["model.forbid_rep_mask"].to(dev)

    if args.workload == "mixed":
        from conette_amd.bucketing import plan_buckets, plan_buckets_by_cost
        rng = np.random.default_rng(1234 + clip0)
        lengths = rng.integers(1 * SR, 30 * SR + 1, size=B)
        if os.environ.get("CN_BUCKET_SECONDS"):   # (round 3's planner: a padded-seconds budget per bucket)
            buckets = plan_buckets(lengths.tolist(), max_padded_seconds=float(os.environ["CN_BUCKET_SECONDS"]), sr=SR)
        else:                                     # minimum modelled cost: padded audio + a fixed cost per bucket
            buckets = plan_buckets_by_cost(lengths.tolist(), fixed_cost_seconds=float(os.environ.get("CN_BUCKET_FIXED", "150")), sr=SR)
        batches = []
        for idx in buckets:
            ls = [int(lengths[i]) for i in idx]
            w_ = torch.from_numpy(synth.synth_waveforms(len(idx), max(ls), 1234 + clip0 + idx[0], lengths=ls)).to(dev)
            t_ = eng.lib.conette_num_audio_frames(max(ls))
            lens_ = torch.tensor(ls, dtype=torch.float32).div(max(ls) // t_).round().to(torch.int32).to(dev)  # convnext.py:312-315
            batches.append((w_, lens_, t_))
        audio_seconds = float(lengths.sum()) / SR
    else:
        L = CLIP_S * SR
        wave = torch.from_numpy(synth.synth_waveforms(B, L, 1234 + clip0)).to(dev)
        t_audio = eng.lib.conette_num_audio_frames(L)
        batches = [(wave, torch.full((B,), t_audio, dtype=torch.int32, device=dev), t_audio)]
        audio_seconds = float(B * CLIP_S)
    # Fixed workload: the timed steps walk round-robin over NB DISTINCT input batches (a real stream never re-reads its input;
    # one 82 MB waveform tensor encoded every step would be served from the 256 MiB Infinity Cache -- VERDICT r04).  Clip j of
    # rotation k is seeded by its global index j + k x total clips, so a sharded job sees the clips of the single-GPU job.
    NB = max(1, int(os.environ.get("CN_BENCH_ROTATE", "4"))) if args.workload == "fixed" else 1
    waves = None
    if args.workload == "fixed":
        waves = [wave] + [torch.from_numpy(synth.synth_waveforms(B, L, 1234 + clip0 + k * total_clips)).to(dev) for k in range(1, NB)]

    # Two pipeline slots and two HIP streams: the decode of batch i (small latency-bound launches, replayed from a
    # hipGraph) overlaps the encode of batch i+1 (big MFMA / VALU kernels; their persistent kernels leave
    # CN_ENC_RESERVE compute units to the decode stream).
    prio = int(os.environ.get("CN_DEC_PRIO", "-1"))  # decode stream priority (negative = higher)
    # CN_ENC_STREAMS=2: the encodes of consecutive batches on alternating streams -- the blocks of one batch start on the compute
    # units the previous batch's tail has left.  Round 3: 3 % slower than one stream; round 5 (fp16 stream, 8-row decode blocks):
    # +4.7 % (13.3 k against 12.75 k clips/s, same box) -- but two encodes in flight time-slice the compute units, so a launch's
    # wall time (HIP events or rocprof alike) is no longer ITS time: 248 us per launch of the dominant class against 148, a
    # "roofline.frac" of 0.20 that measures queueing.  The line's `value` and `roofline` therefore come from ONE encode stream (a
    # launch owns the chip, its duration is the kernel's); the two-stream rate is reported beside it as `also_pipelined["bf16:e2"]`.
    n_enc = int(os.environ.get("CN_ENC_STREAMS", "1"))
    s_encs = [torch.cuda.Stream(dev, priority=int(os.environ.get("CN_ENC_PRIO", "0"))) for _ in range(max(1, min(n_enc, 2)))]
    # Grouped decode (fixed workload): the beam search of G consecutive batches runs as ONE chain over G x B clips, launched when
    # the G-th of them has been encoded.  A search is a latency chain whose kernels re-fetch every layer's weights once per
    # XCD and step whatever the row count, and that traffic is what a decode costs the encoder running beside it
    # (profiles/r04_notes.md section 3: ~0.1 ms per GB): one chain per G batches moves 1 / G of it per clip and launches 1 / G of
    # the kernels.  Every step still encodes its own batch of B clips; a batch's captions leave the device up to G steps later.
    # Captions do not depend on the grouping (row-local kernels; checked step by step against the solo pass like everything else).
    # G = the largest divisor of --steps with G x B <= 256 clips (and <= CN_DEC_GROUP): a timed window ends on a group boundary, so
    # every step's decode completes inside its window; beyond ~256 clips per chain the weights are amortised and a longer
    # chain only adds latency (one box: B = 256: G 1 / 2 / 4 -> 11 030 / 10 868 / 10 805 clips/s; B = 16: G 1 / 4 / 8 / 16 -> 6 567 / 7 049 /
    # 7 188 / 7 339).  (G x B x beam also stays under the 4 096 rows at which the GEMM tile shapes change.)
    G = choose_decode_group(args.steps, B, beam, int(os.environ.get("CN_DEC_GROUP", str(DEC_GROUP_DEFAULT)))) if args.workload == "fixed" else 1
    # CN_DEC_STREAMS decode chains in flight: a decode is a serial chain of ~300 latency-bound launches that stretches to the
    # length of an encode when it shares the chip with one; ungrouped, two chains (batches i-1 and i-2 decode while batch i
    # encodes; three in the precisions with the exact decoder) let a chain take two steps, and the step is bounded by the
    # encoder again.  A chain over G >= 3 batches has G steps to finish: one stream; G = 2: two.  (Measured, one box, bf16 /
    # mixed16 clips/s: G 1 x 2 streams 10 161-10 261 / 8 849, G 2 x 2 10 849 / 9 757, G 3 x 1 10 893 / 9 866, G 4 x 1 10 877 / 9 946,
    # G 6 x 1 10 364 / 9 673; four and more chains: 2x slower.)
    # (mixed-length workload: every bucket of a step is a decode chain of its own -- three streams, dealt bucket by bucket)
    # two schedules exist side by side when G > 1: the grouped one (n_decg chains, n_slotg group slots) and the one-search-per-
    # batch one (n_dec chains, n_slot slots); which of them runs the timed region is MEASURED below (on the
    # driver's box G = 1 won for bf16 while the rule picked G = 4)
    n_decg = max(1, min(int(os.environ.get("CN_DEC_STREAMS", "1" if G >= 3 else "2")), 3))
    n_slotg = n_decg + 1
    n_dec_default = 3 if (args.precision in ("exact", "mixed", "mixed16") or args.workload == "mixed") else 2
    n_dec = max(1, min(int(os.environ.get("CN_DEC_STREAMS", str(n_dec_default))), 3))
    dec_cus = int(os.environ.get("CN_DEC_CUS", "0"))   # > 0: the decode chains are confined to this many compute units (CU-masked streams)
    if dec_cus > 0:
        from conette_amd.engine import make_masked_stream
        s_decs = [make_masked_stream(dev, dec_cus) for _ in range(max(n_dec, n_decg))]
    else:
        s_decs = [torch.cuda.Stream(dev, priority=prio) for _ in range(max(n_dec, n_decg))]
    n_slot = n_dec + 1
    from conette_amd.engine import MAX_DECODE_GRAPHS
    if len(batches) * n_slot > MAX_DECODE_GRAPHS:
        raise SystemExit(f"bench: {len(batches)} length buckets x {n_slot} pipeline slots = {len(batches) * n_slot} decode keys exceed the "
                         f"library's {MAX_DECODE_GRAPHS} cached hipGraphs: every decode would run eagerly (~300 launches); raise "
                         f"CN_BUCKET_SECONDS or lower --batch")
    slots = []
    for k, (w_, lens_, t_) in enumerate(batches):
        for sl in range(n_slot):
            slots.append(dict(
                fe=eng.decode_input_buffer(w_.shape[0], t_, beam, max_pred, slot=n_slot * k + sl),
                clip=torch.empty((w_.shape[0], 527), dtype=torch.float32, device=dev),
                enc_done=torch.cuda.Event(), dec_done=torch.cuda.Event()))
    state = {"i": 0, "last": None}
    bos_dev = [bos_all[torch.zeros(w_.shape[0], dtype=torch.long)].to(dev) for w_, _, _ in batches]  # task "clotho"

    if G > 1:
        w_g, lens_g, t_g = batches[0]
        gslots = []
        for sl in range(n_slotg):
            fe_big = eng.decode_input_buffer(G * B, t_g, beam, max_pred, slot=100 + sl)
            gslots.append(dict(fe=fe_big, clip=torch.empty((B, 527), dtype=torch.float32, device=dev),
                               enc_done=[torch.cuda.Event() for _ in range(G)], dec_done=torch.cuda.Event()))
        lens_big, bos_big = lens_g.repeat(G), bos_dev[0].repeat(G)

    def step_grouped():
        """one pass over this rank's B clips; every G-th call also launches the beam search of the last G batches"""
        i = state["i"]
        g, m = i // G, i % G
        gs = gslots[g % n_slotg]
        s_enc = s_encs[i % len(s_encs)]
        s_dec = s_decs[g % n_decg]
        if state.get("prof") is not None:   # the dominant class is event-timed on every prof_every-th step of the timed region
            eng.profile_enable(state["prof"] if i % prof_every == 0 else ())
        probe = state.get("gap_probe")
        with torch.cuda.stream(s_enc):
            if probe is not None:
                pe = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                pe[0].record(s_enc)
            if g >= n_slotg:
                # the group slot's frame buffer is free again.  EVERY encode of the group waits: with two encoder streams
                # (CN_ENC_STREAMS=2) the m >= 1 encodes run on the stream that did not wait at m == 0
                s_enc.wait_event(gs["dec_done"])
            if probe is not None:
                pe[1].record(s_enc)
            eng.encode(waves[i % NB], out=(gs["fe"][m * B:(m + 1) * B], gs["clip"]), slot=i & 1)
            gs["enc_done"][m].record(s_enc)
            if probe is not None:
                pe[2].record(s_enc)
                probe.append((m, pe))
        if m == G - 1:
            with torch.cuda.stream(s_dec):
                for e_ in gs["enc_done"]:
                    s_dec.wait_event(e_)
                out = eng.decode(gs["fe"], lens_big, bos_big, forbid, beam, min_pred, max_pred, clone=False, slot=100 + (g % n_slotg))
                preds, lps = out["best_preds"], out["best_lprobs"]
                state["last_local"] = (preds[(G - 1) * B:], lps[(G - 1) * B:])   # the captions of this step's own batch
                if state.get("keep") is not None and i < state["keep"][0].shape[0]:
                    kp, kl = state["keep"]                                      # steps g G .. g G + G - 1 of the keep tables
                    kp[g * G:(g + 1) * G].copy_(preds.view(G, B, -1)[:, :, : kp.shape[2]], non_blocking=True)
                    kl[g * G:(g + 1) * G].copy_(lps.view(G, B), non_blocking=True)
                gs["dec_done"].record(s_dec)
            state["last"] = out
            state["last_stream"] = s_dec
        state["i"] = i + 1

    # this rank's captions of a step, staged per pipeline slot: every bucket copies its rows on ITS OWN decode stream before it
    # records dec_done -- the slot's persistent output buffers may be overwritten by decode(i + n_slot) as soon as that event
    # has fired, whatever another stream is still doing (the concatenation used to run later, on the last stream)
    local_p = torch.zeros((n_slot, B, max_pred), dtype=torch.int32, device=dev)
    local_l = torch.zeros((n_slot, B), dtype=torch.float32, device=dev)
    bucket_off = [0]
    for w_, _, _ in batches:
        bucket_off.append(bucket_off[-1] + w_.shape[0])

    def step():
        """one pass over this rank's clips: every (length-bucketed) batch once"""
        i = state["i"]
        lp_s, ll_s = local_p[i % n_slot], local_l[i % n_slot]
        keep = state.get("keep") if (state.get("keep") is not None and i < state["keep"][0].shape[0]) else None
        if state.get("prof") is not None:
            eng.profile_enable(state["prof"] if i % prof_every == 0 else ())
        for k, (w_, lens_, t_) in enumerate(batches):
            sl = slots[n_slot * k + (i % n_slot)]
            bos = bos_dev[k]
            s_enc = s_encs[i % len(s_encs)]
            s_dec = s_decs[(i * len(batches) + k) % n_dec]   # consecutive buckets decode on different streams
            o0, o1 = bucket_off[k], bucket_off[k + 1]
            with torch.cuda.stream(s_enc):
                if i >= n_slot:
                    s_enc.wait_event(sl["dec_done"])          # slot's frame buffer (and its staging rows) are free again
                eng.encode(w_ if waves is None else waves[i % NB], out=(sl["fe"], sl["clip"]), slot=i & 1)
                sl["enc_done"].record(s_enc)
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(sl["enc_done"])
                out = eng.decode(sl["fe"], lens_, bos, forbid, beam, min_pred, max_pred, clone=False, slot=n_slot * k + (i % n_slot))
                preds, lps = out["best_preds"], out["best_lprobs"]
                wcp = min(preds.shape[1], lp_s.shape[1])
                lp_s[o0:o1, :wcp].copy_(preds[:, :wcp], non_blocking=True)
                ll_s[o0:o1].copy_(lps, non_blocking=True)
                if keep is not None:   # every local caption of every timed step: compared with the solo pass after the run, gathered per window
                    kp, kl = keep
                    kp[i, o0:o1].copy_(preds[:, : kp.shape[2]], non_blocking=True)
                    kl[i, o0:o1].copy_(lps, non_blocking=True)
                sl["dec_done"].record(s_dec)
        state["last_local"] = (lp_s, ll_s)   # this rank's captions of the step (before the all-gather); complete once every decode stream has drained
        state["i"] = i + 1
        state["last"] = out
        state["last_stream"] = s_decs[(i * len(batches) + len(batches) - 1) % n_dec]
        return lp_s, ll_s

    def gather_window(first, count):
        """N > 1: the ONE collective of a timed window -- ids + scores of all its steps, all ranks, in clip order (north_star:
        'an RCCL all-gather only to return final token ids').  Enqueued behind the window's last decode on that decode's
        stream, so no rank waits for another inside the window: a per-step gather would be a per-step rendezvous."""
        kp, kl = state["keep"]
        with torch.cuda.stream(state["last_stream"]):
            for s_ in s_decs:
                state["last_stream"].wait_stream(s_)
            return gather_caption_windows(kp[first: first + count], kl[first: first + count], total_clips)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def warm_steps(g_):   # >= 3 calls per slot: the third identical decode call replays its hipGraph
        return max(args.warmup, 3 * n_slot) if g_ == 1 else (max(args.warmup, 3 * n_slotg * g_) + g_ - 1) // g_ * g_

    # Which decode schedule runs the timed region is decided by MEASUREMENT: both candidates -- one search
    # per G batches (the rule's G) and one search per batch -- run their warm-up and two untimed windows of --steps steps; the
    # faster one (by its better window; over all ranks: the slowest rank's) is kept and both rates go into the JSON line.
    # CN_DEC_GROUP set explicitly (or CN_SCHED_TRIAL=0) skips the trial.
    schedule = None
    if args.workload == "fixed" and G > 1 and os.environ.get("CN_DEC_GROUP") is None and os.environ.get("CN_SCHED_TRIAL", "1") != "0":
        def trial(fn, warm):
            state["i"] = 0
            for _ in range(warm):
                fn()
            fence()
            best_dt = float("inf")
            for _ in range(2):
                t0_ = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize(dev)
                best_dt = min(best_dt, time.perf_counter() - t0_)
                fence()
            return best_dt
        t_grouped, t_single = trial(step_grouped, warm_steps(G)), trial(step, warm_steps(1))
        if world > 1:
            tt = torch.tensor([t_grouped, t_single], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_grouped, t_single = float(tt[0]), float(tt[1])
        schedule = {"candidates": {f"G{G}x{n_decg}": round(total_clips * args.steps / t_grouped, 1),
                                   f"G1x{n_dec}": round(total_clips * args.steps / t_single, 1)},
                    "unit": "clips/s (better of two untimed windows)", "rule_G": G}
        if t_single < t_grouped:
            G = 1
        schedule["chosen"] = f"G{G}x{n_decg if G > 1 else n_dec}"
        state["i"] = 0
    warm_used = warm_steps(G)
    if G > 1:
        step_single, step = step, step_grouped
    for _ in range(warm_used):
        step()
    fence()
    # The timed region is at least CN_MIN_TIMED_S (1 s) long whoever calls: `--steps` stays what the caller passed, the number of
    # windows is raised from `--repeat` until steps x windows fill that time (the driver's --steps 20 gave five
    # 98 ms windows).  One untimed window measures the step; every rank takes the slowest rank's estimate.
    t_est = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    t_est = time.perf_counter() - t_est
    if world > 1:
        te_ = torch.tensor([t_est], dtype=torch.float64, device=cdev)
        dist.all_reduce(te_, op=dist.ReduceOp.MAX)
        t_est = float(te_[0])
    fence()
    min_timed_s = float(os.environ.get("CN_MIN_TIMED_S", "1.0"))
    repeat_used = max(1, args.repeat, min(64, int(math.ceil(min_timed_s / max(t_est, 1e-6)))))

    # ---- pre-pass (untimed): which kernel class dominates, encode / decode split --------------------
    w0, lens0, t0_ = batches[0]
    B0 = w0.shape[0]
    bos0 = bos_dev[0]
    enc_classes = ("frontend", "stem", "dwconv_ln", "pw1_gemm", "pw2_gemm", "downsample", "heads")
    eng.profile_enable(enc_classes)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    eng.encode(w0, out=(slots[0]["fe"], slots[0]["clip"]))     # (straight behind the warm-up steps: the chip is at its working clock)
    ev[1].record()
    if G > 1:   # (the grouped pipeline never ran the one-batch search: two calls so that the timed one below replays its graph)
        for _ in range(2):
            eng.decode(slots[0]["fe"], lens0, bos0, forbid, beam, min_pred, max_pred, clone=False, slot=0)
    ev[2].record()
    out = eng.decode(slots[0]["fe"], lens0, bos0, forbid, beam, min_pred, max_pred, clone=False, slot=0)
    ev[3].record()
    torch.cuda.synchronize(dev)
    pre = eng.profile_read()
    eng.profile_enable(())
    encode_ms, decode_ms = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
    stage_ms = {k: round(v[0], 4) for k, v in pre.items()}
    dominant = max(("pw1_gemm", "pw2_gemm", "dwconv_ln"), key=lambda k: pre.get(k, (0.0, 0))[0])
    solo_preds, solo_lps = out["best_preds"].clone(), out["best_lprobs"].clone()  # un-pipelined result of batch 0
    solo_rot_p, solo_rot_l = [solo_preds], [solo_lps]       # un-pipelined results of every batch of the rotation
    for k_ in range(1, NB):
        eng.encode(waves[k_], out=(slots[0]["fe"], slots[0]["clip"]))
        o_ = eng.decode(slots[0]["fe"], lens0, bos0, forbid, beam, min_pred, max_pred, clone=False, slot=0)
        solo_rot_p.append(o_["best_preds"].clone())
        solo_rot_l.append(o_["best_lprobs"].clone())
    solo_rot_p, solo_rot_l = torch.stack(solo_rot_p), torch.stack(solo_rot_l)      # (NB, B, max_pred), (NB, B)
    if NB > 1:   # (the pre-pass below and the parity legs work on batch 0: its frame embeddings back into slot 0)
        eng.encode(w0, out=(slots[0]["fe"], slots[0]["clip"]))
    bm = int(out["sizes"][1].item())
    best = out["best_preds"][:, :bm]
    best_tokens = int((best != 0).sum().item())            # tokens of the returned captions (<eos> included)
    beam_tokens = int((out["mult_preds"] != 0).sum().item())  # row-steps of every hypothesis of the search
    decode_ms_grouped = graph_nodes_grouped = None
    if G > 1:   # the chain the pipeline actually runs: G batches per beam search, solo, replayed from its graph
        gs0 = gslots[0]
        for m_ in range(G):
            gs0["fe"][m_ * B:(m_ + 1) * B].copy_(slots[0]["fe"])
        evg = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        evg[0].record()
        og = eng.decode(gs0["fe"], lens_big, bos_big, forbid, beam, min_pred, max_pred, clone=False, slot=100)
        evg[1].record()
        torch.cuda.synchronize(dev)
        decode_ms_grouped = evg[0].elapsed_time(evg[1])
        graph_nodes_grouped = eng.decode_graph_nodes()
        # (the grouped search returns, for each of its G copies of batch 0, exactly the solo search's captions and scores)
        if not (torch.equal(og["best_preds"].view(G, B, -1), solo_preds[None].expand(G, -1, -1)) and
                torch.equal(og["best_lprobs"].view(G, B), solo_lps[None].expand(G, -1))):
            raise SystemExit("bench: the grouped beam search returned other captions than the search of one batch")

    # ---- timed region: R windows of exactly K steps each, events only around the dominant class ---------
    # (every window is bracketed by barrier + synchronize on both sides; the reported value is the MEDIAN window's)
    # The dominant class's launches are bracketed by HIP events on the encode stream in every prof_every-th timed step (1 = every step):
    # an event pair per launch is 36 extra packets per step on that stream, ~2 % of the step when every step carries them.
    # The stride is kept coprime with the decode group: step i is encode i % G of its group, the encodes of a group overlap the previous
    # group's search to different degrees (gap probe, profiles/r05_notes.md section 8: 5.35 / 5.25 / 4.81 / 4.49 ms at G = 4), and a
    # stride of G would time the same one every time.
    prof_every = max(1, int(os.environ.get("CN_PROF_EVERY", "5")))
    while prof_every > 1 and math.gcd(prof_every, G) > 1:
        prof_every += 1
    state["prof"] = (dominant,)
    state["i"] = 0
    n_rep = repeat_used
    n_timed = n_rep * args.steps
    n_keep = n_timed if (world > 1 or os.environ.get("CN_BENCH_CHECK_ALL", "1") != "0") else 0  # (0: only the last step is compared)
    state["keep"] = (torch.zeros((n_keep, B, solo_preds.shape[1]), dtype=solo_preds.dtype, device=dev),
                     torch.zeros((n_keep, B), dtype=solo_lps.dtype, device=dev))
    win_dt, own_dt, issue_dt = [], [], []
    gathered = None
    if os.environ.get("CN_GAP_PROBE") and G > 1:   # lab: where the encode stream idles (stderr; events around every encode)
        state["gap_probe"] = []
    for w_i in range(n_rep):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        issue_dt.append(time.perf_counter() - t0)   # the host has ENQUEUED the window's work (it runs ahead of the device)
        if world > 1:      # (this rank's own work ends where its last decode ends; the gather behind it waits for the others)
            for s_ in s_decs + s_encs:
                s_.synchronize()
            own_dt.append(time.perf_counter() - t0)
            gathered = gather_window(w_i * args.steps, args.steps)   # inside the timed window: the path's one collective
            torch.cuda.synchronize(dev)
        else:
            torch.cuda.synchronize(dev)
            own_dt.append(time.perf_counter() - t0)   # this rank's own work
        fence()
        win_dt.append(time.perf_counter() - t0)
    if state.get("gap_probe"):
        pr = state.pop("gap_probe")
        for m_ in range(G):
            rows_ = [(pr[j - 1][1][2].elapsed_time(pr[j][1][0]), pr[j][1][0].elapsed_time(pr[j][1][1]), pr[j][1][1].elapsed_time(pr[j][1][2]))
                     for j in range(1, len(pr)) if pr[j][0] == m_]
            a_ = np.array(rows_)
            print(f"[bench] gap probe m={m_}: after previous encode {a_[:, 0].mean() * 1e3:.0f} us (max {a_[:, 0].max() * 1e3:.0f}), "
                f"waiting for the slot {a_[:, 1].mean() * 1e3:.0f} us (max {a_[:, 1].max() * 1e3:.0f}), encode {a_[:, 2].mean():.3f} ms", file=sys.stderr, flush=True)
    # the pipelined steps (encode of batch i next to the decodes of batches i-1, i-2) must reproduce the solo pass bit for bit
    lp_, ll_ = state["last_local"][0][:B0], state["last_local"][1][:B0]
    kp_all, kl_all = state["keep"]
    state["keep"] = None
    kp, kl = kp_all[:, :B0], kl_all[:, :B0]      # batch 0 (the only one of the fixed workload) of every timed step against the solo pass
    wp = min(kp.shape[2], lp_.shape[1])  # timed steps whose captions / scores of batch 0 differ from the solo pass
    rot = torch.arange(kp.shape[0], device=dev) % NB   # step i encoded batch i % NB of the rotation
    bad_steps = int(((kp[:, :, :wp] != solo_rot_p[rot][:, :B0, :wp]).flatten(1).any(dim=1) | (kl != solo_rot_l[rot][:, :B0]).flatten(1).any(dim=1)).sum().item()) if kp.shape[0] else 0
    solo_preds, solo_lps = solo_rot_p[(n_timed - 1) % NB], solo_rot_l[(n_timed - 1) % NB]   # the last timed step's batch
    # what the job returns: the (all-gathered) ids of the last timed step, every clip, in clip order.  Its hash lets a sharded run be
    # compared with the single-GPU run of the same clips (tests/test_gpu_bench_n2.py); on N > 1 every rank must hold the same table.
    import hashlib
    final_ids = gathered[0][-1] if gathered is not None else state["last_local"][0]
    final_ids = trim_captions(final_ids.cpu()) if final_ids.numel() else final_ids.cpu()
    captions_sha = hashlib.sha256(final_ids.to(torch.int32).contiguous().numpy().tobytes()).hexdigest()
    gather_consistent = None
    if world > 1:   # my rows of the gathered table (last window) are what I decoded, scores included; and every rank holds the same table
        first = (n_rep - 1) * args.steps
        ok_g = bool(torch.equal(gathered[0][:, clip0: clip0 + B], kp_all[first: first + args.steps]) and
                    torch.equal(gathered[1][:, clip0: clip0 + B], kl_all[first: first + args.steps]))
        dig = torch.tensor(list(bytes.fromhex(captions_sha)), dtype=torch.int32, device=cdev)
        lo_d, hi_d = dig.clone(), dig.clone()
        dist.all_reduce(lo_d, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_d, op=dist.ReduceOp.MAX)
        flag_g = torch.tensor([1 if (ok_g and torch.equal(lo_d, hi_d)) else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag_g, op=dist.ReduceOp.MIN)
        gather_consistent = bool(flag_g.item())
        if not gather_consistent:
            raise SystemExit("bench: the all-gathered caption table differs between ranks or from a rank's own results")
    pipeline_consistent = bool(bad_steps == 0 and torch.equal(lp_[:, : solo_preds.shape[1]], solo_preds[: lp_.shape[0], : lp_.shape[1]]) and
                               torch.equal(ll_, solo_lps[: ll_.shape[0]]))
    if not pipeline_consistent:
        a_, b_ = lp_[:, : solo_preds.shape[1]], solo_preds[: lp_.shape[0], : lp_.shape[1]]
        rows = (a_ != b_).any(dim=1).nonzero().flatten().tolist()
        print(f"[bench] rank {rank}: {bad_steps} of {n_timed} pipelined steps returned other captions / scores than the un-pipelined pass of the same batch; last step: rows",
              rows[:16], "of", a_.shape[0], "| score diffs:", int((ll_ != solo_lps[: ll_.shape[0]]).sum()), file=sys.stderr, flush=True)
    if world > 1:  # every rank must agree before a number is printed
        flag = torch.tensor([1 if pipeline_consistent else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        pipeline_consistent = bool(flag.item())
    if not pipeline_consistent and os.environ.get("CN_BENCH_STRICT", "1") != "0":
        raise SystemExit("bench: pipelined and un-pipelined results differ (CN_BENCH_STRICT=0 reports the number anyway, with pipeline_consistent false)")
    state["prof"] = None
    prof = eng.profile_read()
    eng.profile_enable(())
    rank_rates = None
    if world > 1:  # a window's time is the slowest rank's; per-rank rates show an imbalance between the GPUs
        mine = torch.tensor(win_dt + own_dt + [float(B), audio_seconds], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        tab = torch.stack(allr).cpu()                       # (world, 2 n_rep + 2)
        win_dt = [float(v) for v in tab[:, :n_rep].max(dim=0).values.tolist()]
        audio_seconds_all = float(tab[:, 2 * n_rep + 1].sum())
        rank_tab = tab
    else:
        audio_seconds_all = audio_seconds
        rank_tab = None
    order = sorted(range(n_rep), key=lambda k: win_dt[k])
    med = order[(n_rep - 1) // 2]          # the median window (lower median for an even count)
    dt = win_dt[med]
    if rank_tab is not None:  # clips/s of every rank in the median window (its own clock, its own shard)
        rr = (rank_tab[:, 2 * n_rep] * args.steps / rank_tab[:, n_rep + med]).tolist()
        rank_rates = {"min": round(min(rr), 2), "max": round(max(rr), 2), "per_rank": [round(v, 2) for v in rr]}

    dom_ms, dom_n = prof[dominant]
    n_prof_steps = sum(1 for i_ in range(n_timed) if i_ % prof_every == 0)
    launches_per_step = dom_n / n_prof_steps
    avg_launch_s = dom_ms * 1e-3 / dom_n
    roof = {"kernel": dominant, "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches_per_step": launches_per_step,
            "launches_timed": dom_n, "timed_steps_with_events": n_prof_steps,
            "traffic": None}
    if args.workload == "fixed":
        fl, by = algorithmic_work(dominant, B, 2.0 if args.precision in ("bf16", "f16", "bf16+f16dec", "mixed", "mixed16") else 4.0)
        if dominant in ("pw1_gemm", "pw2_gemm"):
            achieved = fl / launches_per_step / avg_launch_s / 1e12
            roof.update({"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4)})
        else:
            achieved = by / launches_per_step / avg_launch_s / 1e9
            roof.update({"bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM_GBS, 4)})
        if args.precision == "bf16":
            tr, src = pmc_traffic(dominant, B, launches_per_step)
            if tr is not None:
                roof["traffic"] = round(tr)            # HBM bytes per launch (class average), PMC
                roof["traffic_unit"] = "bytes/launch"
                roof["traffic_source"] = "profiles/" + src
                roof["traffic_measured_on_this_build"] = pmc_is_current(os.path.join(ROOT, "profiles", src))
                roof["algorithmic_bytes_per_launch"] = round(by / launches_per_step)
            mb, src = pmc_mfma_busy(dominant)
            if mb is not None:
                roof["mfma_busy"] = mb                 # share of SIMD cycles with the matrix pipe busy, PMC
                roof["mfma_busy_source"] = "profiles/" + src
                roof["mfma_busy_measured_on_this_build"] = pmc_is_current(os.path.join(ROOT, "profiles", src))

    result = None
    if rank == 0:
        clips_per_s = total_clips * args.steps / dt
        if args.workload == "mixed":
            wl = (f"{total_clips} clips of U[1 s, 30 s] @ 32 kHz (seed 1234), {len(batches)} length buckets per rank, "
                  f"ConvNeXt encoder + beam-{beam} KV-cached decode")
        else:
            wl = (f"B={B}/GPU x 10 s @ 32 kHz clips, ConvNeXt encoder + beam-{beam} KV-cached decode "
                  f"(min 3 / max 20 tokens, V=5631), synthetic seeded checkpoint")
        result = {
            "metric": "clips_per_sec", "value": round(clips_per_s, 2), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "warmup_run": warm_used, "ms_per_step": round(dt / args.steps * 1e3, 3), "pipeline_consistent": pipeline_consistent, "pipeline_steps_checked": n_keep if n_keep else 1,
            "captions_sha256": captions_sha, "gather": ({"collectives_per_window": 2, "consistent_across_ranks": gather_consistent,
                                                        "what": "ids (K, B, max_pred) int32 + scores (K, B) fp32 of all K steps of a window, once per window"} if world > 1 else None),
            "timed_region_s": round(dt, 4), "repeat": n_rep, "repeat_requested": args.repeat, "min_timed_total_s": min_timed_s,
            "host_enqueue_ms_per_step": round(issue_dt[med] / args.steps * 1e3, 3),   # host time to enqueue a step (median window): below ms_per_step = the device, not the host, bounds the step
            "windows": {"clips_per_sec": [round(total_clips * args.steps / w, 2) for w in win_dt], "median_index": med,
                        "spread": round((max(win_dt) - min(win_dt)) / dt, 4), "timed_total_s": round(sum(win_dt), 4)},
            "rank_clips_per_sec": rank_rates,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "exact": "f16x2", "mixed": "bf16 encoder + f16x2 decoder", "f16": "f16",
                      "mixed16": "f16 encoder + f16x2 decoder", "bf16+f16dec": "bf16 encoder + f16 decoder"}[args.precision], "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": B, "global_batch": total_clips, "beam_size": beam,
                       "parallelism": f"dp{world}", "world_size_observed": world_observed, "allreduce_of_ones": ones_sum,
                       "collective_backend": (None if world == 1 else dist.get_backend()),
                       "decode_group": G, "decode_streams": n_decg if G > 1 else n_dec, "encode_streams": len(s_encs), "input_batches_rotated": NB,
                       **({"share_gpu_selftest": True} if share else {})},
            "audio_seconds_per_sec": round(audio_seconds_all * args.steps / dt, 1),
            "decode_tokens_per_sec": round(world * best_tokens / (decode_ms * 1e-3), 1),   # solo decode (pre-pass)
            "decode_tokens_per_sec_pipelined": round(total_clips / B0 * best_tokens * args.steps / dt, 1) if args.workload == "fixed" else None,
            "decode_beam_row_steps_per_sec": round(world * beam_tokens / (decode_ms * 1e-3), 1),
            "encode_ms": round(encode_ms, 3), "decode_ms": round(decode_ms, 3), "stage_ms": stage_ms,
            "decode_grouped": ({"batches_per_search": G, "search_ms": round(decode_ms_grouped, 3), "ms_per_batch": round(decode_ms_grouped / G, 3),
                                "launches_per_batch_step": round(graph_nodes_grouped / max_pred / G, 2)} if G > 1 else None),
            "roofline": roof,
            "decode_schedule": schedule,
        }
        if args.workload == "fixed":
            n_steps = int(out["sizes"][0].item())
            dby = decode_algorithmic_bytes(B0, beam, t0_, n_steps, eng.vocab_size)
            result["roofline_decode"] = {
                "kernel": "conette_decode (whole search, solo pre-pass)", "bound": "hbm", "achieved": round(dby / (decode_ms * 1e-3) / 1e9, 1),
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(dby / (decode_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "algorithmic_bytes_per_search": dby, "search_steps": n_steps, "graph_nodes_per_search": eng.decode_graph_nodes(),
                "launches_per_step": round(eng.decode_graph_nodes() / max_pred, 1),
                "traffic": None}
            if G > 1:   # the chain the pipeline runs: G batches per search (weights touched once per search, not once per batch)
                dbg = decode_algorithmic_bytes(G * B0, beam, t0_, n_steps, eng.vocab_size)
                result["roofline_decode"]["grouped"] = {
                    "batches_per_search": G, "algorithmic_bytes_per_search": dbg, "search_ms": round(decode_ms_grouped, 3),
                    "achieved": round(dbg / (decode_ms_grouped * 1e-3) / 1e9, 1), "unit": "GB/s",
                    "frac": round(dbg / (decode_ms_grouped * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}

    # ---- CPU baseline (rank 0, N = 1 only): the oracle restatement on host cores; its outputs are KEPT as the checker ----
    ora = None
    if rank == 0 and world == 1 and args.cpu_clips > 0:
        from oracle import cpu_ref as O

        n_thr = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n_thr)
        cfg = synth.synth_config_dict()
        nc = min(args.cpu_clips, B0)
        xs = w0[:nc].cpu()[:, None, :]
        with torch.no_grad():
            tc = time.perf_counter()
            otaps, otrace, gtrace = {}, [], []
            o3 = O.model_forward(sd, cfg, xs, sr=SR, task="clotho", beam_size=beam, taps=otaps, trace=otrace)
            tc = time.perf_counter() - tc
            ofe = otaps["stage3"].mean(dim=3).transpose(1, 2).contiguous()   # frame_embs (B, T, 768), convnext.py:306
            omem = otaps["memory"]
            otaps.clear()
            result["cpu_baseline"] = {
                "value": round(nc / tc, 4), "unit": "clips/s", "cores": n_thr, "kind": "port",
                "sample": f"{nc} of the same synthetic clips, same checkpoint, beam {beam}: oracle/cpu_ref.py "
                          f"(stock PyTorch fp32, reference algorithm incl. its no-KV-cache decode), {tc:.1f} s"}
            if args.workload == "fixed" and args.parity_clips > 0:   # (untimed) greedy = beam 1 from the oracle's own memory
                msk = torch.zeros((nc, omem.shape[-1]), dtype=torch.bool)
                g1 = O.generate(sd, omem, msk, bos_all[torch.zeros(nc, dtype=torch.long)], vocab_size=eng.vocab_size,
                                beam_size=1, min_pred_size=min_pred, max_pred_size=max_pred,
                                forbid_rep_mask=sd["model.forbid_rep_mask"], trace=gtrace)
                ora = {"n": nc, "greedy": g1[0], "beam": o3["preds"], "beam_lp": o3["lprobs"],
                       "greedy_margin": clip_min_margins(gtrace, nc), "beam_margin": clip_min_margins(otrace, nc),
                       "frame_embs": ofe}

    # ---- parity of every precision against the ORACLE + attribution of the bf16 disagreement (rank 0, N = 1, untimed) ----
    if rank == 0 and world == 1 and args.parity_clips > 0 and args.workload == "fixed":
        result["parity"] = parity_report(args, Engine, eng, sd, dev, w0, lens0, bos0, forbid, t0_, beam, min_pred, max_pred, ora, result)
    if rank == 0 and world == 1 and args.workload == "fixed" and args.precision == "bf16" and args.also and args.parity_clips > 0:
        result["also_pipelined"] = also_pipelined(args, B0)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

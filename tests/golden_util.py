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

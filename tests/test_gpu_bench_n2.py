"""GPU: the N > 1 code path of bench.py on a one-GPU box -- two ranks sharing cuda:0 (CN_BENCH_SHARE_GPU=1: gloo collectives
through host memory; RCCL refuses two ranks on one device).  Not a measurement: it checks that the sharded run completes,
that every rank's pipelined steps equal its solo pass (the consistency vote is an all-reduce), and that the one JSON line
carries the aggregate and the per-rank rates the scaling run will be read by."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CN_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--repeat", "2", "--warmup", "3",
                        "--cpu-clips", "0", "--parity-clips", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_weak():
    d = _run(["--batch", "16"])
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["config"]["global_batch"] == 32
    assert d["scaling"] == "weak" and d["pipeline_consistent"] is True and d["pipeline_steps_checked"] == 12
    rr = d["rank_clips_per_sec"]
    assert len(rr["per_rank"]) == 2 and rr["min"] > 0
    assert abs(d["value"] - 32 * 6 / d["timed_region_s"]) < 1e-2 * d["value"]          # whole-job aggregate over the median window
    assert len(d["windows"]["clips_per_sec"]) == 2


def test_two_ranks_strong_ragged():
    d = _run(["--global-batch", "21"])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 21 and d["config"]["batch_per_gpu"] == 11
    assert d["pipeline_consistent"] is True and d["value"] > 0

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


def _run(extra, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CN_BENCH_SHARE_GPU"] = "1"
    env["CN_MIN_TIMED_S"] = "0"     # exactly --repeat windows (bench.py raises the count until the timed region fills 1 s otherwise)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--repeat", "2", "--warmup", "3",
                        "--cpu-clips", "0", "--parity-clips", "0", "--also", ""] + extra, env=env, capture_output=True, text=True, timeout=600)
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
    assert d["gather"]["collectives_per_window"] == 2 and d["gather"]["consistent_across_ranks"] is True


def test_two_ranks_strong_ragged():
    d = _run(["--global-batch", "21"])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 21 and d["config"]["batch_per_gpu"] == 11
    assert d["pipeline_consistent"] is True and d["value"] > 0


def test_sharded_captions_equal_the_single_rank_run():
    """The job's product -- the all-gathered ids of every clip, in clip order -- is the same table whether 32 clips run on one
    rank or as two shards of 16 (clip i is generated from seed 1234 + i either way): sha256 over the trimmed id matrix of the
    last timed step, which every rank of the sharded run must also agree on (`gather.consistent_across_ranks`)."""
    two = _run(["--global-batch", "32"])
    one = _run(["--batch", "32"], gpus=1)
    assert two["config"]["global_batch"] == one["config"]["global_batch"] == 32
    assert two["gather"]["consistent_across_ranks"] is True and one["gather"] is None
    assert two["captions_sha256"] == one["captions_sha256"]

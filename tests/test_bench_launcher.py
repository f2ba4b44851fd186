"""CPU: `python bench.py --gpus 2` from a plain shell starts its own ranks (the path the driver's scaling run takes when it
does not wrap the command in torch.distributed.run), here on the gloo self-test leg: rendezvous on 127.0.0.1, shard
bounds, the all-gather of ids + scores, one JSON line from rank 0 with the world size it observed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo"] + extra, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_weak():
    d = _run(["--gpus", "2", "--batch", "5"])
    assert d["n_gpus"] == 2 and d["world_size_observed"] == 2 and d["ok"] and d["scaling"] == "weak"
    assert d["global_batch"] == 10 and d["shard"] == [0, 5]


def test_self_launch_two_ranks_strong_ragged():
    d = _run(["--gpus", "2", "--global-batch", "7"])
    assert d["n_gpus"] == 2 and d["ok"] and d["scaling"] == "strong" and d["global_batch"] == 7 and d["shard"] == [0, 4]


def test_self_launch_eight_ranks_config4_shards():
    """BASELINE config 4: 2048 clips over 8 ranks (shards of 256), and a ragged 2047: every rank's shard bounds, the gathered
    order of ids + scores (rank r's clips land at rows shard_bounds(G, r, 8)) and the trimmed width are checked on all
    ranks (gloo; the RCCL run itself needs the 8-GPU node)."""
    d = _run(["--gpus", "8", "--global-batch", "2048"])
    assert d["n_gpus"] == 8 and d["world_size_observed"] == 8 and d["ok"] and d["scaling"] == "strong"
    assert d["global_batch"] == 2048 and d["shard"] == [0, 256]
    d = _run(["--gpus", "8", "--global-batch", "2047"])
    assert d["ok"] and d["global_batch"] == 2047 and d["shard"] == [0, 256]
    from conette_amd.dist import shard_bounds
    b = [shard_bounds(2047, r, 8) for r in range(8)]
    assert b[0][0] == 0 and b[-1][1] == 2047 and all(b[i][1] == b[i + 1][0] for i in range(7))
    assert sorted(hi - lo for lo, hi in b) == [255] + [256] * 7


def test_wrong_world_size_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bucket_plan():
    sys.path.insert(0, ROOT)
    import conette_amd  # noqa: F401
    from conette_amd.bucketing import padding_waste, plan_buckets
    import random
    rng = random.Random(0)
    lengths = [rng.randint(32000, 960000) for _ in range(200)]
    buckets = plan_buckets(lengths, max_padded_seconds=640.0)
    flat = sorted(i for b in buckets for i in b)
    assert flat == list(range(200))                                   # a partition
    for b in buckets:
        assert len(b) * max(lengths[i] for i in b) <= 640 * 32000 or len(b) == 1
    firsts = [max(lengths[i] for i in b) for b in buckets]
    assert firsts == sorted(firsts)                                   # shortest clips first
    assert padding_waste(lengths, buckets) < 0.15                     # one pad-to-max batch of these clips wastes ~50 %
    assert padding_waste(lengths, [list(range(200))]) > 0.4
    assert plan_buckets([5], 1.0) == [[0]] and plan_buckets([], 1.0) == []


def test_bucket_plan_by_cost_is_optimal():
    """plan_buckets_by_cost: a partition of the sorted clips, shortest first, whose modelled cost (padded audio seconds + a fixed cost
    per bucket) equals the brute-force minimum over all contiguous partitions on small inputs, and beats the budget planner's."""
    sys.path.insert(0, ROOT)
    import itertools
    import random

    import conette_amd  # noqa: F401
    from conette_amd.bucketing import plan_buckets, plan_buckets_by_cost

    def cost(lengths, buckets, fixed):
        return sum(len(b) * max(lengths[i] for i in b) / 32000.0 + fixed for b in buckets)

    rng = random.Random(3)
    for n in (1, 2, 5, 9):
        lengths = [rng.randint(32000, 960000) for _ in range(n)]
        for fixed in (0.0, 20.0, 150.0, 1e6):
            got = plan_buckets_by_cost(lengths, fixed)
            assert sorted(i for b in got for i in b) == list(range(n))
            order = sorted(range(n), key=lambda i: (lengths[i], i))
            best = min(cost(lengths, [order[a:b] for a, b in zip((0,) + cuts, cuts + (n,))], fixed)
                       for k in range(n) for cuts in itertools.combinations(range(1, n), k))
            assert abs(cost(lengths, got, fixed) - best) < 1e-6, (n, fixed)
        assert len(plan_buckets_by_cost(lengths, 1e6)) == 1 and len(plan_buckets_by_cost(lengths, 0.0)) == len(set(lengths))
    lengths = [rng.randint(32000, 960000) for _ in range(200)]
    a, b = plan_buckets_by_cost(lengths, 150.0), plan_buckets(lengths, 960.0)
    assert cost(lengths, a, 150.0) <= cost(lengths, b, 150.0)
    firsts = [max(lengths[i] for i in bk) for bk in a]
    assert firsts == sorted(firsts)
    assert plan_buckets_by_cost([], 150.0) == []


def test_decode_group_choice():
    """bench.py groups the beam searches of G consecutive batches into one chain: G divides --steps, G x B <= 256 clips."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.choose_decode_group(20, 64, 3) == 4 and bench.choose_decode_group(100, 64, 3) == 4
    assert bench.choose_decode_group(30, 64, 3) == 3 and bench.choose_decode_group(50, 64, 3) == 2 and bench.choose_decode_group(7, 64, 3) == 1
    assert bench.choose_decode_group(20, 256, 3) == 1 and bench.choose_decode_group(20, 128, 3) == 2
    assert bench.choose_decode_group(64, 16, 3) == 16 and bench.choose_decode_group(200, 16, 3) == 10
    assert bench.choose_decode_group(20, 64, 3, cap=1) == 1 and bench.choose_decode_group(20, 64, 3, cap=2) == 2
    assert bench.choose_decode_group(20, 64, 8) == 4 and bench.choose_decode_group(16, 32, 8) == 8
    for steps in range(1, 40):
        for b in (1, 11, 16, 64, 100, 256, 1000):
            g = bench.choose_decode_group(steps, b, 3)
            assert steps % g == 0 and (g == 1 or g * b <= 256) and g * b * 3 < 4096 or g == 1

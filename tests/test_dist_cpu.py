"""CPU: the N > 1 path (clip sharding + final all-gather of ids / scores) with world_size 2 on gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_total: int, ret) -> None:
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import conette_amd  # noqa: F401
    from conette_amd.dist import gather_captions, shard_bounds, trim_captions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    full = torch.randint(4, 100, (n_total, 20), generator=g, dtype=torch.int32)
    full[:, 9] = 2
    full[:, 10:] = 0
    scores = torch.arange(n_total, dtype=torch.float32) * -0.5
    lo, hi = shard_bounds(n_total, rank, world)
    p, l = gather_captions(full[lo:hi].clone(), scores[lo:hi].clone(), n_total)
    ok = torch.equal(p, full) and torch.equal(l, scores) and trim_captions(p).shape[1] == 10
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    import conette_amd  # noqa: F401
    from conette_amd.dist import shard_bounds
    for n in (1, 7, 64, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_gather_captions_world2_gloo():
    for n_total in (8, 7):  # even and ragged shards
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]

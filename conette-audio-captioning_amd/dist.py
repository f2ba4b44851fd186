"""Multi-GPU sharding of the hot path.

Clips are independent through every stage, so a batch is cut into contiguous shards, one
process per GPU, each with a full weight replica; the only exchange is one all-gather of the
padded token-id matrix and the scores at the end (RCCL over xGMI on MI355X: backend "nccl";
"gloo" on CPU for the tests).  The reference has no counterpart (single device,
huggingface/model.py:103-104).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/world) items per rank (the last ranks may be short or empty)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def gather_captions(preds: torch.Tensor, lprobs: torch.Tensor, n_total: int, pad_id: int = 0,
                    group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """all-gather the per-shard (b_local, max_pred) ids and (b_local,) scores.

    Shards are padded to ceil(n_total / world) rows so that one fixed-size collective serves
    every rank; returns the first n_total rows in clip order on every rank."""
    if not dist.is_available() or not dist.is_initialized():
        return preds[:n_total], lprobs[:n_total]
    world = dist.get_world_size(group)
    per = (n_total + world - 1) // world
    width = preds.shape[1]
    p = torch.full((per, width), pad_id, dtype=preds.dtype, device=preds.device)
    l = torch.zeros((per,), dtype=lprobs.dtype, device=lprobs.device)
    p[: preds.shape[0]] = preds
    l[: lprobs.shape[0]] = lprobs
    dev = preds.device
    if dist.get_backend(group) == "gloo" and dev.type != "cpu":  # (gloo moves host memory: the N > 1 self-test on one GPU)
        p, l = p.cpu(), l.cpu()
    all_p = torch.empty((world * per, width), dtype=preds.dtype, device=p.device)
    all_l = torch.empty((world * per,), dtype=lprobs.dtype, device=l.device)
    dist.all_gather_into_tensor(all_p, p, group=group)
    dist.all_gather_into_tensor(all_l, l, group=group)
    return all_p[:n_total].to(dev), all_l[:n_total].to(dev)


def gather_caption_windows(preds: torch.Tensor, lprobs: torch.Tensor, n_total: int, pad_id: int = 0,
                           group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE all-gather for a whole window of steps: per-shard (K, b_local, max_pred) ids and (K, b_local) scores of K
    consecutive passes -> (K, n_total, max_pred), (K, n_total) in clip order on every rank.

    The pipelined loop of bench.py calls this once per timed window instead of gather_captions once per step: a collective
    is a rendezvous of all ranks, and one per step would serialise every rank's step-to-step jitter into every step."""
    if not dist.is_available() or not dist.is_initialized():
        return preds[:, :n_total], lprobs[:, :n_total]
    world = dist.get_world_size(group)
    per = (n_total + world - 1) // world
    k, b, width = preds.shape
    p = torch.full((k, per, width), pad_id, dtype=preds.dtype, device=preds.device)
    l = torch.zeros((k, per), dtype=lprobs.dtype, device=lprobs.device)
    p[:, :b] = preds
    l[:, :b] = lprobs
    dev = preds.device
    if dist.get_backend(group) == "gloo" and dev.type != "cpu":  # (gloo moves host memory: the N > 1 self-test on one GPU)
        p, l = p.cpu(), l.cpu()
    all_p = torch.empty((world * k, per, width), dtype=preds.dtype, device=p.device)   # (rank-major concatenation)
    all_l = torch.empty((world * k, per), dtype=lprobs.dtype, device=l.device)
    dist.all_gather_into_tensor(all_p, p.contiguous(), group=group)
    dist.all_gather_into_tensor(all_l, l.contiguous(), group=group)
    all_p = all_p.view(world, k, per, width).permute(1, 0, 2, 3).reshape(k, world * per, width)
    all_l = all_l.view(world, k, per).permute(1, 0, 2).reshape(k, world * per)
    return all_p[:, :n_total].to(dev), all_l[:, :n_total].to(dev)


def trim_captions(preds: torch.Tensor, eos_id: int = 2) -> torch.Tensor:
    """Cut trailing pad columns after gathering: longest (first EOS index) + 1 (beam.py:222-225)."""
    has = preds == eos_id
    first = torch.where(has.any(dim=1), has.int().argmax(dim=1), torch.full((preds.shape[0],), preds.shape[1],
                                                                              device=preds.device))
    return preds[:, : int(first.max().item()) + 1] if preds.numel() else preds

"""Length-bucketed batching for mixed-length clip collections (BASELINE config 5, WavCaps-style 1-30 s clips).

The reference has exactly one way of batching clips of different lengths: zero-pad every clip to the longest one of the
batch (nn/functional/pad.py:11-17 via huggingface/preprocessor.py:143-154) and derive each clip's number of valid audio
frames from the padded length, ``round(len / (Lmax // T))`` (nn/encoders/convnext.py:312-315).  A clip's caption
therefore depends on the batch it is padded in (the ConvNeXt convolutions see the zero tail, and T depends on Lmax).
Bucketing keeps that definition and only chooses the batches: clips are sorted by length and cut into buckets whose
padded cost ``n_clips x longest`` stays under a budget, each bucket is ONE reference-style batch, and the per-clip
results are scattered back to input order.  The contract (tests/test_gpu_bucketing.py): a clip's outputs are identical
to the reference run on its bucket.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch


def plan_buckets(lengths: Sequence[int], max_padded_seconds: float = 640.0, sr: int = 32000,
                 max_clips: int = 256) -> List[List[int]]:
    """Indices of ``lengths`` grouped into buckets, shortest clips first.

    A bucket is closed when adding the next (longer) clip would push its padded cost ``(n + 1) * len_next`` samples
    over ``max_padded_seconds * sr`` or its size over ``max_clips``; every bucket holds at least one clip.  640 padded
    seconds is the footprint of the benchmark batch (64 clips x 10 s)."""
    order = sorted(range(len(lengths)), key=lambda i: (int(lengths[i]), i))
    budget = int(max_padded_seconds * sr)
    buckets: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        if cur and ((len(cur) + 1) * int(lengths[i]) > budget or len(cur) >= max_clips):
            buckets.append(cur)
            cur = []
        cur.append(i)
    if cur:
        buckets.append(cur)
    return buckets


def plan_buckets_by_cost(lengths: Sequence[int], fixed_cost_seconds: float = 150.0, sr: int = 32000,
                         max_clips: int = 256, max_padded_seconds: float = 4000.0) -> List[List[int]]:
    """The partition of the length-sorted clips into contiguous buckets that minimises the MODELLED cost of the whole
    collection (plan by cost, not by a padded-seconds budget):

        cost(bucket) = n_clips x longest clip  +  fixed_cost_seconds        (both in audio seconds)

    The first term is what the encoder works on (it is linear in the padded samples); the second is what a bucket costs
    whatever it holds -- its share of a beam-search chain (a latency chain of ~300 launches, 3.7 ms at any batch size up to
    64: 1.2 ms when three chains run side by side) plus the ~60 encoder launches of a batch that is too small to fill the
    chip -- expressed in the audio seconds the encoder processes in that time (104.6 k audio-s/s on one MI355X: 1.4 ms is
    150 s).  Exact by dynamic programming over the sorted order (an optimal partition of this cost is contiguous in sorted
    order: swapping two clips between buckets never lowers a maximum).  Shortest clips first, like plan_buckets; a bucket
    is still ONE reference-style pad-to-max batch, so the per-bucket contract of this module is unchanged."""
    order = sorted(range(len(lengths)), key=lambda i: (int(lengths[i]), i))
    n = len(order)
    if n == 0:
        return []
    ls = [int(lengths[i]) / float(sr) for i in order]
    INF = float("inf")
    best = [0.0] + [INF] * n     # best[j]: cost of the first j clips
    cut = [0] * (n + 1)
    for j in range(1, n + 1):
        for i in range(max(0, j - max_clips), j):   # bucket = clips i .. j - 1, longest = ls[j - 1]
            padded = (j - i) * ls[j - 1]
            if padded > max_padded_seconds and j - i > 1:
                continue
            c = best[i] + padded + fixed_cost_seconds
            if c < best[j]:
                best[j], cut[j] = c, i
    buckets: List[List[int]] = []
    j = n
    while j > 0:
        buckets.append(order[cut[j]: j])
        j = cut[j]
    return buckets[::-1]


def padding_waste(lengths: Sequence[int], buckets: Sequence[Sequence[int]]) -> float:
    """Share of the padded samples that are padding (0 = every bucket is uniform)."""
    padded = sum(len(b) * max(int(lengths[i]) for i in b) for b in buckets)
    real = sum(int(v) for v in lengths)
    return 1.0 - real / max(padded, 1)


@torch.no_grad()
def caption_bucketed(model, audios: Sequence[torch.Tensor], sr: int = 32000, task: Optional[str] = None,
                     max_padded_seconds: float = 640.0, max_clips: int = 256, fixed_cost_seconds: Optional[float] = None,
                     **kwargs: Any) -> Dict[str, Any]:
    """``model(list_of_clips, ...)`` bucket by bucket; ``audios[i]``: (channels, samples) or (samples,) tensors of one
    sample rate.  Returns the model's output dict in input order (``preds`` / ``mult_preds`` right-padded with the pad id
    to the widest bucket) plus ``"buckets"``: the index lists that were batched together.  ``fixed_cost_seconds`` selects the
    cost-model planner (plan_buckets_by_cost) instead of the padded-seconds budget."""
    clips = [a if a.ndim == 2 else a[None] for a in audios]
    lengths = [int(c.shape[-1]) for c in clips]
    # the budget counts samples at 32 kHz: scale it for other input rates
    if fixed_cost_seconds is not None:   # minimum modelled cost (plan_buckets_by_cost) instead of a padded-seconds budget
        buckets = plan_buckets_by_cost(lengths, fixed_cost_seconds, sr, max_clips)
    else:
        buckets = plan_buckets(lengths, max_padded_seconds * sr / 32000.0, 32000, max_clips)
    n = len(clips)
    per_clip: List[Optional[Dict[str, Any]]] = [None] * n
    keys: List[str] = []
    for idx in buckets:
        t = task if task is None or isinstance(task, str) else [task[i] for i in idx]
        out = model([clips[i] for i in idx], sr=sr, task=t, **kwargs)
        keys = list(out.keys())
        for j, i in enumerate(idx):
            per_clip[i] = {k: out[k][j] for k in keys}
    res: Dict[str, Any] = {"buckets": buckets}
    for k in keys:
        vals = [pc[k] for pc in per_clip]  # type: ignore[index]
        if isinstance(vals[0], torch.Tensor):
            if vals[0].ndim == 0:
                res[k] = torch.stack(vals)
            else:  # id matrices: pad the last dim to the widest bucket
                w = max(int(v.shape[-1]) for v in vals)
                res[k] = torch.stack([torch.nn.functional.pad(v, (0, w - int(v.shape[-1]))) for v in vals])
        else:
            res[k] = vals
    return res

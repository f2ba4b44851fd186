"""Offline feature producer: the on-disk side of the reference's
``get_resample_mean_convnext`` transform (transforms/get.py:240-310), which the training stack packs into HDF files
with the columns ``audio`` (T, 768 frame embeddings), ``audio_shape`` and, with ``only_frame_embs=False``, the clip-level
outputs (conf/dm/hdf.yaml:12-14); ``model(..., preprocess=False)`` (huggingface/model.py:205-212) is the consumer.

Semantics kept from the reference: every file is encoded ALONE (batch of one: resample -> channel mean -> ConvNeXt, so the
features do not depend on what else is in the collection) and batches are formed later by zero-padding the stored
features along time (``audio_padding: "batch"``).  h5py is not available offline, so one ``.npz`` per file stands in for
the HDF row; the column names are the reference's.
"""
from __future__ import annotations

import json
import os
import os.path as osp
from typing import Any, Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np
import torch

COLUMNS = ("audio", "audio_shape", "clip_probs")


def transform_one(model, waveform: Union[str, torch.Tensor], sr: int = 32000) -> Dict[str, torch.Tensor]:
    """One file / one (channels, samples) waveform -> {"audio": (T, 768), "audio_shape": (2,), "clip_probs": (527,)}."""
    if isinstance(waveform, str):
        batch = model.preprocessor(waveform, None, None)
    else:
        wav = waveform if waveform.ndim == 2 else waveform[None]
        batch = model.preprocessor([wav], [int(sr)], None)
    return {"audio": batch["audio"][0], "audio_shape": batch["audio_shape"][0], "clip_probs": batch["clip_probs"][0]}


def write_features(model, files: Sequence[str], out_dir: str) -> List[str]:
    """Encode every audio file on its own and store ``<stem>.npz`` (columns audio, audio_shape, clip_probs, fname) plus
    ``index.json`` (the file order).  Returns the feature paths."""
    os.makedirs(out_dir, exist_ok=True)
    out: List[str] = []
    for f in files:
        feats = transform_one(model, f)
        p = osp.join(out_dir, osp.splitext(osp.basename(f))[0] + ".npz")
        np.savez(p, audio=feats["audio"].float().cpu().numpy(), audio_shape=feats["audio_shape"].cpu().numpy().astype(np.int64),
                 clip_probs=feats["clip_probs"].float().cpu().numpy(), fname=osp.basename(f))
        out.append(p)
    with open(osp.join(out_dir, "index.json"), "w") as fh:
        json.dump([osp.basename(p) for p in out], fh)
    return out


def load_features(paths: Iterable[str]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stored rows -> (audio (B, Tmax, 768) zero-padded along time, audio_shape (B, 2)) for ``model(audio,
    x_shapes=audio_shape, preprocess=False)``."""
    rows = [np.load(p) for p in paths]
    tmax = max(int(r["audio"].shape[0]) for r in rows)
    audio = torch.zeros((len(rows), tmax, 768), dtype=torch.float32)
    for i, r in enumerate(rows):
        audio[i, : r["audio"].shape[0]] = torch.from_numpy(r["audio"])
    shapes = torch.as_tensor(np.stack([r["audio_shape"] for r in rows]))
    return audio, shapes

"""Host side of the online audio preprocessing (row a1 / a7 of SURVEY.md section 8a).

Mirrors ``CoNeTTEPreprocessor`` (reference huggingface/preprocessor.py:21-154): accepts a path,
list of paths, Tensor (T,), (C,T), (B,C,T) or list of (C,T) tensors, resamples to 32 kHz when
needed (HIP resampler), takes the channel mean, zero-pads to the batch maximum and runs the
HIP encoder.  The reference's torchaudio.load is replaced by a PCM WAV reader (stdlib ``wave``).
"""
from __future__ import annotations

import wave as _wave
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Size, Tensor

from .engine import Engine

TARGET_SR = 32_000
FEAT_SIZE = 768


def load_audio(path: str) -> Tuple[Tensor, int]:
    """PCM WAV -> ((C, L) float32 in [-1, 1), sample rate)."""
    with _wave.open(path, "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"Unsupported PCM sample width {width} in {path}.")
    return torch.from_numpy(np.ascontiguousarray(data.reshape(-1, nch).T)), sr


def _is_iterable_str(x: Any) -> bool:
    return isinstance(x, str) or (not isinstance(x, Tensor) and isinstance(x, Iterable)
                                  and all(isinstance(v, str) for v in x))


class CoNeTTEPreprocessor:
    def __init__(self, engine: Engine, verbose: int = 0) -> None:
        self.engine = engine
        self.verbose = verbose

    @property
    def device(self) -> torch.device:
        return self.engine.device

    @property
    def target_sr(self) -> int:
        return TARGET_SR

    @property
    def feat_size(self) -> int:
        return FEAT_SIZE

    def __call__(self, x, sr=None, x_shapes=None) -> Dict[str, Any]:
        return self.forward(x, sr, x_shapes)

    def forward(self, x: Union[Tensor, str, Iterable[str], Iterable[Tensor]],
                sr: Union[None, int, Iterable[int]] = None,
                x_shapes: Union[Tensor, None, List[Size]] = None) -> Dict[str, Any]:
        """preprocessor.py:50-77."""
        wave, shapes = self._load_resample(x, sr, x_shapes)
        frame_embs, clip_probs = self.engine.encode(wave)
        lens = frame_embs_lens(shapes[:, -1], wave.shape[-1], frame_embs.shape[1])
        audio_shape = torch.stack([torch.full_like(lens, self.feat_size), lens], dim=1).to(self.device)
        return {"audio": frame_embs, "audio_shape": audio_shape, "clip_probs": clip_probs}

    def _load_resample(self, x, sr=None, x_shapes=None) -> Tuple[Tensor, Tensor]:
        """preprocessor.py:82-154 (same accepted forms, same error behaviour)."""
        if _is_iterable_str(x):
            if isinstance(x, str):
                x = [x]
            loaded = [load_audio(p) for p in x]
            x = [a for a, _ in loaded]
            sr = [s for _, s in loaded]
        else:
            if isinstance(x, Tensor):
                if x.ndim == 1:
                    x = x.unsqueeze(dim=0).unsqueeze(dim=1)
                elif x.ndim == 2:
                    x = x.unsqueeze(dim=0)
                elif x.ndim == 3:
                    pass
                else:
                    raise ValueError(f"Invalid argument shape {x.shape=}.")
            else:
                x = list(x)
            if isinstance(sr, int):
                sr = [sr]
            elif sr is None:
                sr = [self.target_sr]
            else:
                sr = list(sr)
        if len(sr) == 1 and len(x) != len(sr):
            sr = sr * len(x)
        assert len(x) == len(sr) and len(x) > 0
        dev = self.device
        if isinstance(x, Tensor):
            x = x.to(device=dev, dtype=torch.float32)
        else:
            x = [xi.to(device=dev, dtype=torch.float32) for xi in x]
        if any(sri != self.target_sr for sri in sr):
            if x_shapes is not None:
                raise ValueError(f"Invalid argument {x_shapes=}.")
            if all(s == sr[0] for s in sr) and isinstance(x, Tensor):
                x = self.engine.resample(x, sr[0], self.target_sr)
            else:
                x = [self.engine.resample(xi, sri, self.target_sr) for xi, sri in zip(x, sr)]
        if isinstance(x, Tensor):
            x = x.mean(dim=1)
        else:
            x = [xi.mean(dim=0) for xi in x]
        if x_shapes is None:
            x_shapes = [list(xi.shape) for xi in x]
        x_shapes = torch.as_tensor(x_shapes)
        if not isinstance(x, Tensor):  # pad_and_stack (nn/functional/pad.py:11-17)
            max_len = max(xi.shape[-1] for xi in x)
            out = torch.zeros((len(x), max_len), dtype=torch.float32, device=dev)
            for i, xi in enumerate(x):
                out[i, : xi.shape[-1]] = xi
            x = out
        return x.contiguous(), x_shapes


def frame_embs_lens(input_lens: Tensor, padded_len: int, n_frames: int) -> Tensor:
    """convnext.py:312-315: round(len / (Lmax // T)) with torch fp32 round-half-even, as int32."""
    reduction_factor = int(padded_len) // int(n_frames)
    return input_lens.to("cpu").div(reduction_factor).round().int()

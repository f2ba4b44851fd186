"""Host side of the online audio preprocessing (row a1 / a7 of SURVEY.md section 8a).

Mirrors ``CoNeTTEPreprocessor`` (reference huggingface/preprocessor.py:21-154): accepts a path,
list of paths, Tensor (T,), (C,T), (B,C,T) or list of (C,T) tensors, resamples to 32 kHz when
needed (HIP resampler), takes the channel mean, zero-pads to the batch maximum and runs the
HIP encoder.  The reference's torchaudio.load is replaced by a RIFF/WAVE reader (PCM and float formats) with
soundfile / torchaudio as optional back-ends for other containers.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Size, Tensor

from .engine import Engine

TARGET_SR = 32_000
FEAT_SIZE = 768


def _read_wav(path: str) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE reader: PCM 8 / 16 / 24 / 32 bit, IEEE float 32 / 64 bit, plain or WAVE_FORMAT_EXTENSIBLE headers
    (the stdlib ``wave`` module refuses everything but 8 / 16 / 32-bit PCM).  -> ((frames, channels) float32, sr)."""
    import struct
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path} is not a RIFF/WAVE file.")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = blob[pos : pos + 4], struct.unpack("<I", blob[pos + 4 : pos + 8])[0]
        body = blob[pos + 8 : pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt / data chunk.")
    tag, nch, sr, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the real tag is the first word of the sub-format GUID
        tag = struct.unpack("<H", fmt[24:26])[0]
    n = len(data) // (bits // 8) // max(nch, 1) * nch
    if tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2", count=n).astype(np.float32) / 32768.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(data, dtype=np.uint8, count=n * 3).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(data, dtype="<i4", count=n).astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(data, dtype=np.uint8, count=n).astype(np.float32) - 128.0) / 128.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(data, dtype="<f4", count=n).astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(data, dtype="<f8", count=n).astype(np.float32)
    else:
        raise ValueError(f"Unsupported WAVE format tag {tag} with {bits} bits in {path}.")
    return x.reshape(-1, nch), int(sr)


def load_audio(path: str) -> Tuple[Tensor, int]:
    """Audio file -> ((C, L) float32 in [-1, 1), sample rate); stands in for ``torchaudio.load`` (preprocessor.py:79-80).
    WAV files (PCM 8 / 16 / 24 / 32 bit, float 32 / 64 bit) are read here; any other container (FLAC, OGG, MP3) goes
    through ``soundfile`` or ``torchaudio`` when one of them is installed and is an explicit error otherwise."""
    with open(path, "rb") as f:
        head = f.read(12)
    if head[:4] == b"RIFF" and head[8:12] == b"WAVE":
        data, sr = _read_wav(path)
        return torch.from_numpy(np.ascontiguousarray(data.T)), sr
    try:
        import soundfile
        data, sr = soundfile.read(path, dtype="float32", always_2d=True)
        return torch.from_numpy(np.ascontiguousarray(data.T)), int(sr)
    except ImportError:
        pass
    try:
        import torchaudio
        wav, sr = torchaudio.load(path)
        return wav.to(torch.float32), int(sr)
    except ImportError:
        raise ValueError(f"{path}: only WAV files can be read without soundfile / torchaudio installed.") from None


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
        out = {"audio": frame_embs, "audio_shape": audio_shape, "clip_probs": clip_probs}
        if getattr(self.engine, "certified", False):
            out["_wave"] = wave   # the padded batch: what the exact context re-encodes for clips the margins do not certify
        return out

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

"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the third-party arithmetic the reference's hot path calls but that is
NOT under /root/reference (pinned in /root/reference/requirements.txt:3-14):

* torchlibrosa==0.1.0  ``stft.Spectrogram`` / ``stft.LogmelFilterBank``
  (call sites: src/conette/nn/encoders/convnext.py:11,160-180,276-278)
* librosa ``filters.mel`` (Slaney scale / slaney norm), used by LogmelFilterBank
* torchaudio==0.13.1 ``functional.resample`` (sinc_interpolation, width 6, rolloff 0.99)
  (call sites: src/conette/huggingface/preprocessor.py:8-10,134-141)
* torchoutil~=0.3.0 leaf helpers (call sites: src/conette/nn/decoding/beam.py:10-15,
  src/conette/pl_modules/conette.py:9-13, src/conette/nn/functional/pad.py:8)

PARITY UNPINNED for these leaves: none of the packages is importable here and the reference
holds no golden vector for them; the published algorithms are restated.
The DFT / mel matrices are persisted tensors of the real checkpoint
(``preprocessor.encoder.spectrogram_extractor.stft.conv_{real,imag}.weight``,
``preprocessor.encoder.logmel_extractor.melW``), so on real weights only the *application*
of those matrices (conv1d / matmul / log10) matters, which is plain torch.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F

# ----------------------------------------------------------------------------------------
# librosa.filters.mel  (Slaney mel scale, norm="slaney", htk=False)
# ----------------------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    mels = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    out = np.array(mels, dtype=np.float64)
    out[log_t] = _MIN_LOG_MEL + np.log(f[log_t] / _MIN_LOG_HZ) / _LOGSTEP
    return out


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    freqs = _F_SP * m
    log_t = m >= _MIN_LOG_MEL
    out = np.array(freqs, dtype=np.float64)
    out[log_t] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m[log_t] - _MIN_LOG_MEL))
    return out


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) -> (n_mels, 1 + n_fft//2) float32."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sr) / 2, n_bins, endpoint=True)
    mels = np.linspace(hz_to_mel(np.array([fmin]))[0], hz_to_mel(np.array([fmax]))[0], n_mels + 2)
    mel_f = mel_to_hz(mels)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def dft_conv_kernels(n_fft: int = 1024) -> tuple[np.ndarray, np.ndarray]:
    """torchlibrosa STFT kernels: W[k, n] = exp(-2*pi*i*k*n/n_fft) * hann[n], k = 0..n_fft/2.

    Returns (conv_real.weight, conv_imag.weight), each (n_fft/2+1, 1, n_fft) float32.
    """
    n = np.arange(n_fft, dtype=np.float64)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)
    # reduce the angle modulo n_fft in integers first: exact for every (k, n)
    kn = (np.outer(k, n).astype(np.int64) % n_fft).astype(np.float64)
    ang = -2.0 * np.pi * kn / n_fft
    win = hann_periodic(n_fft)
    real = (np.cos(ang) * win[None, :]).astype(np.float32)
    imag = (np.sin(ang) * win[None, :]).astype(np.float32)
    return real[:, None, :], imag[:, None, :]


class STFT(nn.Module):
    """torchlibrosa.stft.STFT: DFT as two strided conv1d over the reflect-padded waveform."""

    def __init__(self, n_fft=1024, hop_length=320, win_length=1024, window="hann",
                 center=True, pad_mode="reflect", freeze_parameters=True) -> None:
        super().__init__()
        assert window == "hann" and win_length == n_fft and pad_mode == "reflect" and center
        self.n_fft = n_fft
        self.hop_length = hop_length
        out_channels = n_fft // 2 + 1
        self.conv_real = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, bias=False)
        self.conv_imag = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, bias=False)
        real, imag = dft_conv_kernels(n_fft)
        self.conv_real.weight.data = torch.from_numpy(real.copy())
        self.conv_imag.weight.data = torch.from_numpy(imag.copy())
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input: Tensor):
        x = input[:, None, :]
        x = F.pad(x, pad=(self.n_fft // 2, self.n_fft // 2), mode="reflect")
        real = self.conv_real(x)
        imag = self.conv_imag(x)
        real = real[:, None, :, :].transpose(2, 3)
        imag = imag[:, None, :, :].transpose(2, 3)
        return real, imag


class Spectrogram(nn.Module):
    """torchlibrosa.stft.Spectrogram (power=2.0): real**2 + imag**2 -> (B, 1, T, F)."""

    def __init__(self, n_fft=1024, hop_length=320, win_length=1024, window="hann",
                 center=True, pad_mode="reflect", power=2.0, freeze_parameters=True) -> None:
        super().__init__()
        self.power = power
        self.stft = STFT(n_fft, hop_length, win_length, window, center, pad_mode, freeze_parameters)

    def forward(self, input: Tensor) -> Tensor:
        real, imag = self.stft(input)
        spectrogram = real ** 2 + imag ** 2
        if self.power != 2.0:
            spectrogram = spectrogram ** (self.power / 2.0)
        return spectrogram


class LogmelFilterBank(nn.Module):
    """torchlibrosa.stft.LogmelFilterBank: matmul(melW) then power_to_db (ref, amin, top_db)."""

    def __init__(self, sr=32000, n_fft=1024, n_mels=224, fmin=50, fmax=14000, is_log=True,
                 ref=1.0, amin=1e-10, top_db=None, freeze_parameters=True) -> None:
        super().__init__()
        self.is_log, self.ref, self.amin, self.top_db = is_log, ref, amin, top_db
        melW = mel_filterbank(sr, n_fft, n_mels, fmin, fmax).T
        self.melW = nn.Parameter(torch.from_numpy(np.ascontiguousarray(melW)))
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input: Tensor) -> Tensor:
        mel = torch.matmul(input, self.melW)
        if not self.is_log:
            return mel
        log_spec = 10.0 * torch.log10(torch.clamp(mel, min=self.amin, max=np.inf))
        log_spec -= 10.0 * np.log10(np.maximum(self.amin, self.ref))
        if self.top_db is not None:
            log_spec = torch.clamp(log_spec, min=log_spec.max().item() - self.top_db, max=np.inf)
        return log_spec


class SpecAugmentation(nn.Module):
    """Train-only in the reference (convnext.py:296-297); identity stand-in."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__()

    def forward(self, x: Tensor) -> Tensor:
        return x


# ----------------------------------------------------------------------------------------
# torchaudio.functional.resample (0.13.1 defaults)
# ----------------------------------------------------------------------------------------
def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99, dtype=torch.float64) -> tuple[Tensor, int]:
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base_freq)
    idx = torch.arange(-width, width + o, dtype=dtype)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=dtype)[:, None, None] / n + idx
    t = t * base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / o
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=dtype), t.sin() / t)
    kernels = kernels * window * scale
    return kernels, width


def resample(waveform: Tensor, orig_freq: int, new_freq: int) -> Tensor:
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    # torchaudio 0.13.1 builds the kernel in the waveform's dtype (functional.py resample())
    kernel, width = sinc_resample_kernel(orig_freq, new_freq, dtype=waveform.dtype)
    kernel = kernel.to(device=waveform.device)
    shape = waveform.size()
    wave = waveform.reshape(-1, shape[-1])
    num_wavs, length = wave.shape
    wave = F.pad(wave, (width, width + o))
    resampled = F.conv1d(wave[:, None], kernel, stride=o)
    resampled = resampled.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(n * length / o))
    resampled = resampled[..., :target_length]
    return resampled.reshape(shape[:-1] + resampled.shape[-1:])


# ----------------------------------------------------------------------------------------
# torchoutil leaf helpers
# ----------------------------------------------------------------------------------------
def generate_square_subsequent_mask(size: int, diagonal: int = 0, device=None, dtype=torch.float32) -> Tensor:
    mask = torch.full((size, size), -math.inf, dtype=dtype, device=device)
    return torch.triu(mask, diagonal=diagonal + 1)


def indices_to_multihot(indices: Tensor, num_classes: int, dtype=torch.bool, device=None) -> Tensor:
    out = torch.zeros(tuple(indices.shape[:-1]) + (num_classes,), dtype=torch.bool, device=indices.device)
    out.scatter_(-1, indices.long(), True)
    return out.to(dtype=dtype)


def repeat_interleave_nd(x: Tensor, repeats: int, dim: int = 0) -> Tensor:
    return x.repeat_interleave(repeats, dim=dim)


def tensor_to_lengths(x: Tensor, pad_value=None, end_value=None, dim: int = -1) -> Tensor:
    assert end_value is not None and pad_value is None
    contains = (x == end_value)
    first = contains.int().argmax(dim=dim)
    has = contains.any(dim=dim)
    return torch.where(has, first, torch.full_like(first, x.shape[dim]))


def lengths_to_pad_mask(lengths: Tensor, max_len=None, include: bool = True) -> Tensor:
    if max_len is None:
        max_len = int(lengths.max().item())
    max_len = int(max_len)
    ar = torch.arange(max_len, device=lengths.device)[None, :]
    return ar >= lengths[:, None]


def tensor_to_pad_mask(x: Tensor, pad_value=None, end_value=None) -> Tensor:
    assert pad_value is not None
    return x == pad_value


def pad_dim(x: Tensor, target_length: int, dim: int = -1, pad_value: float = 0.0) -> Tensor:
    missing = max(int(target_length) - x.shape[dim], 0)
    if missing == 0:
        return x
    pad = [0, 0] * x.ndim
    d = dim % x.ndim
    pad[2 * (x.ndim - 1 - d) + 1] = missing
    return F.pad(x, pad, value=pad_value)


def probs_to_names(probs: Tensor, threshold, idx_to_name) -> List[List[str]]:
    mask = probs >= threshold
    return [[idx_to_name[int(j)] for j in torch.where(row)[0].tolist()] for row in mask]


def all_eq(seq: Sequence) -> bool:
    seq = list(seq)
    return all(s == seq[0] for s in seq[1:])


def get_device(device="cuda_if_available"):
    if device == "cuda_if_available" or device == "auto":
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if device is None:
        return None
    return torch.device(device)


class Transpose(nn.Module):
    def __init__(self, dim0: int, dim1: int) -> None:
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1

    def forward(self, x: Tensor) -> Tensor:
        return x.transpose(self.dim0, self.dim1)

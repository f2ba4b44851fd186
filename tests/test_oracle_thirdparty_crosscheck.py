"""CPU: independent cross-checks of ``oracle/thirdparty.py``.

The third-party leaves of the hot path (torchlibrosa's STFT + log-mel, librosa's Slaney filter bank, torchaudio's sinc resampler)
are absent from /root/reference and not installable here, so the oracle restates them and the fixtures were generated THROUGH
that restatement: an error in it would sit on both sides of every frontend comparison.  These tests hold the restatement against
implementations that ARE in the image and were written by somebody else:

* the Slaney mel filter bank against ``transformers.audio_utils.mel_filter_bank`` (HF's re-implementation of librosa.filters.mel);
* the periodic Hann window against ``scipy.signal.get_window``;
* the whole log-mel chain (reflect padding, framing, windowed DFT, power, mel, dB) against ``transformers.audio_utils.spectrogram``
  (an rFFT pipeline, where the oracle follows torchlibrosa's DFT-as-conv1d);
* the resampler against the analytic answer for band-limited tones (sample positions, gain, length: a one-sample shift or a wrong
  phase order fails by orders of magnitude) and against ``scipy.signal.resample_poly`` (another anti-aliasing filter: agreement to
  the filters' ripple).
The synthetic checkpoint's persisted tensors (``conette_amd.synth``: melW, DFT kernels) are held to the same references.
"""
import math

import numpy as np
import pytest
import torch

from oracle import thirdparty as T

SR, N_FFT, HOP, N_MELS, FMIN, FMAX = 32000, 1024, 320, 224, 50.0, 14000.0


def _hf_mel():
    from transformers import audio_utils as A
    return A.mel_filter_bank(num_frequency_bins=N_FFT // 2 + 1, num_mel_filters=N_MELS, min_frequency=FMIN, max_frequency=FMAX,
                             sampling_rate=SR, norm="slaney", mel_scale="slaney")          # (513, 224), float64


def test_mel_filterbank_against_independent_slaney_implementation():
    ours = T.mel_filterbank(SR, N_FFT, N_MELS, FMIN, FMAX)                                     # (224, 513)
    ref = _hf_mel().T
    assert ours.shape == ref.shape
    np.testing.assert_allclose(ours, ref, rtol=0, atol=2e-8)                                  # float32 storage of O(0.05) weights
    from conette_amd import synth
    np.testing.assert_allclose(synth._mel_filterbank(), ref if synth._mel_filterbank().shape == ref.shape else ref.T, rtol=0, atol=2e-8)
    # every filter is a unimodal triangle with a contiguous support inside [fmin, fmax]
    freqs = np.arange(N_FFT // 2 + 1) * SR / N_FFT
    for row in ours:
        nz = np.nonzero(row)[0]
        assert len(nz) > 0 and np.all(np.diff(nz) == 1)
        assert freqs[nz[0]] >= FMIN - SR / N_FFT and freqs[nz[-1]] <= FMAX + SR / N_FFT


def test_hann_window_and_dft_kernels():
    from scipy.signal import get_window
    np.testing.assert_allclose(T.hann_periodic(N_FFT), get_window("hann", N_FFT, fftbins=True), rtol=0, atol=1e-7)
    real, imag = T.dft_conv_kernels(N_FFT)                                                     # (513, 1, 1024)
    n = np.arange(N_FFT)
    win = get_window("hann", N_FFT, fftbins=True)
    for k in (0, 1, 37, 256, 512):
        np.testing.assert_allclose(real[k, 0], np.cos(2 * np.pi * k * n / N_FFT) * win, atol=2e-6)
        np.testing.assert_allclose(imag[k, 0], -np.sin(2 * np.pi * k * n / N_FFT) * win, atol=2e-6)
    from conette_amd import synth
    sr_, si_ = synth._dft_kernels(N_FFT)
    np.testing.assert_allclose(np.asarray(sr_).reshape(real.shape), real, atol=2e-6)
    np.testing.assert_allclose(np.asarray(si_).reshape(imag.shape), imag, atol=2e-6)


def test_logmel_chain_against_independent_rfft_pipeline():
    from transformers import audio_utils as A
    g = np.random.default_rng(5)
    n = 3 * SR + 123                                                                          # an odd length
    t = np.arange(n) / SR
    wave = (0.1 * g.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 5200.0 * t)).astype(np.float32)
    spec = T.Spectrogram(n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window="hann", center=True, pad_mode="reflect")
    lm = T.LogmelFilterBank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX, ref=1.0, amin=1e-10, top_db=None)
    with torch.no_grad():
        ours = lm(spec(torch.from_numpy(wave)[None]))[0, 0].numpy()                           # (frames, 224)
    ref = A.spectrogram(wave.astype(np.float64), np.asarray(A.window_function(N_FFT, "hann", periodic=True), dtype=np.float64),
                        frame_length=N_FFT, hop_length=HOP, fft_length=N_FFT, power=2.0, center=True, pad_mode="reflect",
                        mel_filters=_hf_mel(), mel_floor=1e-10, log_mel="dB", reference=1.0, min_value=1e-10, dtype=np.float64).T
    assert ours.shape == ref.shape == (n // HOP + 1, N_MELS)
    # fp32 DFT-as-conv1d over 1024 taps against a float64 rFFT: the power agrees to ~1e-5 relative, i.e. ~5e-5 dB
    np.testing.assert_allclose(ours, ref, rtol=0, atol=2e-3)
    assert float(np.abs(ours - ref).mean()) < 1e-4


@pytest.mark.parametrize("orig,new", [(44100, 32000), (48000, 32000), (16000, 32000), (22050, 32000)])
def test_resampler_against_analytic_tones_and_scipy(orig, new):
    from scipy.signal import resample_poly
    n_in = orig  # one second
    t_in = np.arange(n_in) / orig
    freqs = [220.0, 1000.0, 0.2 * min(orig, new)]                                              # well inside both Nyquist bands
    x = sum(np.sin(2 * np.pi * f * t_in + 0.3 * i) for i, f in enumerate(freqs)) / len(freqs)
    y = T.resample(torch.from_numpy(x.astype(np.float32))[None], orig, new)[0].numpy()
    n_out = int(math.ceil(new * n_in / orig))
    assert y.shape == (n_out,)
    t_out = np.arange(n_out) / new
    ideal = sum(np.sin(2 * np.pi * f * t_out + 0.3 * i) for i, f in enumerate(freqs)) / len(freqs)
    inner = slice(200, n_out - 200)                                                            # away from the zero-padded edges
    # windowed sinc of width 6 / rolloff 0.99: pass-band ripple and transition leak stay below 1e-2 of full scale
    assert np.abs(y[inner] - ideal[inner]).max() < 1e-2, np.abs(y[inner] - ideal[inner]).max()
    # a one-sample shift of the output grid would show as ~2 pi f / new = 0.2 .. 1.2 of full scale
    shifted = np.roll(ideal, 1)
    assert np.abs(y[inner] - shifted[inner]).max() > 10 * np.abs(y[inner] - ideal[inner]).max()
    g = math.gcd(orig, new)
    sp = resample_poly(x, new // g, orig // g)
    assert sp.shape == y.shape
    assert np.abs(y[inner] - sp[inner]).max() < 1.5e-2

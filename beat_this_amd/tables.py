"""Constant tables uploaded once per device: Hann window, FFT twiddles, banded mel
filterbank, RoPE cos/sin.  Formulas follow the third-party code the reference calls
(torchaudio 2.3.1 ``melscale_fbanks``/``Spectrogram`` via beat_this/preprocessing.py:43-53,
rotary-embedding-torch 0.6.4 via beat_this/model/beat_tracker.py:52); see SURVEY.md 8c.
"""
from __future__ import annotations

import math

import numpy as np
import torch

SAMPLE_RATE, N_FFT, HOP, N_MELS, F_MIN, F_MAX = 22050, 1024, 441, 128, 30.0, 11000.0
MEL_MAXLEN = 32


def mel_filterbank() -> torch.Tensor:
    """(513, 128) fp32 slaney filterbank, norm=None -- evaluated in fp32 like torchaudio."""
    f_sp = 200.0 / 3.0
    logstep = math.log(6.4) / 27.0
    min_log_mel = 1000.0 / f_sp

    def hz_to_mel(f: float) -> float:
        return min_log_mel + math.log(f / 1000.0) / logstep if f >= 1000.0 else f / f_sp

    freqs = torch.linspace(0, SAMPLE_RATE // 2, N_FFT // 2 + 1)
    mel_pts = torch.linspace(hz_to_mel(F_MIN), hz_to_mel(F_MAX), N_MELS + 2)
    hz_pts = f_sp * mel_pts
    is_log = mel_pts >= min_log_mel
    hz_pts[is_log] = 1000.0 * torch.exp(logstep * (mel_pts[is_log] - min_log_mel))
    widths = hz_pts[1:] - hz_pts[:-1]
    dist = hz_pts.unsqueeze(0) - freqs.unsqueeze(1)
    rising = (-1.0 * dist[:, :-2]) / widths[:-1]
    falling = dist[:, 2:] / widths[1:]
    return torch.max(torch.zeros(1), torch.min(rising, falling))


def logmel_tables() -> dict:
    """Host arrays for bt_logmel: window[1024], twiddle[1089,2], mel_start/len[128], mel_w[128,32]."""
    window = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32).numpy()
    tw = np.zeros((64 + 512 + 513, 2), dtype=np.float64)
    b, p = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    ang = -2.0 * np.pi * (b * p) / 64.0
    tw[:64, 0], tw[:64, 1] = np.cos(ang).ravel(), np.sin(ang).ravel()
    c, m = np.meshgrid(np.arange(8), np.arange(64), indexing="ij")
    ang = -2.0 * np.pi * (c * m) / 512.0
    tw[64:576, 0], tw[64:576, 1] = np.cos(ang).ravel(), np.sin(ang).ravel()
    k = np.arange(513)
    ang = -2.0 * np.pi * k / 1024.0
    tw[576:, 0], tw[576:, 1] = np.cos(ang), np.sin(ang)
    fb = mel_filterbank().numpy()
    start = np.zeros(N_MELS, dtype=np.int32)
    length = np.zeros(N_MELS, dtype=np.int32)
    w = np.zeros((N_MELS, MEL_MAXLEN), dtype=np.float32)
    for j in range(N_MELS):
        nz = np.nonzero(fb[:, j])[0]
        if len(nz):
            start[j], length[j] = nz[0], nz[-1] - nz[0] + 1
            assert length[j] <= MEL_MAXLEN
            w[j, : length[j]] = fb[nz[0]: nz[-1] + 1, j]
    return dict(window=window, twiddle=tw.astype(np.float32), mel_start=start, mel_len=length, mel_w=w)


def rope_table(freqs: torch.Tensor, n_pos: int = 1536) -> np.ndarray:
    """[n_pos, 16, 2] (cos, sin) of pos * freqs in fp32, as rotary-embedding-torch evaluates them."""
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * freqs.float()[None, :]
    return torch.stack((ang.cos(), ang.sin()), dim=-1).numpy().astype(np.float32)


# Resampler design (replaces soxr.resample(..., quality="HQ") of inference.py:274-275; libsoxr itself is unavailable, so
# this follows its published HQ specification rather than its code): linear phase, pass band flat up to 91.3 % of the
# lower Nyquist frequency, stop band from 100 % of it (no aliasing into the pass band), >= 120 dB rejection (soxr HQ =
# 20 bit).  Kaiser-windowed sinc: beta from the attenuation, length from the transition width (Kaiser's formulas).
RESAMPLE_PASSBAND_END = 0.913
RESAMPLE_ATTENUATION_DB = 125.0


def resample_filter(up: int, down: int):
    """(h float64 [2 half + 1], half): low-pass of the rational resampler x[up/down], designed at the upsampled rate
    (in_rate * up): -6 dB point in the middle of the transition band [0.913, 1.0] x the lower Nyquist frequency,
    Kaiser window for 125 dB, unit DC gain, multiplied by ``up`` (the interpolation gain)."""
    max_rate = max(up, down)
    att = RESAMPLE_ATTENUATION_DB
    beta = 0.1102 * (att - 8.7)
    width = (1.0 - RESAMPLE_PASSBAND_END) * 0.5 / max_rate            # transition width, cycles / sample (upsampled rate)
    n_taps = int(math.ceil((att - 7.95) / (2.285 * 2.0 * math.pi * width))) + 1
    half = (n_taps + 1) // 2
    fc = 0.5 * (1.0 + RESAMPLE_PASSBAND_END) * 0.5 / max_rate          # -6 dB point, cycles / sample
    n = np.arange(-half, half + 1, dtype=np.float64)
    h = 2.0 * fc * np.sinc(2.0 * fc * n) * np.kaiser(2 * half + 1, beta)
    h /= h.sum()
    return h * up, half

"""TEST INFRASTRUCTURE ONLY -- restated torchaudio.transforms.MelSpectrogram (2.3.1)."""
import math

import torch


def _hz_to_mel_slaney(freq: float) -> float:
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if freq >= min_log_hz:
        mels = min_log_mel + math.log(freq / min_log_hz) / logstep
    return mels


def _mel_to_hz_slaney(mels: torch.Tensor) -> torch.Tensor:
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_slaney(f_min)
    m_max = _hz_to_mel_slaney(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


class MelSpectrogram(torch.nn.Module):
    def __init__(self, sample_rate, n_fft, hop_length, f_min, f_max, n_mels,
                 mel_scale, normalized, power):
        super().__init__()
        assert mel_scale == "slaney" and normalized == "frame_length"
        self.n_fft, self.hop, self.power = n_fft, hop_length, power
        self.register_buffer("window", torch.hann_window(n_fft), persistent=False)
        self.register_buffer(
            "fb", melscale_fbanks(n_fft // 2 + 1, float(f_min), float(f_max), n_mels, sample_rate),
            persistent=False)

    def forward(self, x):
        spec = torch.stft(x, n_fft=self.n_fft, hop_length=self.hop, win_length=self.n_fft,
                          window=self.window, center=True, pad_mode="reflect",
                          normalized=True, onesided=True, return_complex=True).abs()
        if self.power != 1:
            spec = spec ** self.power
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

"""Audio loading and the GPU log-mel front end.

``LogMelSpect`` mirrors beat_this.preprocessing.LogMelSpect (preprocessing.py:27-59): same
constructor arguments, ``forward(x: (N,)) -> (frames, 128)``; the STFT / mel / log1p run in
one HIP kernel (csrc/logmel.hip).  Only the reference's fixed configuration is supported.
``load_audio`` mirrors preprocessing.py:6-24 (host file I/O, not part of the GPU path).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, tables


def load_audio(path, dtype="float64"):
    """Decode an audio file to (samples[, channels]) + sample rate; same fallback order and
    the same final RuntimeError as the reference (preprocessing.py:6-24)."""
    try:
        import torchaudio

        waveform, samplerate = torchaudio.load(path, channels_first=False)
        return np.asanyarray(waveform.squeeze().numpy(), dtype=dtype), samplerate
    except Exception:
        try:
            import soundfile as sf

            return sf.read(path, dtype=dtype)
        except Exception:
            try:
                import madmom

                return madmom.io.load_audio_file(str(path), dtype=dtype)
            except Exception:
                pass
    try:  # last resort the reference does not have: plain PCM WAV through the stdlib/scipy
        from scipy.io import wavfile

        sr, data = wavfile.read(str(path))
        if data.dtype.kind == "i":
            data = data.astype(dtype) / float(np.iinfo(data.dtype).max + 1)
        elif data.dtype.kind == "u":
            data = (data.astype(dtype) - 128.0) / 128.0
        return data.astype(dtype), sr
    except Exception:
        raise RuntimeError(f'Could not load audio from "{path}".')


class LogMelSpect(torch.nn.Module):
    def __init__(self, sample_rate=22050, n_fft=1024, hop_length=441, f_min=30, f_max=11000, n_mels=128,
                 mel_scale="slaney", normalized="frame_length", power=1, log_multiplier=1000, device="cpu"):
        super().__init__()
        cfg = (sample_rate, n_fft, hop_length, f_min, f_max, n_mels, mel_scale, normalized, power, log_multiplier)
        if cfg != (22050, 1024, 441, 30, 11000, 128, "slaney", "frame_length", 1, 1000):
            raise ValueError("beat_this_amd.LogMelSpect implements the reference's fixed configuration only")
        self.device = torch.device(device)
        self._tables = None

    def to(self, device, *a, **k):  # mirrors nn.Module.to for the single thing that matters here
        self.device = torch.device(device)
        self._tables = None
        return self

    def _get_tables(self):
        if self._tables is None:
            host = tables.logmel_tables()
            dev = {k: torch.from_numpy(v).to(self.device) for k, v in host.items()}
            t = _lib.LogmelTables(dev["window"].data_ptr(), dev["twiddle"].data_ptr(), dev["mel_start"].data_ptr(),
                                  dev["mel_len"].data_ptr(), dev["mel_w"].data_ptr())
            self._tables = (t, dev)
        return self._tables[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 1:
            raise ValueError(f"expected a 1-D waveform, got shape {tuple(x.shape)}")
        _lib.require_gpu(x, "waveform")
        if x.device != self.device:
            self.to(x.device)
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        if n <= 512:
            raise ValueError("signal too short: reflect padding needs more than 512 samples")
        out = torch.empty((1 + n // 441, 128), dtype=torch.float32, device=x.device)
        import ctypes as C
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().bt_logmel(_lib.stream_ptr(x.device), C.byref(self._get_tables()), x.data_ptr(), n,
                                            out.data_ptr()))
        return out

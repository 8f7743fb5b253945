"""TEST INFRASTRUCTURE ONLY -- stand-in for torchaudio==2.3.1 (not installed here).

Only what beat_this/preprocessing.py:8,17,43-53 uses: ``transforms.MelSpectrogram`` and ``load``.
``load`` decodes integer-PCM WAV files only (scipy), normalised to [-1, 1) float32 like torchaudio's default
``normalize=True`` (int16 / 32768); anything else raises so that ``load_audio`` (preprocessing.py:6-24) falls through.
Restated from the published torchaudio 2.3.1 algorithm; PARITY UNPINNED.
"""
from . import transforms  # noqa: F401


def load(path, channels_first=True, **k):
    import numpy as np
    import torch
    from scipy.io import wavfile

    sr, data = wavfile.read(str(path))
    if data.dtype.kind != "i":
        raise RuntimeError("torchaudio stand-in decodes integer PCM WAV only")
    x = (data.astype(np.float64) / float(np.iinfo(data.dtype).max + 1)).astype(np.float32)
    if x.ndim == 1:
        x = x[:, None]
    t = torch.from_numpy(x)
    return (t.T.contiguous() if channels_first else t), sr

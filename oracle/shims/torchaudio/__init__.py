"""TEST INFRASTRUCTURE ONLY -- stand-in for torchaudio==2.3.1 (not installed here).

Only what beat_this/preprocessing.py:8,43-53 uses: ``transforms.MelSpectrogram``.
``load`` raises so that ``load_audio`` (preprocessing.py:6-24) falls through.
Restated from the published torchaudio 2.3.1 algorithm; PARITY UNPINNED.
"""
from . import transforms  # noqa: F401


def load(*a, **k):
    raise RuntimeError("torchaudio stand-in cannot decode audio")

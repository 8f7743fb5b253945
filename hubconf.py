"""torch.hub entry points, the counterpart of the reference's hubconf.py (hubconf.py:1-18: `beat_this` = load_model, plus the
inference classes), so that `torch.hub.load(<this repo>, "beat_this", ...)` / `"File2Beats"` keep working after the switch.
Nothing but torch and numpy is needed: resampler, log-mel, rotary embedding and the model run in libbeat_this_amd.so."""
dependencies = ["torch", "numpy"]

from beat_this_amd.inference import (  # noqa: E402,F401
    Audio2Beats,
    Audio2Frames,
    BeatThis,
    File2Beats,
    File2File,
    Spect2Frames,
)
from beat_this_amd.inference import load_model as beat_this  # noqa: E402,F401

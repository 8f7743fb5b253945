"""beat_this_amd -- MI355X-native (gfx950) implementation of the beat_this inference hot path.

Public surface mirrors beat_this.inference / hubconf.py of the reference; the arithmetic runs
in hand-written HIP kernels behind the C ABI of include/beat_this_amd.h."""
from .inference import (Audio2Beats, Audio2Frames, File2Beats, File2File, Spect2Frames, load_checkpoint,  # noqa: F401
                        load_model, split_predict_aggregate)
from .model import BeatThis  # noqa: F401
from .postprocessor import Postprocessor  # noqa: F401
from .preprocessing import LogMelSpect, load_audio  # noqa: F401

__version__ = "0.1.0"

"""Import path of the reference (beat_this/model/beat_tracker.py:18): ``from beat_this_amd.model.beat_tracker import BeatThis``."""
from . import BeatThis  # noqa: F401

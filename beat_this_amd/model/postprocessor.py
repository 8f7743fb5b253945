"""Import path of the reference (beat_this/model/postprocessor.py:9,176): ``from beat_this_amd.model.postprocessor import Postprocessor``."""
from ..postprocessor import Postprocessor, deduplicate_peaks  # noqa: F401

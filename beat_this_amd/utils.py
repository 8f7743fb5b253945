"""Host-side helpers with the reference's behaviour (beat_this/utils.py:26-111)."""
from __future__ import annotations

from pathlib import Path

import numpy as np


def replace_state_dict_key(state_dict: dict, old: str, new: str) -> dict:
    """Rename every key containing ``old`` (utils.py:105-111); mutates and returns the dict."""
    for key in [k for k in state_dict if old in k]:
        state_dict[key.replace(old, new)] = state_dict.pop(key)
    return state_dict


def infer_beat_numbers(beats: np.ndarray, downbeats: np.ndarray) -> np.ndarray:
    """Beat counter per beat, 1 on every downbeat (utils.py:26-76), same warnings/errors."""
    if not np.all(np.isin(downbeats, beats)):
        raise ValueError("Not all downbeats are beats.")
    start = 1
    if len(downbeats) >= 2:
        first, second = np.searchsorted(beats, downbeats[:2])
        if first < second - first:
            start = (second - first) - first
        else:
            print("WARNING: There are more beats in the pickup measure than in the first measure. The beat count "
                  "will start from 2 without trying to estimate the length of the pickup measure.")
    else:
        print("WARNING: There are less than two downbeats in the predictions. Something may be wrong. The beat "
              "count will start from 2 without trying to estimate the length of the pickup measure.")
    numbers = np.empty(len(beats), dtype=np.int64)
    counter = start
    di = 0
    nd = len(downbeats)
    for i, b in enumerate(beats):
        if di < nd and b == downbeats[di]:
            counter = 1
            di += 1
        else:
            counter += 1
        numbers[i] = counter
    return numbers


def save_beat_tsv(beats: np.ndarray, downbeats: np.ndarray, outpath) -> None:
    """``time<TAB>beat_number`` lines, the .beats format (utils.py:79-102)."""
    numbers = infer_beat_numbers(beats, downbeats)
    outpath = Path(outpath)
    outpath.parent.mkdir(parents=True, exist_ok=True)
    try:
        with open(outpath, "w") as f:
            f.writelines(f"{b}\t{n}\n" for b, n in zip(beats, numbers))
    except KeyboardInterrupt:
        outpath.unlink()  # no half-written files

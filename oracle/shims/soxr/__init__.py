"""TEST INFRASTRUCTURE ONLY -- stand-in for soxr==0.3.7 (libsoxr, not installed).

NOT bit-compatible with libsoxr; never part of a parity assertion (SURVEY 8c last
row).  Parity starts at the 22.05 kHz waveform.
"""
from math import gcd

import numpy as np
from scipy.signal import resample_poly


def resample(x, in_rate, out_rate, quality="HQ"):
    g = gcd(int(in_rate), int(out_rate))
    return resample_poly(np.asarray(x), int(out_rate) // g, int(in_rate) // g, axis=0)

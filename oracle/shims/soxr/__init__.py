"""TEST INFRASTRUCTURE ONLY -- stand-in for soxr==0.3.7 (libsoxr, not installed).

NOT bit-compatible with libsoxr; never part of a parity assertion against the reference itself (SURVEY 8c last row:
parity starts at the 22.05 kHz waveform).  Restates libsoxr's published "HQ" quality specification (linear phase,
pass band to 91.3 % of the lower Nyquist frequency, stop band from 100 %, 20-bit = ~125 dB rejection) as a Kaiser-windowed
sinc in float64 via scipy; it is the CPU side of the 44.1 kHz end-to-end tests of the GPU resampler.
"""
from math import ceil, gcd, pi

import numpy as np
from scipy.signal import firwin, resample_poly

PASSBAND_END = 0.913
ATTENUATION_DB = 125.0


def hq_filter(up, down):
    """Odd-length linear-phase low-pass at the upsampled rate (Nyquist = 1.0 in firwin's units)."""
    m = max(up, down)
    width = (1.0 - PASSBAND_END) / m                      # in units of the upsampled Nyquist frequency
    n_taps = int(ceil((ATTENUATION_DB - 7.95) / (2.285 * pi * width))) + 1
    half = (n_taps + 1) // 2
    return firwin(2 * half + 1, 0.5 * (1.0 + PASSBAND_END) / m, window=("kaiser", 0.1102 * (ATTENUATION_DB - 8.7)))


def resample(x, in_rate, out_rate, quality="HQ"):
    g = gcd(int(in_rate), int(out_rate))
    up, down = int(out_rate) // g, int(in_rate) // g
    if up == down:
        return np.asarray(x)
    return resample_poly(np.asarray(x, dtype=np.float64), up, down, axis=0, window=hq_filter(up, down))

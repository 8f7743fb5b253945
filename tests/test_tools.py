"""The measurement tools that feed profiles/ and the bench line are code too: the flip counter and the margin statistic of
tools/flip_soak.py on hand-made logit vectors (CPU), and the energy meter of tools/smi.py on the GPU box."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_flip_counter_and_margins_on_hand_made_logits():
    import flip_soak as F

    x = np.full(60, -5.0, dtype=np.float32)
    x[10], x[20], x[30], x[31] = 2.0, 1.0, 3.0, 3.0        # two single peaks, one two-frame plateau (-> 30.5)
    assert F.frames_of(x) == {20, 40, 61}                    # (half-frame units)
    y = x.copy()
    y[20] = -1.0                                             # a beat lost
    y[45] = 0.5                                              # a beat gained
    assert F.flips(x, y) == 2 and F.flips(x, x) == 0
    # margins: the peak at 20 is 1.0 above zero and 6.0 above its neighbours -> margin 1.0; the plateau frames tie (margin 0)
    m = F.margin_stats(x)
    assert m[0] == 0.0 and np.isclose(np.sort(m[m > 0])[0], 1.0)
    z = x.copy()
    z[12] = 1.9999                                           # a runner-up 1e-4 below the peak at 10, inside its window
    assert np.isclose(np.sort(F.margin_stats(z)[F.margin_stats(z) > 0])[0], 1e-4, rtol=1e-2)


@pytest.mark.gpu
def test_energy_meter_reads_the_package_accumulator():
    import time

    import torch

    from tools.smi import EnergyMeter

    m = EnergyMeter(torch.device("cuda:0"))
    if not m.ok:
        pytest.skip(f"energy accumulator not readable here: {m.why}")
    x = torch.randn((4096, 4096), device="cuda:0")
    m.start()
    t0 = time.time()
    while time.time() - t0 < 0.3:
        (x @ x).sum().item()
    j, sec = m.stop()
    assert 0.25 < sec < 2.0 and 30.0 < j / sec < 1500.0, (j, sec)    # between idle (~250 W) and the 1400 W cap

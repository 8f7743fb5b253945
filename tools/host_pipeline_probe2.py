#!/usr/bin/env python
"""Development tool (GPU): where does Audio2Beats.many_async block the CPU (device vs pinned-host inputs)?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beat_this_amd import weights as W
from beat_this_amd.inference import Audio2Beats
from beat_this_amd.model import BeatThis

dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
a2b = Audio2Beats(checkpoint_path=None, device=dev, float16=True)
m = BeatThis(**hp)
m.load_state_dict(W.random_state_dict(hp, seed=1, style="lively"))
a2b.model = m.to(dev).eval()
dtr = [torch.from_numpy(W.synthetic_audio(300.0, seed=i, sr=44100)).to(dev) for i in range(6)]
htr = [t.cpu().pin_memory() for t in dtr]


def one(tr):
    t0 = time.perf_counter()
    spect, foff = a2b.signal2spect_many(tr, 44100)
    t1 = time.perf_counter()
    beat, down = a2b.spect2frames_batch(spect, foff)
    t2 = time.perf_counter()
    p = a2b.frames2beats.ragged_async(beat, down, foff)
    t3 = time.perf_counter()
    return p, (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2))


for name, tr in (("device", dtr), ("pinned host", htr)):
    pend = []
    for _ in range(3):
        pend.append(one(tr)[0])
        if len(pend) > 1:
            pend.pop(0).result()
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    rows = []
    for _ in range(5):
        p, r = one(tr)
        pend.append(p)
        if len(pend) > 1:
            pend.pop(0).result()
        rows.append(r)
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    print(f"{name:12s} signal2spect_many / spect2frames_batch / ragged_async (CPU ms): " + "  ".join("%.1f/%.1f/%.1f" % r for r in rows), flush=True)

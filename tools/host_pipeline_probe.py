#!/usr/bin/env python
"""Development tool (GPU): CPU-side timeline of Audio2Beats.many_async fed from pinned host buffers vs device buffers."""
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
for name, tr in (("device", dtr), ("pinned host", htr), ("device", dtr), ("pinned host", htr)):
    pend = []
    for _ in range(3):
        pend.append(a2b.many_async(tr, 44100))
        if len(pend) > 1:
            pend.pop(0).result()
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    rows = []
    t00 = time.perf_counter()
    for _ in range(8):
        t0 = time.perf_counter()
        pend.append(a2b.many_async(tr, 44100))
        t1 = time.perf_counter()
        if len(pend) > 1:
            pend.pop(0).result()
        t2 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - t00) / 8
    print(f"{name:12s} {tot:6.2f} ms/step   enqueue / wait-for-previous per step: " + "  ".join(f"{a:.1f}/{b:.1f}" for a, b in rows), flush=True)

#!/usr/bin/env python
"""Development tool (GPU): soak of the pipelined multi-track path in the bench configuration -- N batches of 6 x 300 s
tracks through Audio2Beats.many_async (two forward streams, pinned host uploads on the copy stream, two batches in flight),
every batch's framewise logits and beats compared bit for bit with the first batch's.
    python tools/soak.py [batches] [device|pinned] [half|f32|f32x3]
(f32x3: BT_PREC_F32X3 with its deferred range flags -- the soak also asserts that no batch fell back to exact fp32)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from beat_this_amd import weights as W
from beat_this_amd.inference import Audio2Beats
from beat_this_amd.model import BeatThis

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else "pinned"
prec = sys.argv[3] if len(sys.argv) > 3 else "half"
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
a2b = Audio2Beats(checkpoint_path=None, device=dev, float16={"half": True, "f32": "exact", "f32x3": False}[prec])
m = BeatThis(**hp)
m.load_state_dict(W.random_state_dict(hp, seed=1, style="lively"))
a2b.model = m.to(dev).eval()
a2b.model.fp32_split_gemms = prec == "f32x3"
tracks = [torch.from_numpy(W.synthetic_audio(300.0, seed=i, sr=44100)) for i in range(6)]
tracks = [t.pin_memory() for t in tracks] if mode == "pinned" else [t.to(dev) for t in tracks]
ref = None
bad = 0
pend = []
t0 = time.perf_counter()


def collect(h):
    global ref, bad
    beats = h.result()
    logits = torch.cat((h.logits[0], h.logits[1])).cpu()
    if ref is None:
        ref = (logits, beats)
        return
    same = torch.equal(logits, ref[0]) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(beats, ref[1]))
    bad += not same


for i in range(n):
    pend.append(a2b.many_async(tracks, 44100))
    if len(pend) > 1:
        collect(pend.pop(0))
while pend:
    collect(pend.pop(0))
dt = time.perf_counter() - t0
fallbacks = a2b.model.engine().last_fallbacks
print(f"soak ({mode} inputs, {prec}): {n} batches of 6 tracks in {dt:.1f} s ({n * 1800 / dt / 1e3:.1f} k audio-s/s incl. the comparisons), "
      f"{bad} batch(es) differ from the first, {fallbacks} range fallback(s)")
sys.exit(1 if bad or fallbacks else 0)

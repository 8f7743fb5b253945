#!/usr/bin/env python
"""Single-file latency (BASELINE config 1) in detail: Audio2Beats.__call__ on one 30 s / 300 s host waveform, wall time per
call and where it goes (host time until everything is enqueued, GPU time of the call from stream events).
    python tools/latency_probe.py [f32x3|half|exact] [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beat_this_amd import weights as W  # noqa: E402
from beat_this_amd.inference import Audio2Beats  # noqa: E402
from beat_this_amd.model import BeatThis  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
a2b = Audio2Beats(checkpoint_path=None, device=dev, float16={"half": True, "exact": "exact", "f32x3": False}[prec])
m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
m.load_state_dict(W.random_state_dict(hp, seed=1, style="lively"))
a2b.model = m.to(dev)
sig = W.synthetic_audio(secs, seed=7, sr=44100)
for _ in range(5):
    a2b(sig, 44100)
torch.cuda.synchronize()
wall, gpu, host = [], [], []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    spect = a2b.signal2spect(sig, 44100)
    t1 = time.perf_counter()
    bl, dl = a2b.spect2frames(spect)
    t2 = time.perf_counter()
    out = a2b.frames2beats(bl, dl)
    b.record()
    t3 = time.perf_counter()
    b.synchronize()
    wall.append(t3 - t0)
    gpu.append(a.elapsed_time(b) * 1e-3)
    host.append((t1 - t0, t2 - t1, t3 - t2))
med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
# the public call (round 6: ONE library call per track, bt_audio2beats_enqueue) next to the stage-by-stage route timed above
one = []
for _ in range(25):
    t0 = time.perf_counter()
    out1 = a2b(sig, 44100)
    one.append(time.perf_counter() - t0)
one = one[5:]
assert np.array_equal(out1[0], out[0]) and np.array_equal(out1[1], out[1])
print(f"{prec} {secs:.0f} s: Audio2Beats.__call__ (one library call) wall {med(one) * 1e3:.3f} ms (min {min(one) * 1e3:.3f}); stage by stage:")
print(f"{prec} {secs:.0f} s: wall {med(wall) * 1e3:.3f} ms (min {min(wall) * 1e3:.3f}), stream time {med(gpu) * 1e3:.3f} ms; "
      f"host: signal2spect {med([h[0] for h in host]) * 1e3:.3f}, spect2frames {med([h[1] for h in host]) * 1e3:.3f}, "
      f"frames2beats {med([h[2] for h in host]) * 1e3:.3f} ms; {len(out[0])} beats")

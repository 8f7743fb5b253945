#!/usr/bin/env python
"""Two (or more) host threads, each with its own Audio2Beats (own engine) on its own stream, running single-file calls at the same
time: every result must equal the single-threaded result of the same file (same bits: the routes are deterministic), in the
default and in the fp16 precision.  A serving process does exactly this.
    python tools/thread_stress.py [threads] [files_per_thread] [shared]      (shared: ONE Audio2Beats for all threads)"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beat_this_amd import weights as W  # noqa: E402
from beat_this_amd.inference import Audio2Beats  # noqa: E402
from beat_this_amd.model import BeatThis  # noqa: E402

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_files = int(sys.argv[2]) if len(sys.argv) > 2 else 24
shared = len(sys.argv) > 3 and sys.argv[3] == "shared"
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
sd = W.random_state_dict(hp, seed=1, style="lively")


def make(f16):
    a = Audio2Beats(checkpoint_path=None, device=dev, float16=f16)
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(sd)
    a.model = m.to(dev)
    return a


rng = np.random.default_rng(0)
files = []
for i in range(n_files):
    secs = float(rng.uniform(0.5, 50.0) if i % 2 else rng.uniform(50.0, 200.0))
    sr = int(rng.choice([22050, 44100, 48000]))
    files.append((W.synthetic_audio(secs, seed=2000 + i, sr=sr), sr))
bad_total = 0
for f16 in (False, True):
    ref_engine = make(f16)
    want = [ref_engine(sig, sr) for sig, sr in files]
    results = [None] * n_threads
    common = make(f16) if shared else None
    errors = []

    def work(t):
        try:
            a = common if shared else make(f16)
            st = torch.cuda.Stream(device=dev)
            out = []
            with torch.cuda.stream(st):
                order = list(range(n_files))
                np.random.default_rng(t).shuffle(order)
                for j in order:
                    sig, sr = files[j]
                    out.append((j, a(sig, sr)))
            results[t] = (out, a.model.engine().last_fallbacks)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    bad = 0
    for t in range(n_threads):
        if results[t] is None:
            continue
        for j, (b, d) in results[t][0]:
            if not (np.array_equal(b, want[j][0]) and np.array_equal(d, want[j][1])):
                bad += 1
    fb = [r[1] for r in results if r is not None]
    print(f"float16={f16}{' (one shared engine)' if shared else ''}: {n_threads} threads x {n_files} files: {bad} result(s) differ from the single-threaded run, range fallbacks {fb}, errors {errors}", flush=True)
    bad_total += bad + len(errors)
sys.exit(1 if bad_total else 0)

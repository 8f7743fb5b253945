#!/usr/bin/env python
"""Development tool (GPU): run the same batched forward several times and compare bit for bit.  Works in this tree and
in a checkout of an older commit (run it with that checkout as the working directory):
    python tools/determinism_probe.py [repeats]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from beat_this_amd import weights as W
from beat_this_amd.model import BeatThis

dev = torch.device("cuda:0")
KEYS = ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def probe(hp_name, B, half, n_layers=6):
    hp = dict(W.resolve_hparams(hp_name), n_layers=n_layers)
    sd = W.random_state_dict(hp, seed=1, style="lively")
    m = BeatThis(**{k: hp[k] for k in KEYS})
    m.load_state_dict(sd)
    m = m.to(dev)
    x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=7000 + i) for i in range(B)])).to(dev)
    outs = []
    with torch.inference_mode(), torch.autocast("cuda", enabled=half):
        for _ in range(reps):
            outs.append(m(x)["beat"].clone())
    torch.cuda.synchronize()
    # majority result = the mode over repeats per chunk; count deviating (repeat, chunk) pairs
    bad = []
    for i in range(B):
        rows = torch.stack([o[i] for o in outs])
        ref = rows.median(0).values
        for r in range(reps):
            d = float((rows[r] - ref).abs().max())
            if d > 0:
                bad.append((r, i, round(d, 4)))
    print(f"{hp_name} L={n_layers} B={B} half={half}: {len(bad)} deviating (repeat, chunk) pairs of {reps * B}: {bad[:8]}", flush=True)


for half in (False, True):
    probe("final0", 16, half)
    probe("final0", 16, half, n_layers=0)
    probe("small0", 16, half)
    probe("final0", 4, half)

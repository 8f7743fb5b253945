#!/usr/bin/env python
"""Development tool (GPU): where does a batched forward differ from the same chunks forwarded alone?
    python tools/scale_bisect.py            -> per (precision, n_layers, B): worst |batched - alone|, which chunks / frames"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from beat_this_amd import weights as W
from beat_this_amd.model import BeatThis

dev = torch.device("cuda:0")
KEYS = ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")


def run(hp_name, n_layers, B, half, T=1500, same_input=False):
    hp = dict(W.resolve_hparams(hp_name), n_layers=n_layers)
    sd = W.random_state_dict(hp, seed=1, style="lively")
    m = BeatThis(**{k: hp[k] for k in KEYS})
    m.load_state_dict(sd)
    m = m.to(dev)
    x = torch.from_numpy(np.stack([W.synthetic_spect(T, seed=7000 + (0 if same_input else i)) for i in range(B)])).to(dev)
    with torch.inference_mode(), torch.autocast("cuda", enabled=half):
        r = m(x)
        r2 = m(x)
        rep = float((r["beat"] - r2["beat"]).abs().max())
        worst, where = 0.0, []
        for i in range(B):
            ri = m(x[i: i + 1])
            d = (ri["beat"][0] - r["beat"][i]).abs()
            if float(d.max()) > 1e-5:
                bad = torch.nonzero(d > 1e-5)[:, 0]
                where.append((i, int(bad.min()), int(bad.max()), int(bad.numel()), round(float(d.max()), 4)))
            worst = max(worst, float(d.max()))
    print(f"{hp_name} layers={n_layers} B={B} T={T} half={half} same_input={same_input}: worst {worst:.3e}  repeat-diff {rep:.1e}  "
          f"bad chunks (chunk, first frame, last frame, count, max): {where[:6]}", flush=True)


if __name__ == "__main__":
    for half in (False, True):
        run("final0", 6, 16, half)
        run("final0", 0, 16, half)
        run("final0", 1, 16, half)
        run("final0", 6, 2, half)
        run("final0", 6, 16, half, same_input=True)
        run("final0", 6, 16, half, T=1024)
    run("small0", 6, 16, False)

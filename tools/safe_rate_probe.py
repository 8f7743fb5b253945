#!/usr/bin/env python
"""Development (-DBT_DEV build): how many queries of a forward's x3 attention launches overflow fp16 in the fast pass and are
recomputed by the gathered fix-up launch (attn_fix_x3_kernel), and in how many (sequence, head) pairs they sit -- words 1 / 3 of
the workspace's status block; word 2 counts the workgroups of the attention kernels in front (attn2.hip).
    BT_DEV=1 BT_LIB_PATH=tools/variants/lib_dev.so python tools/safe_rate_probe.py [chunks] [style]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beat_this_amd import _lib, weights as W  # noqa: E402
from beat_this_amd.model import BeatThis  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
style = sys.argv[2] if len(sys.argv) > 2 else "lively"
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
m.load_state_dict(W.random_state_dict(hp, seed=1, style=style))
m = m.to(dev)
m.fp32_split_gemms = True
x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=1000 + i) for i in range(B)])).to(dev)
with torch.inference_mode():
    m(x)
    m(x)
torch.cuda.synchronize()
eng = m.engine()
ws = list(eng._ws.values())[-1][0]
st = ws[:16].view(torch.int32).cpu().tolist()
queries = B * 1500 * (3 * 32 + 6 * hp["transformer_dim"] // 32)   # query rows of the nine attention launches of one forward
pairs = B * (3 * 32 + 6 * hp["transformer_dim"] // 32)
print(f"{style}, {B} chunks: status words {st}: {st[1]} of {queries} queries of one forward ({100.0 * st[1] / queries:.3f} %) were left to "
      f"the fix-up launch, in {st[3]} of {pairs} (sequence, head) pairs ({100.0 * st[3] / pairs:.1f} %); {st[2]} attention workgroups in front")

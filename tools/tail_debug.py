#!/usr/bin/env python
"""Development tool (GPU): bt_layer_tail against fp64 on small shapes, with a map of where it goes wrong."""
import ctypes as Ct
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beat_this_amd import _lib as L
from beat_this_amd.pack import PackedPair

dev = torch.device("cuda:0")


from tools.tail_debug_util import pair_sd  # noqa: E402

for C, hidden, M in ((512, 128, 128), (512, 256, 128), (512, 2048, 128), (512, 2048, 777)):
    sd = pair_sd(C, hidden, 3)
    pp = PackedPair(sd, "a.", "f.", C, dev)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn((M, C), generator=g, dtype=torch.float64) * 1.5
    ao = torch.randn((M, C), generator=g, dtype=torch.float64).float().to(L.half_torch_dtype())
    x = x0.float().to(dev).clone()
    aod = ao.to(dev)
    L.check(L.lib().bt_layer_tail(L.stream_ptr(dev), Ct.byref(pp.weights), hidden, aod.data_ptr(), x.data_ptr(), M, 0, 0))
    torch.cuda.synchronize()
    x1 = x0 + ao.double() @ sd["a.to_out.0.weight"].T
    xn = x1 / x1.norm(dim=-1, keepdim=True) * math.sqrt(C) * sd["f.net.0.gamma"]
    ref = x1 + torch.nn.functional.gelu(xn @ sd["f.net.1.weight"].T + sd["f.net.1.bias"]) @ sd["f.net.4.weight"].T + sd["f.net.4.bias"]
    got = x.double().cpu()
    nan = torch.isnan(got)
    err = (got - ref).abs()
    err[nan] = 0
    e1 = (got - x1).abs()
    print(f"C={C} hidden={hidden} M={M}: NaN {int(nan.sum())} of {got.numel()}  max err (non-NaN) {float(err.max()):.3e}  "
          f"rows with NaN {int(nan.any(1).sum())}  cols with NaN {int(nan.any(0).sum())}  "
          f"|got - (x + out-proj)| max {float(e1[~nan].max()) if (~nan).any() else -1:.3e}", flush=True)
    if nan.any():
        rows = torch.nonzero(nan.any(1))[:, 0]
        cols = torch.nonzero(nan.any(0))[:, 0]
        print("   first NaN rows", rows[:8].tolist(), "cols", cols[:8].tolist(), "...", cols[-4:].tolist())

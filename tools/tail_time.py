#!/usr/bin/env python
"""Development tool (GPU): time bt_layer_tail alone (HIP events, C = 512, hidden = 2048) for the loaded library.
    BT_DEV=1 BT_LIB_PATH=tools/bin/lib_X.so python tools/tail_time.py [M ...]"""
import ctypes as Ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beat_this_amd import _lib as L
from beat_this_amd.pack import PackedPair
from tools.tail_debug_util import pair_sd

dev = torch.device("cuda:0")
C, hidden = 512, 2048
pp = PackedPair(pair_sd(C, hidden, 3), "a.", "f.", C, dev)
for M in [int(a) for a in sys.argv[1:]] or [24000]:
    x = torch.randn((M, C), device=dev)
    ao = torch.randn((M, C), device=dev).to(L.half_torch_dtype())
    xb = torch.empty((M, C), device=dev, dtype=L.half_torch_dtype())
    ssq = torch.empty((C // 64, M), device=dev)
    st = L.stream_ptr(dev)

    def go():
        L.check(L.lib().bt_layer_tail(st, Ct.byref(pp.weights), hidden, ao.data_ptr(), x.data_ptr(), M, xb.data_ptr(), ssq.data_ptr()))
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{os.environ.get('BT_LIB_PATH', 'in-tree'):40s} M={M:6d}  {best * 1e3:7.1f} us", flush=True)

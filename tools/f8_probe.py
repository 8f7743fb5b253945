#!/usr/bin/env python
"""hl8 (config 5: fp8 cross terms) against hl32 operands in csrc/gemm3.hip, FF1 / FF2 / out-projection shapes of a 33-chunk slice:
us and joules per launch.  GPU box, development tool.      python tools/f8_probe.py [chunks]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L  # noqa: E402
from gpu_util import pad_rows, to_hl8, to_hl32  # noqa: E402
from tools.smi import Smi  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 33
lib, smi, st = L.lib(), Smi(dev), L.stream_ptr(dev)


def run(label, M, K, N, epi, f8):
    g = torch.Generator().manual_seed(2)
    pack = to_hl8 if f8 else to_hl32
    A = pack(torch.randn((M, K), generator=g)).to(dev)
    W = pack(pad_rows(torch.randn((N, K), generator=g) / K ** 0.5, 256)).to(dev)
    a = L.Gemm3Args()
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi, a.x3 = A.data_ptr(), K, M, K, W.data_ptr(), N, epi, 1 | (0x100 if f8 else 0)
    if epi == 0:
        bias = torch.zeros(N, device=dev); out = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev); ssq = torch.ones((K // 64, M), device=dev)
        a.bias, a.out, a.ldo, a.ssq_in, a.ssq_parts = bias.data_ptr(), out.data_ptr(), N, ssq.data_ptr(), K // 64
    else:
        x = torch.zeros((M, N), device=dev); xb = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev); ssq = torch.zeros((N // 64, M), device=dev)
        a.x, a.ldx, a.xb, a.ssq_out = x.data_ptr(), N, xb.data_ptr(), ssq.data_ptr()
    fn = lambda: L.check(lib.bt_gemm3(st, C.byref(a)))  # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, t0 = smi.energy_uj()
    w0, n = time.time(), 0
    while time.time() - w0 < 1.0:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    e1, t1 = smi.energy_uj()
    sec, j = (t1 - t0) * 1e-9, (e1 - e0) * 1e-6
    print(f"{label:18s} {'hl8 ' if f8 else 'hl32'} M={M} K={K} N={N}: {sec / n * 1e6:8.1f} us  {j / sec:7.1f} W  {j / n:7.4f} J / launch", flush=True)


M = B * 1500
for rep in range(2):
    for f8 in (False, True):
        run("FF1", M, 512, 2048, 0, f8)
        run("FF2", M, 2048, 512, 1, f8)
        run("out-projection", M, 512, 512, 1, f8)

#!/usr/bin/env python
"""Row-count fuzz of the hl32 GEMM's tile selection (csrc/gemm3.hip launch_gemm3): random M in 1 .. 30000 on the main layers'
shapes (out-projection, FF1, FF2) -- what launch_gemm3 picks by itself (x3 = 1) against the 128 x 128 tiles forced (x3 = 3):
the results must agree BIT FOR BIT (every configuration issues the same MFMAs on the same operand pieces in the same k order),
including ragged last tiles, and rows beyond M must stay untouched.
    python tools/gemm3_fuzz.py [n_cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L  # noqa: E402
from gpu_util import pad_rows, to_hl32  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
GUARD = 64   # rows beyond M that must keep their fill value


def run(M, K, N, epi, force, A, W, seed):
    g = torch.Generator().manual_seed(seed)
    a = L.Gemm3Args()
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi, a.x3 = A.data_ptr(), K, M, K, W.data_ptr(), N, epi, force
    keep = []
    if epi == 0:   # FF1: bias + GELU -> hl32 hidden activation
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.full((M + GUARD, 2 * N), 7.0, dtype=torch.float16, device=dev)
        ssq = (torch.rand((K // 64, M), generator=g) + 0.5).to(dev)
        a.bias, a.out, a.ldo, a.ssq_in, a.ssq_parts = bias.data_ptr(), out.data_ptr(), N, ssq.data_ptr(), K // 64
        keep += [bias, ssq]
        res = [out]
    else:          # residual GEMM: x += A W^T, hl32 shadow and statistics of the new x
        x = torch.randn((M + GUARD, N), generator=g).to(dev)
        xb = torch.full((M + GUARD, 2 * N), 7.0, dtype=torch.float16, device=dev)
        ssq = torch.full((N // 64, M + GUARD), 7.0, device=dev)
        a.x, a.ldx, a.xb, a.ssq_out = x.data_ptr(), N, xb.data_ptr(), ssq.data_ptr()
        res = [x, xb]
        keep += [ssq]
        res_ssq = ssq
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    a.status = status.data_ptr()
    L.check(lib.bt_gemm3(L.stream_ptr(dev), C.byref(a)))
    torch.cuda.synchronize()
    if epi != 0:
        # (the statistics buffer is [parts][M] with ld = M: compare the live part only)
        res.append(res_ssq.view(-1)[: (N // 64) * M].clone())
    return res


bad = 0
shapes = [("out-projection", 512, 512, 1), ("FF1", 512, 2048, 0), ("FF2", 2048, 512, 1)]
for c in range(n_cases):
    name, K, N, epi = shapes[c % 3]
    M = int(rng.integers(1, 30001)) if c % 5 else int(rng.choice([1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 4095, 4096, 4097, 8191,
                                                                  8192, 15168, 15169, 19999, 20000, 20001]))
    g = torch.Generator().manual_seed(100 + c)
    A = to_hl32(torch.randn((M, K), generator=g)).to(dev)
    W = to_hl32(pad_rows(torch.randn((N, K), generator=g) / K ** 0.5, 256)).to(dev)
    auto = run(M, K, N, epi, 1, A, W, c)
    sx = run(M, K, N, epi, 3, A, W, c)
    same = all(torch.equal(p, q) for p, q in zip(auto, sx))
    guard_ok = True
    for t in auto[:2] if epi else auto[:1]:
        tail = t[M:]
        if t.dtype == torch.float16:
            guard_ok = guard_ok and bool((tail == 7.0).all())
    finite = all(bool(torch.isfinite(t.float()).all()) for t in auto)
    ok = same and guard_ok and finite
    bad += not ok
    print(f"{c:3d} {name:15s} M = {M:6d}: auto == 128 x 128 tiles {same}, rows beyond M untouched {guard_ok}, finite {finite}", flush=True)
print(f"gemm3 fuzz: {n_cases} cases, {bad} bad")
sys.exit(1 if bad else 0)

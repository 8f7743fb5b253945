"""Time bt_gemm for main-layer shapes while sweeping K: separates k-loop cost from fixed cost."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from beat_this_amd import _lib as L  # noqa: E402
from gpu_util import run_gemm  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


M = 24000
for name, N, epi, flags, a_f32, out_dt in [
    ("ff2/resid", 512, L.GEMM_EPI_RESID, L.GEMM_F_BIAS, False, None),
    ("ff1/store-gelu", 2048, L.GEMM_EPI_STORE, L.GEMM_F_RMS | L.GEMM_F_A_F32 | L.GEMM_F_BIAS | L.GEMM_F_GELU, True, torch.bfloat16),
    ("store-plain-bf16A", 2048, L.GEMM_EPI_STORE, L.GEMM_F_BIAS, False, torch.bfloat16),
    ("store-f32out", 512, L.GEMM_EPI_STORE, L.GEMM_F_BIAS | L.GEMM_F_OUT_F32, False, torch.float32),
]:
    for K in (64, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device=dev)
        if not a_f32:
            A = A.to(torch.bfloat16)
        Npad = (N + 127) // 128 * 128
        W = (torch.randn(Npad, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        x = torch.zeros(M, N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=out_dt) if out_dt is not None else None
        t = timeit(lambda: run_gemm(1, A, W, N, epi, flags, bias=bias, out=out, x=x if epi == L.GEMM_EPI_RESID else None, sync=False))
        print(f"{name:20s} M={M} N={N} K={K:5d}: {t:8.1f} us  {2 * M * N * K / t / 1e6:8.1f} TFLOP/s", flush=True)

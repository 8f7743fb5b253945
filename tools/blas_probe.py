"""Reference point only (not used by the product): what the vendor GEMM library reaches on the main-layer shapes."""
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(s, generator=g).to(torch.bfloat16).to(dev)
for name, M, K, N in (("ff1", 24000, 512, 2048), ("ff2", 24000, 2048, 512), ("out", 24000, 512, 512), ("qkv", 24000, 512, 1552)):
    A, W = rnd(M, K), rnd(N, K)
    for _ in range(3): C = A @ W.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C = A @ W.T
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"vendor GEMM {name} M={M} K={K} N={N}: {us:7.1f} us  {2.0*M*K*N/us/1e6:7.1f} TFLOP/s (plain bf16 GEMM, no epilogue)")

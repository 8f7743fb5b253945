#!/usr/bin/env python
"""Development tool (GPU): does a pinned host-to-device copy on its own stream overlap kernels on another stream here?"""
import time

import torch

dev = torch.device("cuda:0")
n = 6
host = [torch.randn(13_230_000).pin_memory() for _ in range(n)]
copy = torch.cuda.Stream(dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)


def compute(k=12):
    x = a
    for _ in range(k):
        x = (x @ a) * 1e-2
    return x


def upload():
    with torch.cuda.stream(copy):
        out = [h.to(dev, non_blocking=True) for h in host]
    return out


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


t_c = timeit(compute)
t_u = timeit(upload)
t_b = timeit(lambda: (upload(), compute()))
print(f"compute {t_c:.2f} ms, upload {t_u:.2f} ms ({sum(h.numel() for h in host) * 4 / t_u / 1e6:.1f} GB/s), both enqueued together {t_b:.2f} ms "
      f"(sum {t_c + t_u:.2f}, max {max(t_c, t_u):.2f})")
t0 = time.perf_counter()
upload()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"enqueue of the 6 copies returned after {1e3 * (t1 - t0):.2f} ms (blocking if close to the upload time)")

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY' &
import os, sys, time, ctypes as C
sys.path.insert(0, ".")
import torch
from beat_this_amd import _lib
dev = torch.device("cuda:0")
M, D = 24000, 512
g = torch.Generator().manual_seed(0)
def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)
h, W, b = rnd(M, 4 * D), rnd(D, 4 * D, scale=0.02), rnd(D, dtype=torch.float32)
xf = torch.randn((M, D), generator=g).to(dev)
xb, so = torch.empty((M, D), dtype=torch.bfloat16, device=dev), torch.empty((D // 64, M), device=dev)
a = _lib.Gemm3Args()
a.A, a.lda, a.M, a.K, a.W, a.N, a.epi = h.data_ptr(), 4 * D, M, 4 * D, W.data_ptr(), D, 1
a.bias, a.x, a.ldx, a.xb, a.ssq_out = b.data_ptr(), xf.data_ptr(), D, xb.data_ptr(), so.data_ptr()
st = _lib.stream_ptr(dev)
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(300):
        _lib.lib().bt_gemm3(st, C.byref(a))
    torch.cuda.synchronize(); n += 300
print("ff2 launches:", n, "avg us", (time.time() - t0) / n * 1e6)
PY
sleep 3.5
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" | head -4
sleep 1
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" | head -4
wait

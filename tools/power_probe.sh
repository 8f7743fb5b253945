#!/bin/bash
# Sample rocm-smi power / clocks while a GPU workload loops in the background.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -v "^$" | head -30
echo "=== attention loop"
python - <<'PY' &
import os, sys, time, ctypes as C
sys.path.insert(0, ".")
import torch
from beat_this_amd import _lib
dev = torch.device("cuda:0")
L = 1500
nbp = _lib.lib().bt_attn_frag_blocks(L)
SH = 512
g = torch.Generator().manual_seed(0)
q = (torch.randn((SH, nbp, 1024), generator=g) * 0.6).to(torch.bfloat16).to(dev)
k = torch.randn((SH, nbp, 1024), generator=g).to(torch.bfloat16).to(dev)
v = torch.randn((SH, nbp, 1024), generator=g).to(torch.bfloat16).to(dev)
gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
out = torch.zeros((SH * L, 32), dtype=torch.bfloat16, device=dev)
a = _lib.AttnFragArgs()
a.q, a.k, a.v, a.gates, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), gates.data_ptr(), out.data_ptr()
a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div = SH, L, 1, 32, nbp, 1
a.o_outer, a.o_inner, a.o_tok = L, 0, 1
st = _lib.stream_ptr(dev)
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        _lib.lib().bt_attention_frag(st, C.byref(a))
    torch.cuda.synchronize()
    n += 200
print("attention launches:", n, "avg us", (time.time() - t0) / n * 1e6)
PY
sleep 3.5
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk|fclk" | head
sleep 1
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" | head
wait
echo "=== bench loop"
python bench.py --steps 600 --warmup 3 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
sleep 7
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" | head
sleep 1
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" | head
wait
cat /tmp/b.json | cut -c1-200
rocm-smi --showmaxpower 2>&1 | grep -iE "power" | head -3

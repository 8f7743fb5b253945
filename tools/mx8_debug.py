"""debug of csrc/gemm_mx8.hip on the GPU box: which part of the operand / scale layout hypothesis is off"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from beat_this_amd import _lib as L
from test_gpu_mx8 import mx_quantise
dev = torch.device("cuda:0")
M, K, N = 128, 512, 128
g = torch.Generator().manual_seed(1)

def run(ab, asc, wb, wsc):
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    d = [t.to(dev) for t in (ab, asc, wb, wsc)]
    L.check(L.lib().bt_gemm_mx8(L.stream_ptr(dev), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), out.data_ptr(), M, N, K, N))
    torch.cuda.synchronize()
    return out.double().cpu()

def deq(b, s):
    return (b.view(torch.float8_e4m3fn).float().double().view(b.shape[0], K // 32, 32) * torch.exp2(s.double() - 127)[..., None]).view(b.shape[0], K)

a = torch.randn((M, K), generator=g); w = torch.randn((N, K), generator=g)
ab, asc, _ = mx_quantise(a); wb, wsc, _ = mx_quantise(w)
def rel(x, y): return float((x - y).abs().max() / y.abs().max())
# 1. unit scales
u_a, u_w = torch.full_like(asc, 127), torch.full_like(wsc, 127)
print("unit scales:", rel(run(ab, u_a, wb, u_w), deq(ab, u_a) @ deq(wb, u_w).T))
# 2. uniform non-unit scales
s_a, s_w = torch.full_like(asc, 125), torch.full_like(wsc, 130)
print("uniform 125 / 130:", rel(run(ab, s_a, wb, s_w), deq(ab, s_a) @ deq(wb, s_w).T))
# 3. A scales vary per row only
s_a = (120 + torch.arange(M) % 12).to(torch.uint8)[:, None].expand(M, K // 32).contiguous()
print("A per row:", rel(run(ab, s_a, wb, u_w), deq(ab, s_a) @ deq(wb, u_w).T))
s_w = (120 + torch.arange(N) % 12).to(torch.uint8)[:, None].expand(N, K // 32).contiguous()
print("W per row:", rel(run(ab, u_a, wb, s_w), deq(ab, u_a) @ deq(wb, s_w).T))
# 4. A scales vary per k-block only
s_a = (120 + torch.arange(K // 32) % 12).to(torch.uint8)[None, :].expand(M, K // 32).contiguous()
print("A per block:", rel(run(ab, s_a, wb, u_w), deq(ab, s_a) @ deq(wb, u_w).T))
for j in range(4):   # one block scaled
    s_a = torch.full_like(asc, 127); s_a[:, j] = 131
    got, ref = run(ab, s_a, wb, u_w), deq(ab, s_a) @ deq(wb, u_w).T
    # which block did the hardware scale?  fit: out = sum_b c_b * partial_b
    parts = [deq(ab, u_a)[:, 32 * b: 32 * b + 32] @ deq(wb, u_w)[:, 32 * b: 32 * b + 32].T for b in range(K // 32)]
    base = sum(parts)
    diffs = [rel(got, base + 15.0 * parts[b]) for b in range(K // 32)]
    print(f"block {j} x16: err {rel(got, ref):.3g}; best matching block {min(range(K // 32), key=lambda b: diffs[b])} ({min(diffs):.2g})")
print("full:", rel(run(ab, asc, wb, wsc), deq(ab, asc) @ deq(wb, wsc).T))

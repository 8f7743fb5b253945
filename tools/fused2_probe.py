"""Where a workgroup of the frontend's register-chained halves spends its life: bt_attnff_fused / bt_outff_fused
in the hi + lo precision at the 16-chunk scale, each looped alone, plus the per-wave phase clocks of attnff_fused_kernel from a
-DBT_DEV build (tools/build_variant.py dev -DBT_DEV; BT_DEV=1 BT_LIB_PATH=tools/variants/lib_dev.so).  Development tool."""
import ctypes as Ct
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from beat_this_amd import _lib as L
from beat_this_amd.pack import PackedPair
from beat_this_amd.tables import rope_table
from test_gpu_frag import _pair_sd

dev = torch.device("cuda:0")
freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
rope = torch.from_numpy(rope_table(freqs)).to(dev)
st = L.stream_ptr(dev)
PREC = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C in (32, 64, 128):
    sd = _pair_sd(C, 5 + C)
    pp = PackedPair(sd, "a.", "f.", C, dev)
    M = 16 * 1500 * 1024 // C
    x = (torch.randn((M, C)) * 1.5).to(dev)
    ao = torch.randn((M, C)).to(torch.float32 if PREC != 1 else torch.float16).to(dev)
    t1 = timeit(lambda: L.check(L.lib().bt_outff_fused(st, PREC, Ct.byref(pp.weights), ao.data_ptr(), x.data_ptr(), M, 0)))
    x = (torch.randn((M, C)) * 1.5).to(dev)
    t2 = timeit(lambda: L.check(L.lib().bt_attnff_fused(st, PREC, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M)))
    gb = M * C * 4 * 2 / 1e9
    print(f"prec {PREC} C={C}: outff {t1:7.1f} us ({(gb + M * C * 4 / 1e9) / t1 * 1e6:.0f} GB/s algorithmic)   attnff {t2:7.1f} us ({gb / t2 * 1e6:.0f} GB/s)")
    if hasattr(L.lib(), "bt_debug_fused2_buffer"):
        nw = (M + 127) // 128 * 4
        dbg = torch.zeros((nw * 6,), dtype=torch.int64, device=dev)
        L.lib().bt_debug_fused2_buffer(Ct.c_void_p(dbg.data_ptr()))
        L.check(L.lib().bt_attnff_fused(st, PREC, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M))
        torch.cuda.synchronize()
        L.lib().bt_debug_fused2_buffer(Ct.c_void_p(0))
        d = dbg.view(-1, 6).cpu().double()
        print(f"   attnff per wave (shader clocks): entry -> ring start {d[:,0].mean():.0f}, attention steps {d[:,1].mean():.0f}, "
              f"FF tail {d[:,2].mean():.0f}; of which vmcnt wait {d[:,3].mean():.0f}, barrier wait {d[:,4].mean():.0f}; total {d[:, :3].sum(1).mean():.0f}")

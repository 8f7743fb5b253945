"""Time the fused frontend halves (bt_outff_fused, bt_attnff_fused) at the final0 / 16-chunk scale, bf16."""
import ctypes as Ct, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from beat_this_amd import _lib as L
from beat_this_amd.pack import PackedPair
from beat_this_amd.tables import rope_table
from test_gpu_frag import _pair_sd, _mk
dev = torch.device("cuda:0")
freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
rope = torch.from_numpy(rope_table(freqs)).to(dev)
st = L.stream_ptr(dev)


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
    ao = torch.randn((M, C)).to(torch.bfloat16).to(dev)
    t1 = timeit(lambda: L.check(L.lib().bt_outff_fused(st, 1, Ct.byref(pp.weights), ao.data_ptr(), x.data_ptr(), M)))
    x = (torch.randn((M, C)) * 1.5).to(dev)
    t2 = timeit(lambda: L.check(L.lib().bt_attnff_fused(st, 1, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M)))
    print(f"F2_ABL={os.environ.get('BT_F2_ABL', '0')} C={C}: outff {t1:7.1f} us   attnff {t2:7.1f} us")
    if os.environ.get("BT_F2_TIMING"):
        nw = (M + 127) // 128 * 4
        dbg = torch.zeros((nw * 6,), dtype=torch.int64, device=dev)
        L.lib().bt_debug_fused2_buffer(Ct.c_void_p(dbg.data_ptr()))
        L.check(L.lib().bt_attnff_fused(st, 1, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M))
        torch.cuda.synchronize()
        L.lib().bt_debug_fused2_buffer(Ct.c_void_p(0))
        d = dbg.view(-1, 6).cpu().double()
        print(f"   attnff per wave (shader clocks): entry -> ring start {d[:,0].mean():.0f}, attention steps {d[:,1].mean():.0f}, "
              f"FF tail {d[:,2].mean():.0f}; of which vmcnt wait {d[:,3].mean():.0f}, barrier wait {d[:,4].mean():.0f}")

"""Development: the fix-up launch of the x3 attention (attn_fix_x3_kernel) alone -- a main-layer launch of the benchmark's slice
(33 x 16 pairs, 1500 frames) with no, few and many overflowing queries; tools/attn_fix_probe.sh adds the kernel trace.
profiles/r05_fixup_ab.txt."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L
from gpu_util import frag_x3
dev = torch.device("cuda:0")
lib = L.lib()
n_seq, heads, T = 33, 16, 1500
SH = n_seq * heads
nbp = lib.bt_attn_frag_blocks(T)
g = torch.Generator().manual_seed(1)
q = torch.randn((SH, T, 32), generator=g) * 0.5
k = torch.randn((SH, T, 32), generator=g)
v = torch.randn((SH, T, 32), generator=g)
gates = torch.ones((SH, nbp * 32))
def case(name, pairs, per_pair):
    q2, k2 = q.clone(), k.clone()
    for s in range(pairs):
        k2[s, T - 40] = 0.0; k2[s, T - 40, 0] = 24.0
        for j in range(per_pair):
            qi = (37 * j + 5) % T
            q2[s, qi] = 0.0; q2[s, qi, 0] = 25.0
    qd, kd, vd = frag_x3(q2, nbp, "qk").to(dev), frag_x3(k2, nbp, "qk").to(dev), frag_x3(v, nbp, "v").to(dev)
    out = torch.zeros((n_seq * T, 2 * heads * 32), dtype=torch.float16, device=dev)
    scratch = torch.zeros((SH, nbp), dtype=torch.int32, device=dev)
    gd = gates.to(dev)
    a = L.AttnFragArgs()
    a.q, a.k, a.v, a.gates, a.out = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), gd.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div, a.o_outer, a.o_inner, a.o_tok = n_seq, T, heads, heads * 32, nbp, 1, T, 0, 1
    a.x3, a.out_f32, a.status, a.scratch = 13, 0, 0, scratch.data_ptr()
    st = L.stream_ptr(dev)
    for _ in range(3): L.check(lib.bt_attention_frag(st, C.byref(a)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.check(lib.bt_attention_frag(st, C.byref(a)))
    e1.record(); torch.cuda.synchronize()
    nbits = int(sum(bin(w & 0xffffffff).count("1") for w in scratch.cpu().flatten().tolist()))
    print(f"{name:44s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per call (main + fix-up), {nbits} overflowed queries, finite {bool(torch.isfinite(out.float()).all())}", flush=True)
case("no overflow", 0, 0)
case("1 query in 1 pair", 1, 1)
case("1 query in every pair", SH, 1)
case("20 queries in every pair", SH, 20)
case("100 queries in every pair", SH, 100)
case("1400 queries in 8 pairs", 8, 1400)

#!/usr/bin/env python
"""Time the BT_PREC_F32X3 kernels in isolation at the final0 / 16-chunk launch shapes (GPU box):
  * attention variants of csrc/attn2.hip (bt_attn_frag_args.x3 = 1 / 2 / 3) on the main-layer and frontend shapes, with
    a bit-for-bit comparison of the variants' results;
  * the hl32 GEMM of csrc/gemm3.hip on the main-layer shapes (QKV, out-projection, FF1, FF2) and frontend.linear.
    python tools/x3_probe.py [chunks] [attn]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L  # noqa: E402
from gpu_util import frag_x3, pad_rows, to_hl32  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = L.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def attention(n_seq, heads, Lq, label):
    g = torch.Generator().manual_seed(1)
    SH = n_seq * heads
    nbp = lib.bt_attn_frag_blocks(Lq)
    # (random operands straight in the block layout: every 16-byte entry is an arbitrary hi / lo pair, like real data)
    mk = lambda s: (torch.randn((SH, nbp, 2, 1024), generator=g) * s).to(torch.float16)  # noqa: E731
    q, k, v = mk(0.6), mk(1.0), mk(1.0)
    q[:, :, 1] *= 2.0 ** -11
    k[:, :, 1] *= 2.0 ** -11
    v[:, :, 1] *= 2.0 ** -11
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
    outs = {}
    for variant in (1, 2, 5):
        out = torch.zeros((n_seq * Lq, 2 * heads * 32), dtype=torch.float16, device=dev)
        a = L.AttnFragArgs()
        a.q, a.k, a.v, a.gates, a.out = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), gates.data_ptr(), out.data_ptr()
        a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div, a.o_outer, a.o_inner, a.o_tok = n_seq, Lq, heads, heads * 32, nbp, 1, Lq, 0, 1
        a.x3, a.out_f32, a.status = variant, 0, 0
        scratch = torch.zeros((SH, nbp), dtype=torch.int32, device=dev)
        a.scratch = scratch.data_ptr()
        st = L.stream_ptr(dev)
        us = timeit(lambda: L.check(lib.bt_attention_frag(st, C.byref(a))))
        flop = 2 * 2 * SH * Lq * Lq * 32
        outs[variant] = out.clone()
        print(f"attention x3 variant {variant} {label}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s algorithmic "
              f"({3 * flop / us / 1e6:7.1f} on the matrix pipe)  same as variant 1: {torch.equal(outs[variant], outs[1])}", flush=True)


def gemm(M, K, N, epi, label, heads=0, n_seq=0, Lq=0, force=1, nsplit=0):
    g = torch.Generator().manual_seed(2)
    A = to_hl32(torch.randn((M, K), generator=g)).to(dev)
    W = to_hl32(pad_rows(torch.randn((N, K), generator=g) / K ** 0.5, 256)).to(dev)
    a = L.Gemm3Args()
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi, a.x3 = A.data_ptr(), K, M, K, W.data_ptr(), N, epi, force | (nsplit << 4)
    keep = []
    if epi == 0:
        bias = torch.zeros(N, device=dev); out = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev)
        ssq = torch.ones((K // 64, M), device=dev)
        a.bias, a.out, a.ldo, a.ssq_in, a.ssq_parts = bias.data_ptr(), out.data_ptr(), N, ssq.data_ptr(), K // 64
        keep += [bias, out, ssq]
    elif epi == 1:
        x = torch.zeros((M, N), device=dev); xb = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev)
        ssq = torch.zeros((N // 64, M), device=dev)
        a.x, a.ldx, a.xb, a.ssq_out = x.data_ptr(), N, xb.data_ptr(), ssq.data_ptr()
        keep += [x, xb, ssq]
    else:
        from beat_this_amd.tables import rope_table
        nbp = lib.bt_attn_frag_blocks(Lq)
        SH = n_seq * heads
        qf = torch.zeros((SH, nbp, 2, 1024), dtype=torch.float16, device=dev)
        kf, vf = qf.clone(), qf.clone()
        gh = torch.zeros((SH, nbp * 32), device=dev)
        rope = torch.from_numpy(rope_table(10000.0 ** (-torch.arange(0, 32, 2).float() / 32))).to(dev)
        ssq = torch.ones((K // 64, M), device=dev); bg = torch.zeros(heads, device=dev)
        a.ssq_in, a.ssq_parts, a.n_seq, a.L, a.nbp, a.heads, a.rope = ssq.data_ptr(), K // 64, n_seq, Lq, nbp, heads, rope.data_ptr()
        a.qf, a.kf, a.vf, a.gates, a.b_gates = qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gh.data_ptr(), bg.data_ptr()
        keep += [qf, kf, vf, gh, rope, ssq, bg]
    st = L.stream_ptr(dev)
    us = timeit(lambda: L.check(lib.bt_gemm3(st, C.byref(a))))
    flop = 2.0 * M * K * a.N
    print(f"gemm3 x3 {label} [{ {1: 'auto', 2: '256-row tiles', 3: '128-row tiles', 4: '256 x 128 k16 tiles', 5: '64 x 128 tiles'}[force] }, W over {nsplit or 'auto'} XCD groups] M={M} K={K} N={a.N}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s algorithmic "
          f"({3 * flop / us / 1e6:7.1f} on the matrix pipe)", flush=True)


T = 1500
if len(sys.argv) > 2 and sys.argv[2].startswith("loop"):
    # python tools/x3_probe.py 16 loop:5 [seconds] -- one attention variant back to back for a few seconds (power / clock sampling
    # from outside with rocm-smi)
    import time
    want = int(sys.argv[2].split(":")[1])
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    g = torch.Generator().manual_seed(1)
    SH, nbp = B * 32, lib.bt_attn_frag_blocks(T)
    mk = lambda s: (torch.randn((SH, nbp, 2, 1024), generator=g) * s).to(torch.float16)  # noqa: E731
    q, k, v = mk(0.6), mk(1.0), mk(1.0)
    for t in (q, k, v):
        t[:, :, 1] *= 2.0 ** -11
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
    out = torch.zeros((SH * T, 64), dtype=torch.float16, device=dev)
    a = L.AttnFragArgs()
    a.q, a.k, a.v, a.gates, a.out = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div, a.o_outer, a.o_inner, a.o_tok = SH, T, 1, 32, nbp, 1, T, 0, 1
    a.x3, a.out_f32, a.status = want, 0, 0
    scratch = torch.zeros((SH, nbp), dtype=torch.int32, device=dev)
    a.scratch = scratch.data_ptr()
    st = L.stream_ptr(dev)
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for _ in range(20):
            L.check(lib.bt_attention_frag(st, C.byref(a)))
        torch.cuda.synchronize()
        n += 20
    el = time.time() - t0
    print(f"attention x3 variant {want}, {SH} sequences: {n} launches in {el:.2f} s = {el / n * 1e6:.1f} us per launch, "
          f"{3 * 2 * 2 * SH * T * T * 32 / (el / n) / 1e12:.0f} TFLOP/s on the matrix pipe", flush=True)
    sys.exit(0)
attention(B, 16, T, f"main layer ({B} chunks x 16 heads)")
attention(B * 32, 1, T, f"frontend block 0 ({B * 32} sequences x 1 head)")
attention(B * 8, 4, T, f"frontend block 2 ({B * 8} sequences x 4 heads)")
if len(sys.argv) > 2 and sys.argv[2] == "attn":
    sys.exit(0)
M = B * T
if len(sys.argv) > 2 and sys.argv[2].startswith("gloop"):
    # python tools/x3_probe.py 16 gloop:ff1|ff2|qkv|out [seconds] -- one GEMM back to back (power / clock sampling from outside)
    import time
    which = sys.argv[2].split(":")[1]
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    shape = {"ff1": (M, 512, 2048, 0, {}), "ff2": (M, 2048, 512, 1, {}), "out": (M, 512, 512, 1, {}),
             "qkv": (M, 512, 3 * 512 + 16, 2, dict(heads=16, n_seq=B, Lq=T))}[which]
    def timeit(fn, n=20):   # noqa: F811  (the set-up of gemm() once, then launches for `secs` seconds)
        t0, k = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            k += 20
        return (time.time() - t0) / k * 1e6
    gemm(shape[0], shape[1], shape[2], shape[3], which, **shape[4])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "cfgs":
    # python tools/x3_probe.py B cfgs -- every tile configuration on the residual / FF1 shapes of a B-chunk forward (which one should
    # launch_gemm3 pick at this M?  isolated launches, operands warm in L2 / MALL)
    for f in (1, 2, 3, 4, 5):
        gemm(M, 512, 512, 1, "out-projection", force=f)
    for f in (1, 2, 3, 4, 5):
        gemm(M, 2048, 512, 1, "FF2", force=f)
    for f in (1, 2, 3, 4):
        gemm(M, 512, 2048, 0, "FF1", force=f)
    sys.exit(0)
gemm(M, 512, 3 * 512 + 16, 2, "QKV", heads=16, n_seq=B, Lq=T)
for f in (3, 4, 3, 4):
    gemm(M, 512, 512, 1, "out-projection", force=f)
    gemm(M, 512, 2048, 0, "FF1", force=f)
for f in (3, 2):
    gemm(M, 2048, 512, 1, "FF2", force=f)
    gemm(M, 1024, 512, 1, "frontend.linear", force=f)
gemm(M, 2048, 512, 1, "FF2 (auto)")
for ns in (1, 2, 4):
    gemm(M, 512, 2048, 0, "FF1", force=3, nsplit=ns)
    gemm(M, 512, 2048, 0, "FF1", force=2, nsplit=ns)
    gemm(M, 2048, 512, 1, "FF2", force=1, nsplit=ns)
    gemm(M, 1024, 512, 1, "frontend.linear", force=1, nsplit=ns)
    if ns < 4:
        gemm(M, 512, 512, 1, "out-projection", force=3, nsplit=ns)

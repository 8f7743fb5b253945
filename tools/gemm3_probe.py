#!/usr/bin/env python
"""Time bt_gemm3 on the four main-layer shapes of the final0 forward (16 chunks, M = 24000)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from beat_this_amd import _lib

dev = torch.device("cuda:0")
M, D = 24000, 512
g = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0, dtype=torch.float16):
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def timeit(name, a, flop, n=20):
    st = _lib.stream_ptr(dev)
    for _ in range(3):
        _lib.check(_lib.lib().bt_gemm3(st, C.byref(a)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.lib().bt_gemm3(st, C.byref(a))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"ABL={os.environ.get('BT_G3_ABL', '0')} {name}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")


which = sys.argv[1:] or ["ff1", "ff2", "out", "qkv"]
x = rnd(M, D)
ssq = torch.rand((D // 64, M), generator=g).to(dev) * 64
xf = torch.randn((M, D), generator=g).to(dev)
if "ff1" in which:
    a = _lib.Gemm3Args()
    W, b, out = rnd(4 * D, D, scale=0.05), rnd(4 * D, dtype=torch.float32), torch.empty((M, 4 * D), dtype=torch.float16, device=dev)
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi = x.data_ptr(), D, M, D, W.data_ptr(), 4 * D, 0
    a.bias, a.ssq_in, a.ssq_parts, a.out, a.ldo = b.data_ptr(), ssq.data_ptr(), D // 64, out.data_ptr(), 4 * D
    if os.environ.get("BT_G3_ABL") == "8":
        ssq = torch.zeros((8 * 1024 * 1024,), device=dev)  # room for the timing dump
        a.ssq_in = ssq.data_ptr()
    timeit("ff1 (N=2048,K=512)", a, 2.0 * M * D * 4 * D)
    if os.environ.get("BT_G3_ABL") == "8":
        torch.cuda.synchronize()
        nw = 3008 * 4
        d = ssq.view(torch.int64)[: nw * 4].view(-1, 4).cpu().double()
        print(f"   per wave: loop {d[:,0].mean():.0f} cyc (vmcnt wait {d[:,1].mean():.0f}, barrier {d[:,2].mean():.0f}), epilogue {d[:,3].mean():.0f}")
if "ff2" in which:
    a = _lib.Gemm3Args()
    h, W, b = rnd(M, 4 * D), rnd(D, 4 * D, scale=0.02), rnd(D, dtype=torch.float32)
    xb, so = torch.empty((M, D), dtype=torch.float16, device=dev), torch.empty((D // 64, M), device=dev)
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi = h.data_ptr(), 4 * D, M, 4 * D, W.data_ptr(), D, 1
    a.bias, a.x, a.ldx, a.xb, a.ssq_out = b.data_ptr(), xf.data_ptr(), D, xb.data_ptr(), so.data_ptr()
    if os.environ.get("BT_G3_ABL") == "8":
        dbgbuf = torch.zeros((8 * 1024 * 1024,), device=dev)
        a.out = dbgbuf.data_ptr()
    timeit("ff2 (N=512,K=2048)", a, 2.0 * M * D * 4 * D)
    if os.environ.get("BT_G3_ABL") == "8":
        torch.cuda.synchronize()
        nw = 188 * 8 if os.environ.get('BT_G3_BIG') != '0' else 752 * 4
        d = dbgbuf.view(torch.int64)[: nw * 4].view(-1, 4).cpu().double()
        print(f"   per wave: loop {d[:,0].mean():.0f} cyc (vmcnt wait {d[:,1].mean():.0f}, barrier {d[:,2].mean():.0f}), epilogue {d[:,3].mean():.0f}")
if "out" in which:
    a = _lib.Gemm3Args()
    W = rnd(D, D, scale=0.02)
    xb, so = torch.empty((M, D), dtype=torch.float16, device=dev), torch.empty((D // 64, M), device=dev)
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi = x.data_ptr(), D, M, D, W.data_ptr(), D, 1
    a.x, a.ldx, a.xb, a.ssq_out = xf.data_ptr(), D, xb.data_ptr(), so.data_ptr()
    timeit("out (N=512,K=512)", a, 2.0 * M * D * D)
if "qkv" in which:
    from beat_this_amd.tables import rope_table
    a = _lib.Gemm3Args()
    H, L, B = 16, 1500, int(os.environ.get('QKV_B', '16'))
    nbp = _lib.lib().bt_attn_frag_blocks(L)
    W = rnd(3 * D + 128, D, scale=0.05)
    rope = torch.from_numpy(rope_table(10000.0 ** (-torch.arange(0, 32, 2).float() / 32))).to(dev)
    qf = torch.empty((B * H, nbp, 1024), dtype=torch.float16, device=dev)
    kf, vf = torch.empty_like(qf), torch.empty_like(qf)
    gh, bg = torch.empty((B * H, nbp * 32), device=dev), rnd(H, dtype=torch.float32)
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi = x.data_ptr(), D, M, D, W.data_ptr(), 3 * D + H, 2
    a.ssq_in, a.ssq_parts, a.n_seq, a.L, a.nbp, a.heads = ssq.data_ptr(), D // 64, B, L, nbp, H
    a.rope, a.qf, a.kf, a.vf, a.gates, a.b_gates = rope.data_ptr(), qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gh.data_ptr(), bg.data_ptr()
    a.M = B * L
    timeit(f"qkv (N=1552,K=512,B={B})", a, 2.0 * B * L * D * (3 * D + H))
if "ff1x3" in which:   # BT_PREC_F32X3 FF1 (hl32 operands); BT_G3_ABL=8 on a -DBT_DEV build dumps the per-wave timing
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from gpu_util import pad_rows, to_hl32
    a = _lib.Gemm3Args()
    A3 = to_hl32(torch.randn((M, D), generator=g)).to(dev)
    W3 = to_hl32(pad_rows(torch.randn((4 * D, D), generator=g) / D ** 0.5, 256)).to(dev)
    b3, out3 = torch.zeros(4 * D, device=dev), torch.empty((M, 8 * D), dtype=torch.float16, device=dev)
    dbg3 = torch.ones((8 * 1024 * 1024,), device=dev)
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi, a.x3 = A3.data_ptr(), D, M, D, W3.data_ptr(), 4 * D, 0, 3
    a.bias, a.ssq_in, a.ssq_parts, a.out, a.ldo = b3.data_ptr(), dbg3.data_ptr(), D // 64, out3.data_ptr(), 4 * D
    timeit("ff1 x3 (N=2048,K=512)", a, 2.0 * M * D * 4 * D)
    if os.environ.get("BT_G3_ABL") == "8":
        torch.cuda.synchronize()
        nw = 3008 * 4
        d = dbg3.view(torch.int64)[: nw * 4].view(-1, 4).cpu().double()
        print(f"   per wave: loop {d[:,0].mean():.0f} cyc (vmcnt wait {d[:,1].mean():.0f}, barrier {d[:,2].mean():.0f}), epilogue {d[:,3].mean():.0f}")

#!/usr/bin/env python
"""Joules per launch of the big kernels of the default (BT_PREC_F32X3) path, each looped alone for about a second at the
launch shape of the benchmark's forward slice (33 chunks), from the package energy accumulator (tools/smi.py).  VERDICT r4
next-round item 2: "make joules a measured quantity".  GPU box, development tool.

    python tools/energy_probe.py [chunks=33] [seconds=1.0]

Prints one line per kernel (us / launch, W while it loops, J / launch, TFLOP/s on the matrix pipe) and the J-by-category sum
of one benchmark step (66 chunks = two slices) next to what bench.py measured for the whole step, if gpurun_out/ holds one.
"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L  # noqa: E402
from gpu_util import pad_rows, to_hl8, to_hl8a, to_hl32  # noqa: E402
from tools.smi import Smi  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 33
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
T = 1500
lib = L.lib()
smi = Smi(dev)
st = L.stream_ptr(dev)
rows = []


def loop(label, fn, pipe_flop, launches_per_step):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    time.sleep(0.2)
    e0, t0 = smi.energy_uj()
    w0, n = time.time(), 0
    while time.time() - w0 < SECS:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    e1, t1 = smi.energy_uj()
    sec = (t1 - t0) * 1e-9
    j = (e1 - e0) * 1e-6
    row = dict(kernel=label, us=sec / n * 1e6, watts=j / sec, joules=j / n, pipe_tflops=pipe_flop / (sec / n) / 1e12,
               sclk=smi.sclk_mhz(), launches_per_step=launches_per_step)
    rows.append(row)
    print(f"{label:46s} {row['us']:8.1f} us  {row['watts']:7.1f} W  {row['joules']:7.4f} J / launch  {row['pipe_tflops']:7.0f} TFLOP/s on the pipe  "
          f"sclk {row['sclk']}", flush=True)


def attention(SH_seq, heads, variant, label, per_step, out_f32=0):
    g = torch.Generator().manual_seed(1)
    SH = SH_seq * heads
    nbp = lib.bt_attn_frag_blocks(T)
    mk = lambda s: (torch.randn((SH, nbp, 2, 1024), generator=g) * s).to(torch.float16)  # noqa: E731
    q, k, v = mk(0.6), mk(1.0), mk(1.0)
    for t in (q, k, v):
        t[:, :, 1] *= 2.0 ** -11
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
    out = torch.zeros((SH_seq * T, (1 if out_f32 else 2) * heads * 32), dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    a = L.AttnFragArgs()
    a.q, a.k, a.v, a.gates, a.out = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div, a.o_outer, a.o_inner, a.o_tok = SH_seq, T, heads, heads * 32, nbp, 1, T, 0, 1
    a.x3, a.out_f32, a.status = variant, out_f32, 0
    scratch = torch.zeros((SH, nbp), dtype=torch.int32, device=dev)
    a.scratch = scratch.data_ptr()
    mf = (2.5 if variant & 8 else 3.0) * 2 * 2 * SH * T * T * 32
    loop(label, lambda: L.check(lib.bt_attention_frag(st, C.byref(a))), mf, per_step)
    del qd, kd, vd, out, scratch


def gemm(M, K, N, epi, label, per_step, heads=0, n_seq=0, f8=False, cfg=1):
    """f8: hl8 operands, the cross terms on the block-scaled fp8 MFMA (BT_OPT_X3_GEMM_FP8; epi 0 / 1 also WRITE their activation as
    hl8, like the engine's level 2); pipe flops are counted as 3 per product in both forms, so that the columns compare"""
    g = torch.Generator().manual_seed(2)
    A = (to_hl8a if f8 else to_hl32)(torch.randn((M, K), generator=g)).to(dev)
    W = (to_hl8 if f8 else to_hl32)(pad_rows(torch.randn((N, K), generator=g) / K ** 0.5, 256)).to(dev)
    a = L.Gemm3Args()
    a.A, a.lda, a.M, a.K, a.W, a.N, a.epi, a.x3 = A.data_ptr(), K, M, K, W.data_ptr(), N, epi, 1 | ((0x100 | (0x200 if epi < 2 else 0)) if f8 else 0)
    keep = []
    if epi == 0:
        bias = torch.zeros(N, device=dev)
        out = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev)
        ssq = torch.ones((K // 64, M), device=dev)
        a.bias, a.out, a.ldo, a.ssq_in, a.ssq_parts = bias.data_ptr(), out.data_ptr(), N, ssq.data_ptr(), K // 64
        keep += [bias, out, ssq]
    elif epi == 1:
        x = torch.zeros((M, N), device=dev)
        xb = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev)
        ssq = torch.zeros((N // 64, M), device=dev)
        a.x, a.ldx, a.xb, a.ssq_out = x.data_ptr(), N, xb.data_ptr(), ssq.data_ptr()
        keep += [x, xb, ssq]
    else:
        from beat_this_amd.tables import rope_table
        nbp = lib.bt_attn_frag_blocks(T)
        SH = n_seq * heads
        qf = torch.zeros((SH, nbp, 2, 1024), dtype=torch.float16, device=dev)
        kf, vf = qf.clone(), qf.clone()
        gh = torch.zeros((SH, nbp * 32), device=dev)
        rope = torch.from_numpy(rope_table(10000.0 ** (-torch.arange(0, 32, 2).float() / 32))).to(dev)
        ssq = torch.ones((K // 64, M), device=dev)
        bg = torch.zeros(heads, device=dev)
        a.ssq_in, a.ssq_parts, a.n_seq, a.L, a.nbp, a.heads, a.rope = ssq.data_ptr(), K // 64, n_seq, T, nbp, heads, rope.data_ptr()
        a.qf, a.kf, a.vf, a.gates, a.b_gates = qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gh.data_ptr(), bg.data_ptr()
        keep += [qf, kf, vf, gh, rope, ssq, bg]
    loop(label, lambda: L.check(lib.bt_gemm3(st, C.byref(a))), 3 * 2.0 * M * K * a.N, per_step)


# idle package power (one second without work)
torch.cuda.synchronize()
time.sleep(0.5)
e0, t0 = smi.energy_uj()
time.sleep(1.0)
e1, t1 = smi.energy_uj()
idle_w = (e1 - e0) * 1e-6 / ((t1 - t0) * 1e-9)
print(f"idle package power {idle_w:.1f} W, cap {smi.cap_w()} W", flush=True)

M = B * T
S = 2   # forward slices per benchmark step (66 chunks = 2 x 33)
attention(B, 16, 13, "attention, main layer, P16 (default)", 6 * S)
attention(B, 16, 5, "attention, main layer, three-term", 0)
attention(B * 32, 1, 13, "attention, frontend, P16 (default)", 3 * S, out_f32=1)
attention(B * 32, 1, 5, "attention, frontend, three-term", 0, out_f32=1)
gemm(M, 512, 3 * 512 + 16, 2, "QKV + RoPE + gates (gemm3 epi 2)", 6 * S, heads=16, n_seq=B)
gemm(M, 512, 512, 1, "out-projection (gemm3 epi 1)", 6 * S)
gemm(M, 512, 2048, 0, "FF1 + GELU (gemm3 epi 0)", 6 * S)
gemm(M, 2048, 512, 1, "FF2 (gemm3 epi 1)", 6 * S)
gemm(M, 1024, 512, 1, "frontend.linear (gemm3 epi 1)", 1 * S)
if not lib.bt_half_is_bf16():   # BASELINE config 5: the same launches with the cross terms on the block-scaled fp8 MFMA (not in the step's sum)
    gemm(M, 512, 3 * 512 + 16, 2, "QKV + RoPE + gates, fp8 cross terms", 0, heads=16, n_seq=B, f8=True)
    gemm(M, 512, 512, 1, "out-projection, fp8 cross terms", 0, f8=True)
    gemm(M, 512, 2048, 0, "FF1 + GELU, fp8 cross terms", 0, f8=True)
    gemm(M, 2048, 512, 1, "FF2, fp8 cross terms", 0, f8=True)
    if len(sys.argv) > 3:   # tile-configuration study (bt_gemm3_args.x3 & 15: 2 = 256-row tiles forced, 3 = 128 x 128 forced)
        gemm(M, 512, 2048, 0, "FF1, fp8 cross terms, 256 x 256 tiles", 0, f8=True, cfg=2)
        gemm(M, 512, 512, 1, "out-projection, fp8 cross terms, 256 x 256 tiles", 0, f8=True, cfg=2)
        gemm(M, 2048, 512, 1, "FF2, fp8 cross terms, 128 x 128 tiles", 0, f8=True, cfg=3)
        gemm(M, 512, 2048, 0, "FF1, hl32, 128 x 128 tiles", 0, cfg=3)
        gemm(M, 512, 2048, 0, "FF1, hl32, 256 x 256 tiles", 0, cfg=2)
known = sum(r["joules"] * r["launches_per_step"] for r in rows)
known_ms = sum(r["us"] * r["launches_per_step"] for r in rows) * 1e-3
print(f"sum over the launches of one 66-chunk step that were probed here: {known:.2f} J in {known_ms:.2f} ms of kernel time "
      f"(frontend halves, convolutions, stem, head, log-mel, resampler not probed)")
out = dict(chunks=B, idle_W=idle_w, cap_W=smi.cap_w(), rows=rows, probed_J_per_step=known, probed_ms_per_step=known_ms)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "energy_probe.json"), "w"), indent=1)

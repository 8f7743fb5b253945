#!/usr/bin/env python
"""Time bt_attention_frag on the two shapes of the final0 forward (16 chunks):
frontend time direction (512 sequences x 1 head) and main layers (16 sequences x 16 heads), L = 1500."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch

from beat_this_amd import _lib
import ctypes as C

dev = torch.device("cuda:0")
L = 1500
nbp = _lib.lib().bt_attn_frag_blocks(L)
for name, n_seq, heads in (("front", 512, 1), ("main", 16, 16)):
    SH = n_seq * heads
    g = torch.Generator(device="cpu").manual_seed(0)
    q = (torch.randn((SH, nbp, 1024), generator=g) * 0.3).to(torch.float16).to(dev)
    k = torch.randn((SH, nbp, 1024), generator=g).to(torch.float16).to(dev)
    v = torch.randn((SH, nbp, 1024), generator=g).to(torch.float16).to(dev)
    gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
    out = torch.zeros((n_seq * L, heads * 32), dtype=torch.float16, device=dev)
    a = _lib.AttnFragArgs()
    a.q, a.k, a.v, a.gates, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div = n_seq, L, heads, heads * 32, nbp, 1
    a.o_outer, a.o_inner, a.o_tok = L, 0, 1
    st = _lib.stream_ptr(dev)
    for _ in range(3):
        _lib.check(_lib.lib().bt_attention_frag(st, C.byref(a)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        _lib.lib().bt_attention_frag(st, C.byref(a))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flop = 4.0 * SH * L * L * 32
    print(f"ABL={os.environ.get('BT_ATTN_ABL', '0')} {name}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
    if os.environ.get("BT_ATTN_ABL") == "128":
        torch.cuda.synchronize()
        n_w = ((SH + 7) // 8 * 8) * ((47 + 3) // 4) * 4
        d = gates.view(torch.int64).flatten()[: n_w * 4].view(-1, 4).cpu().double()
        print(f"   per-wave pass: {d[:,0].mean():.0f} shader ticks, {d[:,1].mean():.1f} wall ticks (100 MHz) => "
              f"{d[:,0].sum() / d[:,1].sum() * 0.1:.3f} GHz; pass {d[:,1].mean() / 100:.1f} us; entry -> pass "
              f"{d[:,2].mean() / 100:.2f} us; pass start -> key loop {d[:,3].mean() / 100:.2f} us")

#!/usr/bin/env python
"""Development tool (GPU): single-operator determinism at the frontend's real sizes -- each fused kernel is launched
several times on identical inputs and the outputs are compared bit for bit."""
import ctypes as Ct
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beat_this_amd import _lib as L
from beat_this_amd.pack import PackedPair
from beat_this_amd.tables import rope_table

dev = torch.device("cuda:0")
B, T = 16, 1500
reps = 6


def pair_sd(C, seed):
    H = C // 32
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=1.0):
        return torch.randn(*shape, generator=g) * s
    return {"a.norm.gamma": 1 + 0.1 * rn(C), "a.to_qkv.weight": rn(3 * C, C, s=1.6 / math.sqrt(C)),
            "a.to_gates.weight": rn(H, C, s=0.3), "a.to_gates.bias": rn(H, s=0.3),
            "a.to_out.0.weight": rn(C, C, s=1 / math.sqrt(C)),
            "f.net.0.gamma": 1 + 0.1 * rn(C), "f.net.1.weight": rn(4 * C, C, s=1 / math.sqrt(C)),
            "f.net.1.bias": rn(4 * C, s=0.2), "f.net.4.weight": rn(C, 4 * C, s=0.5 / math.sqrt(C)), "f.net.4.bias": rn(C, s=0.2)}


def count_diff(outs, rows_per=None):
    ref = outs[0]
    n = 0
    info = []
    for r, o in enumerate(outs[1:], 1):
        d = (o.float() - ref.float()).abs()
        if float(d.max()) > 0:
            n += 1
            bad_rows = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0)[:, 0]
            info.append((r, int(bad_rows.numel()), int(bad_rows.min()), int(bad_rows.max()), round(float(d.max()), 4)))
    return n, info[:4]


freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
rope = torch.from_numpy(rope_table(freqs)).to(dev)
st = L.stream_ptr(dev)
lib = L.lib()
half_dt = L.half_torch_dtype()
for C in (32, 64, 128):
    F = 1024 // C
    H = C // 32
    M = B * T * F
    pp = PackedPair(pair_sd(C, 100 + C), "a.", "f.", C, dev)
    g = torch.Generator().manual_seed(C)
    x0 = (torch.randn((M, C), generator=g) * 1.5).to(dev)
    for prec in (0, 1):
        dt = torch.float32 if prec == 0 else half_dt
        ao = (torch.randn((M, C), generator=g)).to(dt).to(dev)
        outs = []
        for _ in range(reps):
            x = x0.clone()
            L.check(lib.bt_attnff_fused(st, prec, Ct.byref(pp.weights), rope.data_ptr(), x.data_ptr(), M))
            outs.append(x)
        torch.cuda.synchronize()
        print(f"attnff_fused C={C} prec={prec}: deviating repeats {count_diff(outs)}", flush=True)
        outs = []
        for _ in range(reps):
            x = x0.clone()
            L.check(lib.bt_outff_fused(st, prec, Ct.byref(pp.weights), ao.data_ptr(), x.data_ptr(), M))
            outs.append(x)
        torch.cuda.synchronize()
        print(f"outff_fused  C={C} prec={prec}: deviating repeats {count_diff(outs)}", flush=True)
    # half time-direction QKV + fragment attention
    nbp = lib.bt_attn_frag_blocks(T)
    SH = B * F * H
    qs, os_ = [], []
    for _ in range(reps):
        qf = torch.zeros((SH, nbp, 1024), dtype=half_dt, device=dev)
        kf, vf = torch.zeros_like(qf), torch.zeros_like(qf)
        gh = torch.zeros((SH, nbp * 32), dtype=torch.float32, device=dev)
        L.check(lib.bt_qkv_front(st, Ct.byref(pp.weights), rope.data_ptr(), x0.data_ptr(), B, T, F, qf.data_ptr(), kf.data_ptr(),
                                 vf.data_ptr(), gh.data_ptr(), nbp))
        out = torch.zeros((M, C), dtype=half_dt, device=dev)
        a = L.AttnFragArgs()
        a.q, a.k, a.v, a.gates, a.out = qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gh.data_ptr(), out.data_ptr()
        a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div = B * F, T, H, C, nbp, F
        a.o_outer, a.o_inner, a.o_tok = T * F, 1, F
        L.check(lib.bt_attention_frag(st, Ct.byref(a)))
        qs.append(torch.cat([qf.reshape(SH, -1), kf.reshape(SH, -1), vf.reshape(SH, -1)], 1))
        os_.append(out)
    torch.cuda.synchronize()
    print(f"qkv_front    C={C}: deviating repeats {count_diff(qs)}", flush=True)
    print(f"attn_frag    C={C}: deviating repeats {count_diff(os_)}", flush=True)

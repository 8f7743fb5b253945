"""Scratch: fused2 kernels at model scale (many workgroups per CU), compared with the unfused kernels."""
import ctypes as Ct, math, os, sys
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
for prec in (0, 1):
    for C in (32, 64, 128):
        sd = _pair_sd(C, 5 + C)
        pp = PackedPair(sd, "a.", "f.", C, dev)
        M = 1500 * 1024 // C
        x0 = _mk((M, C), 7 + C, 1.5).float().to(dev)
        # reference: unfused kernels
        xa = x0.clone()
        L.check(L.lib().bt_attn_freq_fused(L.stream_ptr(dev), prec, Ct.byref(pp.weights), rope.data_ptr(), xa.data_ptr(), M))
        L.check(L.lib().bt_ff_fused(L.stream_ptr(dev), prec, Ct.byref(pp.weights), xa.data_ptr(), M))
        for rep in range(3):
            xb = x0.clone()
            L.check(L.lib().bt_attnff_fused(L.stream_ptr(dev), prec, Ct.byref(pp.weights), rope.data_ptr(), xb.data_ptr(), M))
            torch.cuda.synchronize()
            d = (xa - xb).abs()
            bad = (d > 0.05 * xa.abs().max()).sum().item()
            print(f"attnff prec={prec} C={C} M={M} rep={rep}: max diff {d.max().item():.3e} (max |x| {xa.abs().max().item():.2f}), bad elems {bad}, bad rows {int((d.max(1).values > 0.05 * xa.abs().max()).sum())}")

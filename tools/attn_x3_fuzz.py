#!/usr/bin/env python
"""Shape fuzz of the hi + lo attention kernels (csrc/attn2.hip, bt_attention_frag): random (sequences, heads, length) -- ragged
last blocks, single-block sequences, launches on either side of the kernel-selection rule -- through the forward's own choice
(x3 = 4) and the three forced kernels (1: 128-key tiles, 2: 64-key tiles, 5: hand-scheduled two-query-block loop), in the
three-term and in the P16 arithmetic: within one arithmetic all must agree BIT FOR BIT, rows beyond the last query must stay
untouched, and nothing may be left to the range flag on ordinary operands.
    python tools/attn_x3_fuzz.py [n_cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beat_this_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(n_cases):
    heads = int(rng.choice([1, 2, 4, 16]))
    n_seq = int(rng.integers(1, 41)) if heads < 16 else int(rng.integers(1, 9))
    Lq = int(rng.choice([1, 31, 32, 33, 64, 255, 256, 257, 1012, 1488, 1489, 1500])) if c % 3 == 0 else int(rng.integers(1, 1537))
    SH = n_seq * heads
    nbp = lib.bt_attn_frag_blocks(Lq)
    g = torch.Generator().manual_seed(300 + c)
    mk = lambda s: (torch.randn((SH, nbp, 2, 1024), generator=g) * s).to(torch.float16)  # noqa: E731
    q, k, v = mk(0.6), mk(1.0), mk(1.0)
    for t in (q, k, v):
        t[:, :, 1] *= 2.0 ** -11
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    gates = torch.rand((SH, nbp * 32), generator=g).to(dev)
    line = []
    for p16 in (0, 8):
        outs = {}
        for variant in (4, 1, 2, 5):
            out = torch.full((n_seq * Lq + 8, 2 * heads * 32), 7.0, dtype=torch.float16, device=dev)
            status = torch.zeros(4, dtype=torch.int32, device=dev)
            a = L.AttnFragArgs()
            a.q, a.k, a.v, a.gates, a.out = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), gates.data_ptr(), out.data_ptr()
            a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div, a.o_outer, a.o_inner, a.o_tok = n_seq, Lq, heads, heads * 32, nbp, 1, Lq, 0, 1
            a.x3, a.out_f32, a.status = variant | p16, 0, status.data_ptr()
            scratch = torch.zeros((SH, nbp), dtype=torch.int32, device=dev)
            a.scratch = scratch.data_ptr()
            L.check(lib.bt_attention_frag(L.stream_ptr(dev), C.byref(a)))
            torch.cuda.synchronize()
            outs[variant] = (out, int(status[0].item()))
        ref = outs[1][0]
        same = all(torch.equal(outs[vv][0], ref) for vv in (4, 2, 5))
        guard = bool((ref[n_seq * Lq:] == 7.0).all())
        finite = bool(torch.isfinite(ref[: n_seq * Lq].float()).all())
        flags = [outs[vv][1] for vv in (4, 1, 2, 5)]
        ok = same and guard and finite and not any(flags)
        bad += not ok
        line.append(f"{'P16' if p16 else '3-term'}: same {same}, tail untouched {guard}, finite {finite}, flags {flags}")
    print(f"{c:3d} {n_seq:3d} sequences x {heads:2d} heads, L = {Lq:4d}: " + "; ".join(line), flush=True)
print(f"attention x3 fuzz: {n_cases} cases x 2 arithmetics, {bad} bad")
sys.exit(1 if bad else 0)

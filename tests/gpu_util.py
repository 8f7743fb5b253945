"""Helpers for the -m gpu parity tests: thin ctypes wrappers over the single-operator C-ABI
entry points, and a JSONL report written under gpurun_out/ (merged back by gpurun)."""
import ctypes as C
import json
import os

import numpy as np
import torch

from conftest import ROOT

REPORT = os.path.join(ROOT, "gpurun_out", "test_report.jsonl")


def report(name, **vals):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    clean = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(test=name, **clean)) + "\n")


def dev():
    return torch.device("cuda:0")


def tdtype(prec):
    return torch.float32 if prec == 0 else torch.bfloat16


def run_gemm(prec, A, W, N, epi, flags, bias=None, out=None, ldo=0, x=None, conv=None, qkv=None, sync=True):
    """A: device tensor (rows, lda) ; W: (Npad, K) in compute dtype."""
    from beat_this_amd import _lib

    a = _lib.GemmArgs()
    a.A, a.lda, a.W = A.data_ptr(), A.shape[-1], W.data_ptr()
    a.M = conv["M"] if conv else A.shape[0]
    a.N, a.K, a.epi, a.flags = N, W.shape[1], epi, flags
    a.bias = bias.data_ptr() if bias is not None else 0
    if out is not None:
        a.out, a.ldo = out.data_ptr(), ldo or out.shape[-1]
    if x is not None:
        a.x, a.ldx = x.data_ptr(), x.shape[-1]
    if conv:
        a.conv_C2, a.conv_T, a.conv_F = conv["C2"], conv["T"], conv["F"]
    if qkv:
        a.gates, a.inner, a.heads = qkv["gates"].data_ptr(), qkv["inner"], qkv["heads"]
        a.rope, a.pdiv, a.pmod = qkv["rope"].data_ptr(), qkv["pdiv"], qkv["pmod"]
        a.map_T, a.map_F = qkv.get("map_T", 0), qkv.get("map_F", 0)
    _lib.check(_lib.lib().bt_gemm(_lib.stream_ptr(dev()), prec, C.byref(a)))
    if sync:
        torch.cuda.synchronize()


def run_attn(prec, qkv, gates, out, n_seq, L, heads, o_div=1, o_outer=None, o_inner=0, o_tok=1, small=False):
    from beat_this_amd import _lib

    a = _lib.AttnArgs()
    a.qkv, a.ld, a.gates, a.out = qkv.data_ptr(), qkv.shape[-1], gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.o_div = n_seq, L, heads, heads * 32, o_div
    a.o_outer = L if o_outer is None else o_outer
    a.o_inner, a.o_tok = o_inner, o_tok
    _lib.check(_lib.lib().bt_attention(_lib.stream_ptr(dev()), prec, C.byref(a), int(small)))
    torch.cuda.synchronize()


def pad_rows(w, mult=128):
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    out = torch.zeros((npad, w.shape[1]), dtype=w.dtype)
    out[:n] = w
    return out

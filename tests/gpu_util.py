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


def HALF():
    """torch dtype of the library's half-precision operands (float16, or bfloat16 in a -DBT_HALF_BF16 build)"""
    from beat_this_amd import _lib

    return _lib.half_torch_dtype()


def dev():
    return torch.device("cuda:0")


def tdtype(prec):
    return torch.float32 if prec == 0 else HALF()


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


def run_attn(prec, qkv, gates, out, n_seq, L, heads, o_div=1, o_outer=None, o_inner=0, o_tok=1):
    from beat_this_amd import _lib

    a = _lib.AttnArgs()
    a.qkv, a.ld, a.gates, a.out = qkv.data_ptr(), qkv.shape[-1], gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.o_div = n_seq, L, heads, heads * 32, o_div
    a.o_outer = L if o_outer is None else o_outer
    a.o_inner, a.o_tok = o_inner, o_tok
    _lib.check(_lib.lib().bt_attention(_lib.stream_ptr(dev()), prec, C.byref(a)))
    torch.cuda.synchronize()


def pad_rows(w, mult=128):
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    out = torch.zeros((npad, w.shape[1]), dtype=w.dtype)
    out[:n] = w
    return out


# ---- fragment-major attention operands (csrc/attn2.hip) ---------------------------------------------
def frag_qk(x, nbp):
    """[SH, L, 32] -> half [SH, nbp, 4 (quarter), 32 (token), 8]"""
    SH, L, _ = x.shape
    pad = torch.zeros((SH, nbp * 32, 32), dtype=x.dtype)
    pad[:, :L] = x
    return pad.view(SH, nbp, 32, 4, 8).permute(0, 1, 3, 2, 4).contiguous().to(HALF())


def unfrag_qk(fr, L):
    SH, nbp = fr.shape[:2]
    return fr.view(SH, nbp, 4, 32, 8).permute(0, 1, 3, 2, 4).reshape(SH, nbp * 32, 32)[:, :L]


def frag_v(x, nbp):
    """[SH, L, 32] -> half [SH, nbp, 2 (s), 2 (g), 32 (d), 8]; token = 16 s + 8 (j >> 2) + 4 g + (j & 3)"""
    SH, L, _ = x.shape
    pad = torch.zeros((SH, nbp * 32, 32), dtype=x.dtype)
    pad[:, :L] = x
    t = pad.view(SH, nbp, 2, 2, 2, 4, 32)           # [s][jh][g][jl][d]
    return t.permute(0, 1, 2, 4, 6, 3, 5).contiguous().view(SH, nbp, 2, 2, 32, 8).to(HALF())


def unfrag_v(fr, L):
    SH, nbp = fr.shape[:2]
    t = fr.view(SH, nbp, 2, 2, 32, 2, 4)             # [s][g][d][jh][jl]
    return t.permute(0, 1, 2, 5, 3, 6, 4).reshape(SH, nbp * 32, 32)[:, :L]


def run_attn_frag(qf, kf, vf, gates, out, n_seq, L, heads, nbp, o_div=1, o_outer=None, o_inner=0, o_tok=1, variant=0):
    """variant (bt_attn_frag_args.x3 of the half path): 0 = the kernel the launch size selects, -1 = the one-query-block kernel,
    -2 = two query blocks per wave on the hand-scheduled loop (round 6)"""
    from beat_this_amd import _lib

    a = _lib.AttnFragArgs()
    a.x3 = variant
    a.q, a.k, a.v, a.gates, a.out = qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), gates.data_ptr(), out.data_ptr()
    a.n_seq, a.L, a.heads, a.inner, a.nbp, a.o_div = n_seq, L, heads, heads * 32, nbp, o_div
    a.o_outer = L if o_outer is None else o_outer
    a.o_inner, a.o_tok = o_inner, o_tok
    _lib.check(_lib.lib().bt_attention_frag(_lib.stream_ptr(dev()), C.byref(a)))
    torch.cuda.synchronize()


# ---- BT_PREC_F32X3 operand formats (csrc/gemm3.hip, csrc/attn2.hip) --------------------------------------------------
def to_hl32(x):
    """fp32 [M, K] (K % 32 == 0) -> fp16 [M, 2 K]: per 32 columns the 32 hi halves, then the 32 lo halves (x = hi + lo)."""
    x = x.float()
    m, k = x.shape
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return torch.stack([hi.view(m, k // 32, 32), lo.view(m, k // 32, 32)], 2).reshape(m, 2 * k).contiguous()


def from_hl32(h):
    """fp16 [M, 2 K] hl32 -> float64 [M, K] = hi + lo"""
    m, k2 = h.shape
    t = h.double().view(m, k2 // 64, 2, 32)
    return (t[:, :, 0] + t[:, :, 1]).reshape(m, k2 // 2)


HL8_ACT_SHIFT = 8.0   # the byte sections of an hl8 ACTIVATION carry 2^-3 of what a weight's carry (csrc/common.h: range to 3584)


def to_hl8(x, act=False):
    """fp32 [M, K] (K % 32 == 0) -> the "hl8" operand form of csrc/gemm3.hip (X3 = 2), returned as fp16 [M, 2 K] (128 B per 32
    columns like hl32): 32 hi halves | 32 hi bytes = e4m3(x) | 32 lo bytes = e4m3(2^11 (x - hi)) for a weight; an activation
    (act=True) carries e4m3(x / 8) and e4m3(2^8 (x - hi))."""
    x = x.float()
    m, k = x.shape
    hi = x.to(torch.float16)
    s = HL8_ACT_SHIFT if act else 1.0
    lo = (x - hi.float()) * (2048.0 / s)
    out = torch.empty((m, k // 32, 128), dtype=torch.uint8)
    out[:, :, :64] = hi.contiguous().view(m, k // 32, 32).view(torch.uint8)
    out[:, :, 64:96] = (x / s).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).view(m, k // 32, 32)
    out[:, :, 96:] = lo.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).view(m, k // 32, 32)
    return out.view(m, k // 32 * 128).view(torch.float16).contiguous()


def to_hl8a(x):
    return to_hl8(x, act=True)


def hl8_parts(h, act=False):
    """fp16 [M, 2 K] hl8 -> float64 (hi halves, the value the hi bytes stand for, the value the lo bytes stand for), each [M, K]"""
    m, k2 = h.shape
    s = HL8_ACT_SHIFT if act else 1.0
    b = h.contiguous().view(torch.uint8).view(m, k2 // 64, 128)
    hi = b[:, :, :64].contiguous().view(torch.float16).double().reshape(m, k2 // 2)
    h8 = b[:, :, 64:96].contiguous().view(torch.float8_e4m3fn).double().reshape(m, k2 // 2) * s
    l8 = b[:, :, 96:].contiguous().view(torch.float8_e4m3fn).double().reshape(m, k2 // 2) / 2048.0 * s
    return hi, h8, l8


def hl8_matmul(a, w):
    """what the hl8 GEMM computes (float64): hi . hi + e4m3(a) . lo8(w) + lo8(a) . e4m3(w); a an hl8 activation, w an hl8 weight"""
    ah, a8, al = hl8_parts(a, act=True)
    wh, w8, wl = hl8_parts(w)
    return ah @ wh.T + a8 @ wl.T + al @ w8.T


def frag_x3(x, nbp, kind):
    """[SH, L, 32] fp32-valued -> fp16 [SH, nbp, 2 (hi | lo), 1024]: the 4 KB blocks of the X3 attention operands"""
    f = frag_qk if kind == "qk" else frag_v
    x = x.float()
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    SH = x.shape[0]
    return torch.stack([f(hi.float(), nbp).view(SH, nbp, 1024), f(lo.float(), nbp).view(SH, nbp, 1024)], 2).contiguous()


def unfrag_x3(fr, L, kind):
    """inverse of frag_x3 -> float64 [SH, L, 32]"""
    u = unfrag_qk if kind == "qk" else unfrag_v
    return u(fr[:, :, 0].contiguous(), L).double() + u(fr[:, :, 1].contiguous(), L).double()

"""Narrow formats for the CORRECTION terms of BT_PREC_F32X3 only.  Operand-format study on the CPU
oracle (development tool, imports oracle/: NOT product code), built on tools/prec_study.py.

The hi + lo scheme computes  a.b ~ hi_a.hi_b + hi_a.lo_b + lo_a.hi_b  on three fp16 MFMAs.  The two cross terms are ~2^-11
of the main term, so their operands need far fewer bits than fp16 carries -- and gfx950's block-scaled MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4) runs fp8 operands at 2x and fp6 / fp4 operands at 4x the fp16 rate.  This script asks
what that would cost in accuracy: every matmul / convolution of the forward (QKV, scores, P.V, out-projection, feed-forward,
convolutions, frontend.linear) with

    main term    hi_a . hi_b                        fp16 x fp16, fp32 accumulate (as today)
    cross terms  Q(hi_a) . Q(lo_b) + Q(lo_a) . Q(hi_b)   operands in a narrow MX format: blocks of 32 along k share a
                                                    power-of-two scale (e8m0), elements e4m3 / e5m2 / e3m2 / e2m3 / e2m1

against the fp32 oracle: max / rms logit error, beat / downbeat flips.  "f16" cross terms = the shipped scheme.

    python tools/x3_narrow_study.py [final0|small0] [T] [lively|outlier]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

import prec_study as P
from beat_this_amd import weights as W
from oracle import beat_this_oracle as O

FORMATS = {  # name: (exponent bits, mantissa bits, largest finite value, exponent bias)
    "e4m3": (4, 3, 448.0, 7), "e5m2": (5, 2, 57344.0, 15), "e3m2": (3, 2, 28.0, 3), "e2m3": (2, 3, 7.5, 1), "e2m1": (2, 1, 6.0, 1)}
CROSS = {"fmt": "f16"}


def q_elem(x, fmt):
    """round-to-nearest-even into a small float format (subnormals kept, saturating)"""
    eb, mb, vmax, bias = FORMATS[fmt]
    ax = x.abs().clamp_min(1e-38)
    e = torch.floor(torch.log2(ax)).clamp(min=1 - bias)          # exponent of the binade (subnormals share the lowest)
    q = torch.exp2(e - mb)
    y = torch.round(x / q) * q
    return y.clamp(-vmax, vmax)


def q_mx(x, fmt, dim=-1):
    """MX block format along `dim`: blocks of 32 share a power-of-two scale chosen so that the block maximum lands in the
    format's top binade (OCP MX: scale = 2^(floor(log2(amax)) - emax_elem))"""
    if fmt == "f16":
        return x.to(torch.float16).float()
    eb, mb, vmax, bias = FORMATS[fmt]
    x = x.transpose(dim, -1)
    shp = x.shape
    k = shp[-1]
    pad = (-k) % 32
    xp = F.pad(x, (0, pad)).reshape(*shp[:-1], (k + pad) // 32, 32)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    emax_elem = torch.floor(torch.log2(torch.tensor(vmax)))
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax_elem)
    y = q_elem(xp / scale, fmt) * scale
    return y.reshape(*shp[:-1], k + pad)[..., :k].transpose(dim, -1)


def split(x):
    hi = x.to(torch.float16).float()
    lo = (x - hi).to(torch.float16).float()
    return hi, lo


def mm(a, b, site):
    """a [.., m, k] @ b [.., k, n] on the hi + lo scheme with the cross terms' operands in CROSS['fmt']"""
    ah, al = split(a)
    bh, bl = split(b)
    f = CROSS["fmt"]
    if f == "f16":
        return ah @ bh + (ah @ bl + al @ bh)
    return ah @ bh + (q_mx(ah, f, -1) @ q_mx(bl, f, -2) + q_mx(al, f, -1) @ q_mx(bh, f, -2))


def conv_x3(x, w):
    """the (2,3)/(2,1) convolution as the implicit GEMM the kernels run: unfold -> mm"""
    b, c, fdim, t = x.shape
    cols = F.unfold(x, kernel_size=(2, 3), stride=(2, 1), padding=(0, 1))      # [b, c*2*3, f/2 * t]
    out = mm(cols.transpose(1, 2), w.reshape(w.shape[0], -1).T, "conv")          # [b, L, co]
    return out.transpose(1, 2).reshape(b, w.shape[0], fdim // 2, t)


def attention(x, sd, pfx, heads, tag):
    b, n, dim = x.shape
    xn = O.rmsnorm(x, sd[pfx + "norm.gamma"])
    qkv = mm(xn, sd[pfx + "to_qkv.weight"].T, tag + "qkv")
    d = qkv.shape[-1] // (3 * heads)
    qkv = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    fr = sd[pfx + "rotary_embed.freqs"]
    q, k = O.rope(q, fr), O.rope(k, fr)
    s = mm(q, k.transpose(-1, -2), tag + "qk") * (d ** -0.5)
    p = torch.exp(s - s.amax(-1, keepdim=True))
    out = mm(p, v, tag + "pv") / p.sum(-1, keepdim=True)
    gates = mm(xn, sd[pfx + "to_gates.weight"].T, tag + "qkv") + sd[pfx + "to_gates.bias"]
    out = out * torch.sigmoid(gates).permute(0, 2, 1)[..., None]
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return mm(out, sd[pfx + "to_out.0.weight"].T, tag + "out")


def forward(sd, x):
    x = O.stem(x, sd)
    for i in range(3):
        p = f"frontend.blocks.{i}."
        x = P.partial_ft(x, sd, p + "partial.")
        x = conv_x3(x, sd[p + "conv2d.weight"])
        x = F.gelu(O.batchnorm(x, sd, p + "norm.", 1))
    b, c, f, t = x.shape
    x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)
    x = mm(x, sd["frontend.linear.weight"].T, "lin") + sd["frontend.linear.bias"]
    dim = x.shape[-1]
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer_blocks.layers."))
    for l in range(n_layers):
        p = f"transformer_blocks.layers.{l}."
        x = P.attention(x, sd, p + "0.", dim // 32, "m_") + x
        x = P.feedforward(x, sd, p + "1.", "m_") + x
    x = O.rmsnorm(x, sd["transformer_blocks.norm.gamma"])
    bd = x @ sd["task_heads.beat_downbeat_lin.weight"].T + sd["task_heads.beat_downbeat_lin.bias"]
    return bd[..., 0] + bd[..., 1], bd[..., 1]


P.mm = mm
P.attention = attention


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "final0"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    style = sys.argv[3] if len(sys.argv) > 3 else "lively"
    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=1, style=style)
    torch.set_num_threads(16)
    rows = {}
    with torch.inference_mode():
        for seed in (3, 4, 5):
            x = torch.from_numpy(W.synthetic_spect(T, seed=seed))[None]
            ob, od = O.model_forward(sd, x)
            rb, rd = O.model_forward(sd, x, torch.float64)
            ref_beats, ref_down = O.postp_minimal(ob[0], od[0])
            e32 = float(max((ob - rb).abs().max(), (od - rd).abs().max()))
            rows.setdefault("fp32 oracle vs fp64 oracle (noise floor)", []).append((e32, 0.0, 0, 0, len(ref_beats), len(ref_down)))
            for fmt in ("f16", "e5m2", "e4m3", "e2m3", "e3m2", "e2m1"):
                CROSS["fmt"] = fmt
                b, d = forward(sd, x)
                eb, ed = (b - rb).abs(), (d - rd).abs()
                bt, dt = O.postp_minimal(b[0], d[0])
                fb = len(set(np.round(bt * 50).astype(int)) ^ set(np.round(ref_beats * 50).astype(int)))
                fd = len(set(np.round(dt * 50).astype(int)) ^ set(np.round(ref_down * 50).astype(int)))
                rows.setdefault(fmt, []).append((float(max(eb.max(), ed.max())), float(torch.cat([eb, ed]).pow(2).mean().sqrt()), fb, fd,
                                                 len(ref_beats), len(ref_down)))
                print(f"seed {seed} cross terms {fmt:5s} max {rows[fmt][-1][0]:.2e} rms {rows[fmt][-1][1]:.2e} flips {fb} / {fd}", flush=True)
    print(f"\n{name} T={T} style={style}: three inputs; errors against the float64 oracle, flips against the fp32 oracle's beats")
    print(f"{'cross-term operands':44s} {'max |dlogit|':>12s} {'rms':>10s} {'flips b/d':>10s} {'of':>10s}  MFMA units / product")
    units = {"f16": "3.0", "e5m2": "2.0", "e4m3": "2.0", "e2m3": "1.5", "e3m2": "1.5", "e2m1": "1.5"}
    for k, v in rows.items():
        print(f"{k:44s} {max(r[0] for r in v):12.2e} {np.mean([r[1] for r in v]):10.2e} {sum(r[2] for r in v):4d} /{sum(r[3] for r in v):4d} "
              f"{sum(r[4] for r in v):4d} /{sum(r[5] for r in v):4d}  {units.get(k, '')}")


if __name__ == "__main__":
    main()

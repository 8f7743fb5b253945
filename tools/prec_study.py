"""Operand-rounding study on the CPU oracle (development tool, imports oracle/: NOT product code).

Simulates the half-precision paths of the HIP kernels -- every matmul/conv operand rounded to a given type, fp32
accumulation, fp32 residual stream -- per site, and reports logit error and beat flips against the fp32 oracle.
    python tools/prec_study.py [final0|small0] [T]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from beat_this_amd import weights as W
from oracle import beat_this_oracle as O

SITES = ["conv", "lin", "f_qkv", "f_qk", "f_pv", "f_out", "f_ff1", "f_ff2", "m_qkv", "m_qk", "m_pv", "m_out", "m_ff1", "m_ff2"]
MODE = {s: None for s in SITES}
GELU = {"approx": "none"}


def rnd(x, how):
    if how is None:
        return x
    if how == "bf16":
        return x.to(torch.bfloat16).float()
    if how == "f16":
        return x.to(torch.float16).float()
    if how == "bf16x2":  # hi + lo split: ~16 bits
        hi = x.to(torch.bfloat16).float()
        return hi + (x - hi).to(torch.bfloat16).float()
    if how == "f16x2":
        hi = x.to(torch.float16).float()
        return hi + (x - hi).to(torch.float16).float()
    if how == "e4m3":
        s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
        return (x / s).to(torch.float8_e4m3fn).float() * s
    raise ValueError(how)


def mm(a, b, site):
    h = MODE[site]
    return rnd(a, h) @ rnd(b, h)


def attention(x, sd, pfx, heads, tag):
    b, n, dim = x.shape
    xn = O.rmsnorm(x, sd[pfx + "norm.gamma"])
    # kernels fold gamma and the norm factor differently; operand rounding of xn is what matters
    qkv = mm(xn, sd[pfx + "to_qkv.weight"].T, tag + "qkv")
    d = qkv.shape[-1] // (3 * heads)
    qkv = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    fr = sd[pfx + "rotary_embed.freqs"]
    q, k = O.rope(q, fr), O.rope(k, fr)
    att = torch.softmax(mm(q, k.transpose(-1, -2), tag + "qk") * (d ** -0.5), dim=-1)
    out = mm(att, v, tag + "pv")
    gates = mm(xn, sd[pfx + "to_gates.weight"].T, tag + "qkv") + sd[pfx + "to_gates.bias"]
    out = out * torch.sigmoid(gates).permute(0, 2, 1)[..., None]
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return mm(out, sd[pfx + "to_out.0.weight"].T, tag + "out")


def feedforward(x, sd, pfx, tag):
    h = O.rmsnorm(x, sd[pfx + "net.0.gamma"])
    h = F.gelu(mm(h, sd[pfx + "net.1.weight"].T, tag + "ff1") + sd[pfx + "net.1.bias"], approximate=GELU["approx"])
    return mm(h, sd[pfx + "net.4.weight"].T, tag + "ff2") + sd[pfx + "net.4.bias"]


def partial_ft(x, sd, pfx):
    b, c, f, t = x.shape
    heads = c // 32
    y = x.permute(0, 3, 2, 1).reshape(b * t, f, c)
    y = y + attention(y, sd, pfx + "attnF.", heads, "f_")
    y = y + feedforward(y, sd, pfx + "ffF.", "f_")
    y = y.view(b, t, f, c).permute(0, 2, 1, 3).reshape(b * f, t, c)
    y = y + attention(y, sd, pfx + "attnT.", heads, "f_")
    y = y + feedforward(y, sd, pfx + "ffT.", "f_")
    return y.view(b, f, t, c).permute(0, 3, 1, 2)


def forward(sd, x):
    x = O.stem(x, sd)
    for i in range(3):
        p = f"frontend.blocks.{i}."
        x = partial_ft(x, sd, p + "partial.")
        h = MODE["conv"]
        x = F.conv2d(rnd(x, h), rnd(sd[p + "conv2d.weight"], h), stride=(2, 1), padding=(0, 1))
        x = F.gelu(O.batchnorm(x, sd, p + "norm.", 1), approximate=GELU["approx"])
    b, c, f, t = x.shape
    x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)
    x = mm(x, sd["frontend.linear.weight"].T, "lin") + sd["frontend.linear.bias"]
    dim = x.shape[-1]
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer_blocks.layers."))
    for l in range(n_layers):
        p = f"transformer_blocks.layers.{l}."
        x = attention(x, sd, p + "0.", dim // 32, "m_") + x
        x = feedforward(x, sd, p + "1.", "m_") + x
    x = O.rmsnorm(x, sd["transformer_blocks.norm.gamma"])
    bd = x @ sd["task_heads.beat_downbeat_lin.weight"].T + sd["task_heads.beat_downbeat_lin.bias"]
    return bd[..., 0] + bd[..., 1], bd[..., 1]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "final0"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    style = sys.argv[3] if len(sys.argv) > 3 else "lively"
    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=1, style=style)
    x = torch.from_numpy(W.synthetic_spect(T, seed=3))[None]
    torch.set_num_threads(16)
    with torch.inference_mode():
        rb, rd = forward(sd, x)
        ob, od = O.model_forward(sd, x)
        print("self-check vs oracle", float((rb - ob).abs().max()))
        ref_beats, ref_down = O.postp_minimal(rb[0], rd[0])
        print(f"spread {float(rb.std()):.3f}  beats {len(ref_beats)} downbeats {len(ref_down)}")

        def report(label):
            b, d = forward(sd, x)
            eb, ed = (b - rb).abs(), (d - rd).abs()
            bt, dt = O.postp_minimal(b[0], d[0])
            fb = len(set(np.round(bt * 50).astype(int)) ^ set(np.round(ref_beats * 50).astype(int)))
            fd = len(set(np.round(dt * 50).astype(int)) ^ set(np.round(ref_down * 50).astype(int)))
            print(f"{label:34s} max {float(max(eb.max(), ed.max())):.2e} rms {float(torch.cat([eb, ed]).pow(2).mean().sqrt()):.2e}"
                  f"  flips b {fb} d {fd}", flush=True)

        for how in ("bf16", "f16", "f16x2"):
            for s in SITES:
                MODE[s] = how
            report(f"all {how}")
        for s in SITES:
            MODE[s] = None
        GELU["approx"] = "tanh"
        report("fp32, tanh GELU (FF + conv)")
        for s in SITES:
            MODE[s] = "f16"
        report("all f16, tanh GELU")
        GELU["approx"] = "none"
        for s in SITES:
            MODE[s] = "f16"
        MODE["f_pv"] = MODE["m_pv"] = "bf16"
        report("all f16, P.V bf16")
        if "--quick" in sys.argv:
            return
        # attribution: one site at a time in bf16 and f16
        for how in ("bf16", "f16"):
            for s in SITES:
                for t in SITES:
                    MODE[t] = None
                MODE[s] = how
                report(f"only {s} {how}")
        # mixed candidates
        for s in SITES:
            MODE[s] = "f16"
        for s in ("f_qkv", "m_qkv"):
            MODE[s] = "f16x2"
        report("f16, qkv f16x2")
        for s in ("f_qk", "m_qk"):
            MODE[s] = "f16x2"
        report("f16, qkv+qk f16x2")


if __name__ == "__main__":
    main()

"""Which products of BT_PREC_F32X3 need all three MFMAs?  Operand-rounding study on the CPU oracle (development tool,
imports oracle/: NOT product code), built on tools/prec_study.py: every matmul on hi + lo operands (f16x2) as the baseline,
then cheaper treatments of single sites, with the logit error and beat flips against the fp32 oracle.
    python tools/x3_study.py [final0|small0] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import prec_study as P
from beat_this_amd import weights as W
from oracle import beat_this_oracle as O

OPT = {"p_round": False, "v_hi": False, "k_hi": False, "q_hi": False}


def attention(x, sd, pfx, heads, tag):
    b, n, dim = x.shape
    xn = O.rmsnorm(x, sd[pfx + "norm.gamma"])
    qkv = P.mm(xn, sd[pfx + "to_qkv.weight"].T, tag + "qkv")
    d = qkv.shape[-1] // (3 * heads)
    qkv = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    fr = sd[pfx + "rotary_embed.freqs"]
    q, k = O.rope(q, fr), O.rope(k, fr)
    qq = P.rnd(q, "f16" if OPT["q_hi"] else "f16x2")
    kk = P.rnd(k, "f16" if OPT["k_hi"] else "f16x2")
    s = (qq @ kk.transpose(-1, -2)) * (d ** -0.5)
    p = torch.exp(s - s.amax(-1, keepdim=True))
    p = p.to(torch.float16).float() if OPT["p_round"] else P.rnd(p, "f16x2")   # the SAME p in numerator and denominator
    vv = P.rnd(v, "f16" if OPT["v_hi"] else "f16x2")
    out = (p @ vv) / p.sum(-1, keepdim=True)
    gates = P.mm(xn, sd[pfx + "to_gates.weight"].T, tag + "qkv") + sd[pfx + "to_gates.bias"]
    out = out * torch.sigmoid(gates).permute(0, 2, 1)[..., None]
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return P.mm(out, sd[pfx + "to_out.0.weight"].T, tag + "out")


P.attention = attention


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "final0"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=1, style="lively")
    x = torch.from_numpy(W.synthetic_spect(T, seed=3))[None]
    torch.set_num_threads(16)
    with torch.inference_mode():
        ob, od = O.model_forward(sd, x)
        ref_beats, ref_down = O.postp_minimal(ob[0], od[0])

        def report(label):
            b, d = P.forward(sd, x)
            eb, ed = (b - ob).abs(), (d - od).abs()
            bt, dt = O.postp_minimal(b[0], d[0])
            fb = len(set(np.round(bt * 50).astype(int)) ^ set(np.round(ref_beats * 50).astype(int)))
            fd = len(set(np.round(dt * 50).astype(int)) ^ set(np.round(ref_down * 50).astype(int)))
            print(f"{label:56s} max {float(max(eb.max(), ed.max())):.2e} rms {float(torch.cat([eb, ed]).pow(2).mean().sqrt()):.2e}"
                  f"  flips b {fb} d {fd}", flush=True)

        for s in P.SITES:
            P.MODE[s] = "f16x2"
        report("every product on hi + lo operands")
        for key, label in (("p_round", "... probabilities rounded to fp16 (both sums)"), ("v_hi", "... V hi only"),
                           ("k_hi", "... K hi only"), ("q_hi", "... Q hi only")):
            OPT[key] = True
            report(label)
            OPT[key] = False
        OPT["p_round"] = True
        for s in ("f_pv", "m_pv"):
            pass
        report("probabilities rounded (again, for the record)")


if __name__ == "__main__":
    main()

"""Flip-rate / precision frontier of the default path.  Development tool: imports oracle/,
NOT product code.

Question: how far is each arithmetic scheme from moving a beat, measured over many tracks and not one?  For every
(weight style, track) the CPU oracle is evaluated in fp32 (= the reference's own arithmetic, the yardstick of "identical
beat indices") and in fp64 (the truth both sit around).  The oracle's own fp32-vs-fp64 flips are the NOISE FLOOR: a scheme
whose flips against the fp32 oracle stay within that floor is as identical to the reference as the reference is to itself
across machines / BLAS builds.

    python tools/flip_soak.py oracle [--tracks 48] [--styles lively,outlier,init] [--threads 4]
        CPU (build container): soak_cache/<style>_<k>.npz = fp32 + fp64 framewise logits of the oracle for a 300 s
        22.05 kHz synthetic track (seed 1000 + k) on seeded final0 weights.  soak_cache/ is git-ignored but travels to the
        GPU box with the snapshot.
    python tools/flip_soak.py gpu [--schemes exact,x3,x3p16m,x3p16,x3p16f8ff,x3p16f8,half] [--tracks N]
        GPU box: the real kernels (Audio2Frames.many from the same waveforms) against the cached oracle logits.
    python tools/flip_soak.py sim [--schemes p16,vhi,e4m3,e2m3] [--device cuda|cpu] [--tracks N]
        operand-rounding simulations of schemes that are not built (tools/x3_narrow_study.py's forward, per site).
    python tools/flip_soak.py report
        gpurun_out/flip_soak/*.json (+ soak_cache) -> the table of profiles/r05_flip_frontier.txt.

Flips = symmetric difference of the peak-picked beat (downbeat) frame sets (oracle postp_minimal on every logit vector,
so the post-processor is the same code on both sides), per 1000 beats of the fp32 oracle.  Margin = for every fp32-oracle
peak frame and every near-peak, how close the decision was (see margin_stats) -- the distribution that says how large a
logit error has to be before it can move a beat.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from beat_this_amd import weights as W
from oracle import beat_this_oracle as O

CACHE = os.path.join(ROOT, "soak_cache")
OUT = os.path.join(ROOT, "gpurun_out", "flip_soak")
SECONDS, SR, WEIGHT_SEED, MODEL = 300.0, 22050, 1, "final0"


def track(k):
    return W.synthetic_audio(SECONDS, seed=1000 + k, sr=SR)


def frames_of(logits):
    """peak-picked frame indices (x 2: deduplicated means can be half frames) of one logit vector"""
    return set(np.round(O.deduplicate_peaks(O.peak_frames(torch.as_tensor(logits))) * 2).astype(np.int64))


def flips(a, b):
    return len(frames_of(a) ^ frames_of(b))


def margin_stats(x):
    """How close is each peak decision of logit vector x (fp32 oracle)?  A frame t is a peak iff x[t] == max(x[t-3..t+3]) and
    x[t] > 0.  The decision margin of a peak is min(x[t] - second largest in its window, x[t] - 0); of a non-peak with
    x[t] > 0 that is the runner-up of its window: (window max - x[t]).  A perturbation smaller than half the margin cannot
    flip that decision.  Returns the sorted margins (the frontier's x axis)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    pad = np.concatenate([np.full(3, -np.inf), x, np.full(3, -np.inf)])
    win = np.lib.stride_tricks.sliding_window_view(pad, 7)            # [n, 7]
    others = np.delete(win, 3, axis=1)
    omax = others.max(1)
    is_peak = (x >= omax) & (x > 0)
    m_peak = np.minimum(x - omax, x)[is_peak]                           # peak stays a peak while error < margin / 2
    cand = (~is_peak) & (x > 0)
    m_cand = (omax - x)[cand]                                           # a positive non-peak becomes one
    m_zero = np.abs(x[(x >= omax)])                                     # window maxima near 0 (sign decision)
    return np.sort(np.concatenate([m_peak, m_cand, m_zero]))


def cmd_oracle(args):
    os.makedirs(CACHE, exist_ok=True)
    torch.set_num_threads(args.threads)
    hp = W.resolve_hparams(MODEL)
    sds = {}
    for style in args.styles.split(","):
        sd = W.random_state_dict(hp, seed=WEIGHT_SEED, style=style)
        sds[style] = (sd, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    for k in range(args.first, args.tracks):          # (tracks outermost: an interrupted run covers all styles evenly)
        for style, (sd, sd64) in sds.items():
            path = os.path.join(CACHE, f"{style}_{k:03d}.npz")
            if os.path.exists(path):
                continue
            t0 = time.time()
            with torch.inference_mode():
                spect = O.logmel(torch.from_numpy(track(k)))
                b32, d32 = O.spect2frames(sd, spect)
                b64, d64 = _spect2frames64(sd64, spect)
            np.savez(path + ".tmp.npz", b32=b32.numpy(), d32=d32.numpy(), b64=b64.numpy(), d64=d64.numpy())
            os.replace(path + ".tmp.npz", path)
            print(f"{style} track {k}: {time.time() - t0:.0f} s, fp32-vs-fp64 max {float((b32.double() - b64).abs().max()):.2e}, "
                  f"flips {flips(b32, b64)} / {flips(d32, d64)} of {len(frames_of(b32))} / {len(frames_of(d32))}", flush=True)


def _spect2frames64(sd64, spect):
    """O.spect2frames in float64 with float64 aggregation (O.aggregate rounds to fp32: the truth should not be)"""
    chunks, starts = O.split_chunks(spect)
    n = spect.shape[0]
    beat = torch.full((n,), -1000.0, dtype=torch.float64)
    down = torch.full((n,), -1000.0, dtype=torch.float64)
    preds = [O.model_forward(sd64, c[None], torch.float64) for c in chunks]
    for s, (b, d) in reversed(list(zip(starts, preds))):
        s = int(s)
        beat[s + O.BORDER: s + O.CHUNK - O.BORDER] = b[0][O.BORDER:-O.BORDER]
        down[s + O.BORDER: s + O.CHUNK - O.BORDER] = d[0][O.BORDER:-O.BORDER]
    return beat, down


def cached(style, k):
    z = np.load(os.path.join(CACHE, f"{style}_{k:03d}.npz"))
    return z["b32"], z["d32"], z["b64"], z["d64"]


def available(style, limit):
    ks = sorted(int(os.path.basename(p).split("_")[1][:3]) for p in glob.glob(os.path.join(CACHE, f"{style}_*.npz")))
    return [k for k in ks if k < limit]


def score(name, style, k, b, d, rows):
    b32, d32, b64, d64 = cached(style, k)
    b, d = np.asarray(b, dtype=np.float64), np.asarray(d, dtype=np.float64)
    e32 = max(np.abs(b - b32).max(), np.abs(d - d32).max())
    e64 = max(np.abs(b - b64).max(), np.abs(d - d64).max())
    rms = float(np.sqrt(np.mean(np.concatenate([b - b64, d - d64]) ** 2)))
    rows.append(dict(scheme=name, style=style, track=k, max_vs_fp32=float(e32), max_vs_fp64=float(e64), rms_vs_fp64=rms,
                     flips_beat=flips(b, b32), flips_down=flips(d, d32), flips_beat_vs64=flips(b, b64), flips_down_vs64=flips(d, d64),
                     n_beats=len(frames_of(b32)), n_down=len(frames_of(d32))))
    return rows[-1]


def cmd_gpu(args):
    from beat_this_amd.inference import Audio2Frames
    from beat_this_amd.model import BeatThis

    os.makedirs(OUT, exist_ok=True)
    dev = torch.device("cuda:0")
    hp = W.resolve_hparams(MODEL)
    rows = []
    for style in args.styles.split(","):
        ks = available(style, args.tracks)
        if not ks:
            continue
        sd = W.random_state_dict(hp, seed=WEIGHT_SEED, style=style)
        m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
        m.load_state_dict(sd)
        m = m.to(dev)
        sigs = [torch.from_numpy(track(k)).to(dev) for k in ks]
        for scheme in args.schemes.split(","):
            # exact = exact fp32 MFMAs; x3 = hi + lo operands with the three-term P.V of rounds 3 - 4; x3p16 = the same with the
            # probabilities as fp16 hi parts in P.V (round 5 default; x3p16m: in the main layers only); half = fp16 operands
            # x3p16f8ff / x3p16f8 = x3p16 with the cross terms of the feed-forward / of all main-layer GEMMs on fp8 (BT_OPT_X3_GEMM_FP8 = 1 / 2)
            mode, p16, f8 = {"exact": ("exact", 2, 0), "x3": (False, 0, 0), "x3p16m": (False, 1, 0), "x3p16": (False, 2, 0), "half": (True, 2, 0),
                             "x3p16f": (False, 3, 0),   # P16 in the frontend's attention only (round 6: where do the flips come from?)
                             "x3p16f8ff": (False, 2, 1), "x3p16f8": (False, 2, 2)}[scheme]
            a2f = Audio2Frames(checkpoint_path=None, device=dev, float16=mode)
            m.fp32_split_gemms = True
            a2f.model = m
            m.engine().set_options({"x3_attn_p16": p16, "x3_gemm_fp8": f8})
            fb0 = m.engine().last_fallbacks
            t0 = time.time()
            for i in range(0, len(ks), 6):
                outs = a2f.many(sigs[i: i + 6], SR)
                for k, (b, d) in zip(ks[i: i + 6], outs):
                    score(scheme, style, k, b.cpu().numpy(), d.cpu().numpy(), rows)
            torch.cuda.synchronize()
            mine = [r for r in rows if r["scheme"] == scheme and r["style"] == style]
            print(f"{style:8s} {scheme:10s} {len(mine)} tracks {time.time() - t0:.1f} s: max vs fp32 {max(r['max_vs_fp32'] for r in mine):.2e}, "
                  f"flips {sum(r['flips_beat'] for r in mine)} / {sum(r['flips_down'] for r in mine)} of "
                  f"{sum(r['n_beats'] for r in mine)} / {sum(r['n_down'] for r in mine)}, range fallbacks "
                  f"{m.engine().last_fallbacks - fb0}", flush=True)
        m.engine().set_options({"x3_attn_p16": 1, "x3_gemm_fp8": 0})
    json.dump(rows, open(os.path.join(OUT, f"gpu_{args.tag}.json"), "w"))


# ---- simulated schemes (operand rounding on the oracle's forward; tools/x3_narrow_study.py) ------------------------------
def cmd_sim(args):
    import prec_study as P
    import x3_narrow_study as N

    os.makedirs(OUT, exist_ok=True)
    dev = torch.device(args.device)
    if dev.type == "cuda":
        torch.backends.cuda.matmul.allow_tf32 = False
    torch.set_num_threads(args.threads)
    hp = W.resolve_hparams(MODEL)
    split = N.split

    def mm_site(a, b, site):
        """hi + lo products with one site class degraded: scheme 'p16' = probabilities hi only in P.V; 'vhi' = V hi only;
        'e4m3' / 'e2m3' = both cross terms' operands in that MX format, everywhere"""
        s = SCHEME["name"]
        ah, al = split(a)
        bh, bl = split(b)
        if s == "x3sim":
            return ah @ bh + (ah @ bl + al @ bh)
        # BASELINE config 5 as written, report-only: the main layers' QKV / out-projection / FF1 / FF2 with BOTH
        # operands as OCP MX e4m3 (blocks of 32 along k share an E8M0 scale: what v_mfma_scale_f32_32x32x64_f8f6f4 multiplies at
        # twice the fp16 rate), everything else -- frontend, frontend.linear, attention scores and P.V -- on fp16 operands like
        # the half path; fp32 accumulation, residual stream, norms and GELU everywhere.  "halfsim" = the half path itself in the
        # same simulation (all sites fp16 operands): the yardstick config 5 has to be read against.
        if s in ("mxfp8", "halfsim"):
            if s == "mxfp8" and site in ("m_qkv", "m_out", "m_ff1", "m_ff2"):
                return N.q_mx(a, "e4m3", -1) @ N.q_mx(b, "e4m3", -2)
            return ah @ bh
        # round 6 candidates: the ACTIVATION operand of one main-layer GEMM as its fp16 hi part only (two MFMAs instead of three, half the
        # operand bytes from its producer): ff2ahi = the FF hidden activation into FF2, ff1ahi / qkvahi / outahi likewise
        if s in ("ff2ahi", "ff1ahi", "qkvahi", "outahi", "ff2ahi_p16m"):
            site_of = {"ff2ahi": "m_ff2", "ff1ahi": "m_ff1", "qkvahi": "m_qkv", "outahi": "m_out", "ff2ahi_p16m": "m_ff2"}[s]
            if site == site_of:
                return ah @ bh + ah @ bl
            if s == "ff2ahi_p16m" and site == "m_pv":
                return ah @ bh + ah @ bl
            return ah @ bh + (ah @ bl + al @ bh)
        if s == "p16":
            return ah @ bh + ah @ bl if site.endswith("pv") else ah @ bh + (ah @ bl + al @ bh)
        if s == "p16m":      # main layers only
            return ah @ bh + ah @ bl if site == "m_pv" else ah @ bh + (ah @ bl + al @ bh)
        if s == "vhi":
            return ah @ bh + al @ bh if site.endswith("pv") else ah @ bh + (ah @ bl + al @ bh)
        if s == "p16vhi":
            return ah @ bh if site.endswith("pv") else ah @ bh + (ah @ bl + al @ bh)
        return ah @ bh + (N.q_mx(ah, s, -1) @ N.q_mx(bl, s, -2) + N.q_mx(al, s, -1) @ N.q_mx(bh, s, -2))

    def attention_sim(x, sd, pfx, heads, tag):
        """N.attention with the row sum taken over the probabilities the P.V product actually uses (hi part only for the
        p16 schemes: numerator and denominator see the same rounded values)"""
        b, n, dim = x.shape
        xn = O.rmsnorm(x, sd[pfx + "norm.gamma"])
        qkv = mm_site(xn, sd[pfx + "to_qkv.weight"].T, tag + "qkv")
        d = qkv.shape[-1] // (3 * heads)
        qkv = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        fr = sd[pfx + "rotary_embed.freqs"]
        q, k = O.rope(q, fr), O.rope(k, fr)
        s = mm_site(q, k.transpose(-1, -2), tag + "qk") * (d ** -0.5)
        p = torch.exp(s - s.amax(-1, keepdim=True))
        hi_only = SCHEME["name"] in ("p16", "p16vhi", "mxfp8", "halfsim") or (SCHEME["name"] in ("p16m", "ff2ahi_p16m") and tag == "m_")
        den = p.to(torch.float16).float().sum(-1, keepdim=True) if hi_only else p.sum(-1, keepdim=True)
        out = mm_site(p, v, tag + "pv") / den
        gates = mm_site(xn, sd[pfx + "to_gates.weight"].T, tag + "qkv") + sd[pfx + "to_gates.bias"]
        out = out * torch.sigmoid(gates).permute(0, 2, 1)[..., None]
        out = out.permute(0, 2, 1, 3).reshape(b, n, heads * d)
        return mm_site(out, sd[pfx + "to_out.0.weight"].T, tag + "out")

    SCHEME = {"name": "x3sim"}
    N.mm = mm_site
    P.mm = mm_site
    P.attention = attention_sim
    N.attention = attention_sim
    _patch_rope_device()
    rows = []
    for style in args.styles.split(","):
        ks = available(style, args.tracks)
        sd = {k: v.to(dev) for k, v in W.random_state_dict(hp, seed=WEIGHT_SEED, style=style).items()}
        for scheme in args.schemes.split(","):
            SCHEME["name"] = scheme
            t0 = time.time()
            for k in ks:
                with torch.inference_mode():
                    spect = O.logmel(torch.from_numpy(track(k)))
                    chunks, starts = O.split_chunks(spect)
                    preds = []
                    for c in chunks:
                        b, d = N.forward(sd, c[None].to(dev))
                        preds.append((b[0].cpu(), d[0].cpu()))
                    b, d = O.aggregate(preds, starts, spect.shape[0])
                r = score(scheme, style, k, b.numpy(), d.numpy(), rows)
                print(f"{style} {scheme} track {k}: max {r['max_vs_fp32']:.2e} flips {r['flips_beat']} / {r['flips_down']} ({time.time() - t0:.0f} s)", flush=True)
            json.dump(rows, open(os.path.join(OUT, f"sim_{args.tag}.json"), "w"))


def _patch_rope_device():
    """the oracle's rope builds its angle table on the CPU; for --device cuda the table has to follow the operand"""
    def rope(t, freqs):
        n = t.shape[-2]
        ang = torch.arange(n, dtype=torch.float32, device=t.device)[:, None] * freqs.float()[None, :]
        cos = ang.cos().repeat_interleave(2, -1).to(t.dtype)
        sin = ang.sin().repeat_interleave(2, -1).to(t.dtype)
        te, to = t[..., 0::2], t[..., 1::2]
        rot = torch.stack((-to, te), dim=-1).flatten(-2)
        return t * cos + rot * sin
    O.rope = rope


def cmd_report(args):
    rows = []
    for p in sorted(glob.glob(os.path.join(OUT, "*.json"))):
        rows += json.load(open(p))
    lines = []
    summary = {}   # scheme -> style -> counts (the JSON twin of the table: bench.py's parity object quotes it with the file's sha)
    styles = sorted({r["style"] for r in rows}) or args.styles.split(",")
    hdr = (f"{'scheme':12s} {'style':8s} {'tracks':>6s} {'beats':>8s} {'flips b / d vs fp32 oracle':>28s} {'per 1000 beats':>15s} "
           f"{'vs fp64':>11s} {'max |dlogit| vs fp32':>21s} {'vs fp64':>9s} {'rms vs fp64':>12s}")
    lines.append(hdr)
    for style in styles:
        ks = available(style, 10 ** 6)
        if ks:   # the noise floor: the oracle's own fp32 against its fp64
            fb = fd = nb = nd = 0
            mx = 0.0
            sq = []
            for k in ks:
                b32, d32, b64, d64 = cached(style, k)
                fb += flips(b32, b64)
                fd += flips(d32, d64)
                nb += len(frames_of(b32))
                nd += len(frames_of(d32))
                mx = max(mx, np.abs(b32 - b64).max(), np.abs(d32 - d64).max())
                sq.append(np.mean(np.concatenate([b32 - b64, d32 - d64]) ** 2))
            summary.setdefault("oracle_fp32_vs_fp64", {})[style] = dict(tracks=len(ks), decisions=nb + nd, flips=fb + fd,
                                                                       flips_per_1000=round(1000.0 * (fb + fd) / max(1, nb + nd), 4))
            lines.append(f"{'oracle fp32':12s} {style:8s} {len(ks):6d} {nb:8d} {'(vs its own fp64) ' + str(fb) + ' / ' + str(fd):>28s} "
                         f"{1000.0 * (fb + fd) / max(1, nb + nd):15.3f} {'':>11s} {'':>21s} {mx:9.2e} {np.sqrt(np.mean(sq)):12.2e}")
        for scheme in sorted({r["scheme"] for r in rows if r["style"] == style}):
            mine = {r["track"]: r for r in rows if r["scheme"] == scheme and r["style"] == style}.values()   # (last run of a track wins)
            nb, nd = sum(r["n_beats"] for r in mine), sum(r["n_down"] for r in mine)
            fb, fd = sum(r["flips_beat"] for r in mine), sum(r["flips_down"] for r in mine)
            f64 = sum(r["flips_beat_vs64"] + r["flips_down_vs64"] for r in mine)
            summary.setdefault(scheme, {})[style] = dict(tracks=len(mine), decisions=nb + nd, flips=fb + fd,
                                                         flips_per_1000=round(1000.0 * (fb + fd) / max(1, nb + nd), 4),
                                                         max_abs_logit_vs_fp32=float(max(r["max_vs_fp32"] for r in mine)))
            lines.append(f"{scheme:12s} {style:8s} {len(mine):6d} {nb:8d} {str(fb) + ' / ' + str(fd):>28s} {1000.0 * (fb + fd) / max(1, nb + nd):15.3f} "
                         f"{f64:11d} {max(r['max_vs_fp32'] for r in mine):21.2e} {max(r['max_vs_fp64'] for r in mine):9.2e} "
                         f"{np.sqrt(np.mean([r['rms_vs_fp64'] ** 2 for r in mine])):12.2e}")
    # the margin distribution of the fp32 oracle's decisions: how large an error has to be to move a beat
    for style in styles:
        ks = available(style, 10 ** 6)
        if not ks:
            continue
        ms = np.sort(np.concatenate([np.concatenate([margin_stats(cached(style, k)[0]), margin_stats(cached(style, k)[1])]) for k in ks]))
        nb = sum(len(frames_of(cached(style, k)[0])) + len(frames_of(cached(style, k)[1])) for k in ks)
        lines.append(f"margins {style}: {len(ks)} tracks, {nb} beats + downbeats; decisions with margin below 1e-5 / 3e-5 / 1e-4 / 3e-4 / 1e-3 / 3e-3: "
                     + " / ".join(str(int((ms < t).sum())) for t in (1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3)))
    # the rule the default arithmetic is chosen by: over the soak, a candidate default may not flip more than
    # the exact fp32 MFMA path -- in total, and the per-style counts are printed beside it
    if "exact" in summary:
        for scheme in sorted(summary):
            if scheme in ("exact", "oracle_fp32_vs_fp64") or set(summary[scheme]) != set(summary["exact"]) or \
                    any(summary[scheme][st]["tracks"] != summary["exact"][st]["tracks"] for st in summary["exact"]):
                continue
            tot = sum(v["flips"] for v in summary[scheme].values())
            tot_e = sum(v["flips"] for v in summary["exact"].values())
            per = ", ".join(f"{st} {summary[scheme][st]['flips']} / {summary['exact'][st]['flips']}" for st in sorted(summary["exact"]))
            summary[scheme]["rule"] = dict(total=tot, exact_total=tot_e, admitted=bool(tot <= tot_e),
                                           per_style_not_above_exact=bool(all(summary[scheme][st]["flips"] <= summary["exact"][st]["flips"]
                                                                              for st in summary["exact"])))
            lines.append(f"rule {scheme:10s}: {tot} flips against the exact path's {tot_e} -> {'admitted' if tot <= tot_e else 'NOT admitted'} "
                         f"as a default (per style, scheme / exact: {per})")
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
        json.dump(summary, open(os.path.splitext(args.out)[0] + ".json", "w"), indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["oracle", "gpu", "sim", "report"])
    ap.add_argument("--tracks", type=int, default=48)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--styles", default="lively,outlier,init")
    ap.add_argument("--schemes", default=None)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--tag", default="run")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.schemes is None:
        args.schemes = {"gpu": "exact,x3,x3p16m,x3p16,x3p16f8ff,x3p16f8,half", "sim": "p16,vhi,e4m3,e2m3"}.get(args.cmd, "")
    {"oracle": cmd_oracle, "gpu": cmd_gpu, "sim": cmd_sim, "report": cmd_report}[args.cmd](args)


if __name__ == "__main__":
    main()

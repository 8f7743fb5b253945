"""Parity AT THE BENCHMARKED BATCH SIZES (BASELINE.json configs 2, 3, 4), not just single chunks: the HIP path against the
CPU oracle on the same seeded inputs, with beat / downbeat flip counts through the minimal post-processor, plus a
size-independent property over the WHOLE batch: every chunk of a big batch equals the same chunk forwarded alone
(rows are independent in every kernel, so a tile-scheduling or LDS-DMA ordering bug that corrupts a few rows only at
scale -- DESIGN.md section 5 records one -- shows up here even where the oracle is too slow to cover every chunk)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gpu_util import dev, report

pytestmark = pytest.mark.gpu

KEYS = ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")


def _setup(hp_name, B, seed=1, style="lively"):
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis

    hp = W.resolve_hparams(hp_name)
    sd = W.random_state_dict(hp, seed=seed, style=style)
    m = BeatThis(**{k: hp[k] for k in KEYS})
    m.load_state_dict(sd)
    x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=7000 + i) for i in range(B)]))
    return sd, m.to(dev()), x


def _flips(pp, gb, gd, ob, od):
    """symmetric difference of the beat / downbeat frame sets (half-frame resolution: merged plateaus sit on .5)"""
    b1, d1 = pp(gb, gd)
    b2, d2 = pp(ob, od)
    key = lambda a: set(np.round(np.asarray(a) * 100).astype(np.int64))  # noqa: E731
    return len(key(b1) ^ key(b2)), len(key(d1) ^ key(d2)), len(b2), len(d2)


def _oracle(sd, x, idx):
    from oracle import beat_this_oracle as O

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {}
    with torch.inference_mode():
        for i in idx:
            b, d = O.model_forward(sd, x[i: i + 1])
            out[i] = (b[0], d[0])
    return out


X3_TOL = 1.5e-4   # the default precision (BT_PREC_F32X3, P16 attention in the main layers only: BT_OPT_X3_ATTN_P16 = 1)
X3_TOL_P16_ALL = 3e-4   # the opt-in level 2 (round 5's default): admission bound of profiles/r05_flip_frontier.txt


def _check(name, hp_name, B, half, x3, oracle_idx, tol, flips_allowed, p16=1, fp8=0):
    from beat_this_amd.postprocessor import Postprocessor

    sd, m, x = _setup(hp_name, B)
    xd = x.to(dev())
    m.fp32_split_gemms = x3     # BT_PREC_F32X3 (hi + lo operands) instead of the exact fp32 MFMA path
    m.engine().set_options({"x3_attn_p16": p16, "x3_gemm_fp8": fp8})
    with torch.inference_mode(), torch.autocast("cuda", enabled=half):
        r = m(xd)
        # every chunk of the batch alone (16 at a time to keep it quick): must reproduce the batched result
        worst_alone = 0.0
        for i in range(B):
            ri = m(xd[i: i + 1])
            worst_alone = max(worst_alone, float((ri["beat"][0] - r["beat"][i]).abs().max()),
                              float((ri["downbeat"][0] - r["downbeat"][i]).abs().max()))
    assert torch.isfinite(r["beat"]).all() and torch.isfinite(r["downbeat"]).all()
    if x3:
        assert m.engine().last_fallbacks == 0, "the hi + lo path fell back to exact fp32 (range flag) on ordinary inputs"
    ref = _oracle(sd, x, oracle_idx)
    pp = Postprocessor("minimal")
    err, fb, fd, nb, nd = 0.0, 0, 0, 0, 0
    for i, (ob, od) in ref.items():
        gb, gd = r["beat"][i].float().cpu(), r["downbeat"][i].float().cpu()
        err = max(err, float((gb - ob).abs().max()), float((gd - od).abs().max()))
        f = _flips(pp, gb, gd, ob, od)
        fb, fd, nb, nd = fb + f[0], fd + f[1], nb + f[2], nd + f[3]
    report("scale_parity", config=name, B=B, chunks_vs_oracle=len(ref), max_abs_logit=err, flips_beat=fb,
           flips_downbeat=fd, n_beats=nb, n_downbeats=nd, batch_vs_alone=worst_alone)
    assert worst_alone <= 1e-5, f"batched result differs from single-chunk result by {worst_alone}"
    assert err < tol, err
    if flips_allowed is not None:
        assert fb <= flips_allowed[0] and fd <= flips_allowed[1], (fb, fd)


def _ref_report():
    return json.load(open(os.path.join(GOLDEN, "reference_autocast_report.json")))


def test_cfg2_final0_16_chunks_fp32_vs_oracle():
    # the parity-gated path: 1e-3 on the logits and IDENTICAL beat / downbeat frames on all 16 chunks
    _check("cfg2_f32", "final0", 16, half=False, x3=False, oracle_idx=range(16), tol=1e-3, flips_allowed=(0, 0))


def test_cfg2_final0_16_chunks_half_vs_oracle():
    # half-precision operands: bounded by the reference's OWN float16-autocast error on the final0 golden case
    # (tests/golden/reference_autocast_report.json, generated from the unmodified reference)
    ref = _ref_report()["final0_lively_T1500_f16"]
    _check("cfg2_half", "final0", 16, half=True, x3=False, oracle_idx=range(16),
           tol=max(ref["max_abs_beat"], ref["max_abs_downbeat"]), flips_allowed=None)


def test_cfg3_small0_128_chunks_fp32_vs_oracle():
    _check("cfg3_small0_f32", "small0", 128, half=False, x3=False, oracle_idx=[0, 1, 37, 63, 64, 100, 126, 127], tol=1e-3,
           flips_allowed=(0, 0))


def test_cfg2_final0_16_chunks_f32x3_vs_oracle():
    # BT_PREC_F32X3 (hi + lo operands on the LDS-DMA kernels): the SAME gate as the exact path -- 1e-3, tested at 3e-4 (1e-4
    # with the three-term P.V), and IDENTICAL beat / downbeat frames -- on all 16 chunks, plus batch-vs-alone on every chunk
    _check("cfg2_f32x3", "final0", 16, half=False, x3=True, oracle_idx=range(16), tol=X3_TOL, flips_allowed=(0, 0))


def test_cfg2_final0_16_chunks_f32x3_p16_everywhere_vs_oracle():
    # the opt-in BT_OPT_X3_ATTN_P16 = 2 (round 5's default): inside the gate, held to 3e-4
    _check("cfg2_f32x3_p16all", "final0", 16, half=False, x3=True, oracle_idx=range(0, 16, 3), tol=X3_TOL_P16_ALL, flips_allowed=(0, 0), p16=2)


def test_cfg2_final0_16_chunks_f32x3_three_term_vs_oracle():
    _check("cfg2_f32x3_p16off", "final0", 16, half=False, x3=True, oracle_idx=range(0, 16, 3), tol=1e-4, flips_allowed=(0, 0), p16=0)


@pytest.mark.parametrize("level", [1, 2])
def test_cfg5_final0_fp8_cross_terms_vs_oracle(level):
    """BASELINE config 5 as offered (BT_OPT_X3_GEMM_FP8): the cross terms of the main layers' GEMMs on block-scaled fp8 MFMAs.  Inside
    the 1e-3 gate on the logits; beats reported (an opt-in speed setting: its flip rate over the soak is in DESIGN.md section 3);
    every chunk the same bits alone and in the batch, like the default."""
    _check(f"cfg5_fp8_level{level}", "final0", 16, half=False, x3=True, oracle_idx=range(0, 16, 3), tol=1e-3, flips_allowed=None, fp8=level)


def test_bench_slice_final0_33_chunks_f32x3_vs_oracle():
    # the benchmark's forward slice (66 chunks as 2 x 33 on two streams): 33 chunks, partial GEMM tiles and CU rounds
    _check("slice33_f32x3", "final0", 33, half=False, x3=True, oracle_idx=[0, 16, 32], tol=X3_TOL, flips_allowed=(0, 0))


def test_cfg3_small0_128_chunks_f32x3_vs_oracle():
    _check("cfg3_small0_f32x3", "small0", 128, half=False, x3=True, oracle_idx=[0, 63, 127], tol=X3_TOL, flips_allowed=(0, 0))


def test_cfg4_share_final0_64_chunks_f32x3_vs_oracle():
    # BASELINE config 4's per-GPU share (and the headline's launch shape: 64 / 66 chunks): the gate-carrying path at 1e-4 and
    # 0 flips against the oracle on four chunks spread over the batch, batch-vs-alone on all 64
    _check("cfg4_share_f32x3", "final0", 64, half=False, x3=True, oracle_idx=[0, 21, 42, 63], tol=X3_TOL, flips_allowed=(0, 0))


def test_cfg4_share_final0_64_chunks_half_batch_consistency():
    _check("cfg4_share_half", "final0", 64, half=True, x3=False, oracle_idx=[0, 63], tol=0.05, flips_allowed=None)


@pytest.mark.parametrize("case", ["small0_lively_T1500", "small0_lively_T1012", "final0_lively_T1500"])
def test_half_path_against_reference_autocast_goldens(case):
    """Our half-precision path vs the reference's own reduced-precision forward on the same input: error against the
    fp32 golden must not exceed the reference's float16-autocast error, and flips must not exceed its flips."""
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis
    from beat_this_amd.postprocessor import Postprocessor
    from oracle.cases import MODEL_CASES

    name, hpn, wseed, style, T, iseed = next(c for c in MODEL_CASES if c[0] == case)
    g = np.load(os.path.join(GOLDEN, "model_logits.npz"))
    rep = _ref_report()
    hp = W.resolve_hparams(hpn)
    m = BeatThis(**{k: hp[k] for k in KEYS})
    m.load_state_dict(W.random_state_dict(hp, seed=wseed, style=style))
    m = m.to(dev())
    x = torch.from_numpy(W.synthetic_spect(T, seed=iseed))[None].to(dev())
    with torch.inference_mode(), torch.autocast("cuda", enabled=True):
        r = m(x)
    gb, gd = r["beat"][0].float().cpu(), r["downbeat"][0].float().cpu()
    ob, od = torch.from_numpy(g[name + "_beat"]), torch.from_numpy(g[name + "_downbeat"])
    err = max(float((gb - ob).abs().max()), float((gd - od).abs().max()))
    fb, fd, nb, nd = _flips(Postprocessor("minimal"), gb, gd, ob, od)
    r16, rbf = rep[name + "_f16"], rep[name + "_bf16"]
    report("half_vs_reference_autocast", case=name, max_abs_logit=err, flips_beat=fb, flips_downbeat=fd, n_beats=nb,
           ref_f16_max_abs=max(r16["max_abs_beat"], r16["max_abs_downbeat"]), ref_f16_flips=[r16["flips_beat"], r16["flips_downbeat"]],
           ref_bf16_max_abs=max(rbf["max_abs_beat"], rbf["max_abs_downbeat"]), ref_bf16_flips=[rbf["flips_beat"], rbf["flips_downbeat"]])
    assert err <= max(r16["max_abs_beat"], r16["max_abs_downbeat"])
    assert fb + fd <= r16["flips_beat"] + r16["flips_downbeat"] + 1


@pytest.mark.parametrize("half", [False, True, "f32x3"])
@pytest.mark.parametrize("hp_name,B", [("final0", 16), ("small0", 48), ("final0", 33)])
def test_batched_forward_is_repeatable_bit_for_bit(hp_name, B, half):
    """The same batch forwarded five times must give five identical results.  Round 2 found (this way) a write-after-read
    race between a fast wave's LDS-DMA refill and a slow wave's outstanding fragment reads in the weight rings of
    csrc/fused2.hip (raw s_barrier without lgkmcnt(0)): ~1 % of the workgroups of a 16-chunk launch computed garbage, a
    different set on every run, invisible to single-chunk goldens."""
    sd, m, x = _setup(hp_name, B)
    xd = x.to(dev())
    m.fp32_split_gemms = half == "f32x3"
    with torch.inference_mode(), torch.autocast("cuda", enabled=half is True):
        outs = [m(xd) for _ in range(5)]
    torch.cuda.synchronize()
    bad = sum(int(not (torch.equal(o["beat"], outs[0]["beat"]) and torch.equal(o["downbeat"], outs[0]["downbeat"])))
              for o in outs[1:])
    report("repeatable", model=hp_name, B=B, half=half, deviating_repeats=bad)
    assert bad == 0


# (beat flips, downbeat flips, of 2163 / 1877) of the half path on the benchmark's first track; bench.py's `half_path.parity`
# reports the same pair.  Update DELIBERATELY, with the reason in the commit, when a kernel change moves it.
HALF_FLIPS_TRACK0 = (10, 10)


def test_half_path_flips_on_the_benchmark_track_are_pinned():
    """The half path (float16=True) is not under the 1e-3 / identical-beats gate, but what it does to the beats is part of the
    record: on the benchmark's own first track (final0, "lively" weights, 300 s at 44.1 kHz through resampler and log-mel)
    its near-threshold peaks flip a FIXED number of beats against the CPU oracle.  Round 3 moved that number (6 / 9 ->
    10 / 10) with a 1-ulp change of the log-mel kernel and only the bench noticed; this test pins it, and pins the fp32-class
    default path at zero on the same track."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from beat_this_amd.model import BeatThis
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams("final0")
    sd = W.random_state_dict(hp, seed=1, style="lively")
    sig = W.synthetic_audio(300.0, seed=0, sr=44100)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.inference_mode():
        ob, od = O.audio2frames(sd, sig, 44100)
    obeats, odown = O.postp_minimal(ob, od)
    key = lambda a: set(np.round(np.asarray(a) * 100).astype(np.int64))  # noqa: E731
    got = {}
    for mode in (True, False):
        a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=mode, dbn=False)
        m = BeatThis(**{k: hp[k] for k in KEYS})
        m.load_state_dict(sd)
        a2b.model = m.to(dev())
        h = a2b.many_async([torch.from_numpy(sig).to(dev())], 44100)
        beats, downs = h.result()[0]
        err = max(float((h.logits[0].cpu() - ob).abs().max()), float((h.logits[1].cpu() - od).abs().max()))
        got[mode] = (len(key(beats) ^ key(obeats)), len(key(downs) ^ key(odown)), err)
    report("pinned_flips", half=got[True][:2], half_err=got[True][2], default=got[False][:2], default_err=got[False][2],
           n_beats=len(obeats), n_downbeats=len(odown))
    assert got[False][:2] == (0, 0) and got[False][2] < X3_TOL
    assert got[True][:2] == HALF_FLIPS_TRACK0, f"half-path flips moved: {got[True][:2]} (pinned {HALF_FLIPS_TRACK0})"


@pytest.mark.parametrize("hp_name,B", [("small0", 4), ("final0", 3)])
def test_outlier_weights_in_three_precisions(hp_name, B):
    """Trained-like stress weights (weights.random_state_dict style="outlier": residual-stream outlier channels of ~10^3,
    heavy-tailed matrices, sharp attention, frontend activations of ~10^2) -- what a real checkpoint is expected to look like
    to the fp16 range and to the hi + lo representation of BT_PREC_F32X3: the exact path and the default (f32x3) path stay
    inside the 1e-3 gate with identical beats, the default path WITHOUT falling back to the exact one, and the half path is
    reported."""
    from beat_this_amd.postprocessor import Postprocessor

    sd, m, x = _setup(hp_name, B, seed=1, style="outlier")
    xd = x.to(dev())
    ref = _oracle(sd, x, range(B))
    pp = Postprocessor("minimal")
    res = {}
    for mode in ("f32", "f32x3", "f32x3_fp8", "half"):
        m.fp32_split_gemms = mode.startswith("f32x3")
        m.engine().set_options({"x3_gemm_fp8": 2 if mode == "f32x3_fp8" else 0})   # (BASELINE config 5's opt-in arithmetic)
        before = m.engine().last_fallbacks
        with torch.inference_mode(), torch.autocast("cuda", enabled=mode == "half"):
            r = m(xd)
        err, fb, fd, nb = 0.0, 0, 0, 0
        for i, (ob, od) in ref.items():
            gb, gd = r["beat"][i].float().cpu(), r["downbeat"][i].float().cpu()
            err = max(err, float((gb - ob).abs().max()), float((gd - od).abs().max()))
            f = _flips(pp, gb, gd, ob, od)
            fb, fd, nb = fb + f[0], fd + f[1], nb + f[2]
        res[mode] = (err, fb, fd, m.engine().last_fallbacks - before)
        report("outlier_weights", model=hp_name, mode=mode, max_abs_logit=err, flips_beat=fb, flips_downbeat=fd, n_beats=nb,
               range_fallbacks=res[mode][3], logit_spread=float(torch.stack([v[0] for v in ref.values()]).std()))
    assert res["f32"][0] < 1e-3 and res["f32"][1:3] == (0, 0)
    assert res["f32x3"][3] == 0, "BT_PREC_F32X3 fell back to the exact path on outlier channels of ~10^3"
    assert res["f32x3"][0] < 1e-3 and res["f32x3"][1:3] == (0, 0)
    # the fp8 cross terms too: their activations' byte sections reach 3584 (csrc/common.h), the outlier channels stay below
    assert res["f32x3_fp8"][3] == 0, "BT_OPT_X3_GEMM_FP8 fell back to the exact path on outlier channels of ~10^3"
    assert res["f32x3_fp8"][0] < 1e-3
    m.engine().set_options({"x3_gemm_fp8": 0})
    assert np.isfinite(res["half"][0])

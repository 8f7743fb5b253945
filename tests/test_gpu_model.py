"""End-to-end parity on the MI355X: HIP path (through the reference-shaped Python API) against
the committed golden outputs of the reference and against the CPU oracle on the same inputs."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gpu_util import dev, report

pytestmark = pytest.mark.gpu

LOGIT_TOL_F32 = 1e-3   # BASELINE.json north_star: framewise logits within 1e-3 (fp32 path)
X3_TOL = 1.5e-4        # what the tests hold the default precision to (BT_PREC_F32X3, P16 attention in the main layers only =
                       # BT_OPT_X3_ATTN_P16 1, the default since round 6): measured <= 1.2e-4 on every model of the suite
X3_TOL_P16_ALL = 3e-4  # ... and the opt-in level 2 (P16 in the frontend as well, round 5's default)


def _model(hp, seed, style):
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis

    hparams = W.resolve_hparams(hp)
    m = BeatThis(**{k: hparams[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim",
                                            "stem_dim")})
    m.load_state_dict(W.random_state_dict(hp, seed=seed, style=style))
    return m.to(dev())


def _cases():
    from oracle.cases import MODEL_CASES
    return MODEL_CASES


@pytest.mark.parametrize("case", range(6))
def test_forward_fp32_matches_reference_golden(case):
    from beat_this_amd import weights as W

    name, hpn, wseed, style, T, iseed = _cases()[case]
    g = np.load(os.path.join(GOLDEN, "model_logits.npz"))
    m = _model(hpn, wseed, style)
    x = torch.from_numpy(W.synthetic_spect(T, seed=iseed))[None].to(dev())
    with torch.inference_mode():
        r = m(x)
    eb = float(np.abs(r["beat"][0].cpu().numpy() - g[name + "_beat"]).max())
    ed = float(np.abs(r["downbeat"][0].cpu().numpy() - g[name + "_downbeat"]).max())
    report("forward_fp32", case=name, err_beat=eb, err_downbeat=ed)
    assert eb < LOGIT_TOL_F32 and ed < LOGIT_TOL_F32


@pytest.mark.parametrize("case", [0, 2, 3, 4])
def test_forward_half_close_and_reported(case):
    """Half-precision path (float16=True, fp16 MFMA operands): not under the 1e-3 gate; bounded by the largest error of
    the reference's OWN float16 autocast forward against its fp32 forward on the golden cases
    (tests/golden/reference_autocast_report.json: 1.2e-2 ... 1.5e-2) -- the same yardstick
    test_half_path_against_reference_autocast_goldens applies case by case."""
    from beat_this_amd import weights as W

    name, hpn, wseed, style, T, iseed = _cases()[case]
    g = np.load(os.path.join(GOLDEN, "model_logits.npz"))
    m = _model(hpn, wseed, style)
    x = torch.from_numpy(W.synthetic_spect(T, seed=iseed))[None].to(dev())
    with torch.inference_mode(), torch.autocast("cuda", enabled=True):
        r = m(x)
    ref = g[name + "_beat"]
    eb = float(np.abs(r["beat"][0].cpu().numpy() - ref).max())
    rms = float(np.sqrt(np.mean((r["beat"][0].cpu().numpy() - ref) ** 2)))
    rep = json.load(open(os.path.join(GOLDEN, "reference_autocast_report.json")))
    bound = max(max(v["max_abs_beat"], v["max_abs_downbeat"]) for k, v in rep.items() if k.endswith("_f16"))
    report("forward_half", case=name, err_beat=eb, rms=rms, spread=float(ref.std()), bound=bound)
    from beat_this_amd import _lib
    if _lib.lib().bt_half_is_bf16():   # (a -DBT_HALF_BF16 development build: the reference's bfloat16 yardstick)
        bound = max(max(v["max_abs_beat"], v["max_abs_downbeat"]) for k, v in rep.items() if k.endswith("_bf16"))
    assert eb <= bound


def test_forward_batched_and_deterministic():
    from beat_this_amd import weights as W

    g = np.load(os.path.join(GOLDEN, "model_logits.npz"))
    m = _model("small0", 1, "lively")
    xb = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=20 + i) for i in range(3)])).to(dev())
    with torch.inference_mode():
        r1 = m(xb)
        r2 = m(xb)
    assert torch.equal(r1["beat"], r2["beat"]) and torch.equal(r1["downbeat"], r2["downbeat"])
    eb = float(np.abs(r1["beat"].cpu().numpy() - g["small0_lively_B3_beat"]).max())
    report("forward_batched", err_beat=eb)
    assert eb < LOGIT_TOL_F32


def test_intermediate_taps_against_oracle():
    """Stage-by-stage check of the fp32 path against the CPU oracle (small model, short input):
    localises an error to a kernel instead of only seeing it in the logits."""
    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    sd = W.random_state_dict("small0", seed=3, style="lively")
    x = torch.from_numpy(W.synthetic_spect(300, seed=8))[None]
    with torch.inference_mode():
        b64, d64 = O.model_forward(sd, x, torch.float64)
    m = _model("small0", 3, "lively")
    with torch.inference_mode():
        r = m(x.to(dev()))
    eb = float((r["beat"][0].cpu().double() - b64[0]).abs().max())
    report("forward_vs_oracle64", err_beat=eb)
    assert eb < LOGIT_TOL_F32


def test_spect2frames_piece_and_api_types():
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Spect2Frames

    g = np.load(os.path.join(GOLDEN, "model_logits.npz"))
    s2f = Spect2Frames(checkpoint_path=None, device="cuda:0", float16=False)
    s2f.model = _model("small0", 1, "lively")
    piece = torch.from_numpy(W.synthetic_spect(3100, seed=30)).to(dev())
    beat, down = s2f(piece)
    assert beat.dtype == torch.float32 and beat.shape == (3100,) and beat.device.type == "cuda"
    eb = float(np.abs(beat.cpu().numpy() - g["small0_lively_piece3100_beat"]).max())
    ed = float(np.abs(down.cpu().numpy() - g["small0_lively_piece3100_downbeat"]).max())
    report("spect2frames_piece", err_beat=eb, err_downbeat=ed)
    assert eb < LOGIT_TOL_F32 and ed < LOGIT_TOL_F32
    # several pieces at once (extension) == one at a time
    many = s2f.spect2frames_many([piece, piece[:1000], piece[:1600]])
    assert torch.equal(many[0][0], beat)
    b1000, _ = s2f(piece[:1000].contiguous())
    assert torch.equal(many[1][0], b1000)


def test_logmel_against_golden_and_oracle():
    from beat_this_amd import weights as W
    from beat_this_amd.preprocessing import LogMelSpect
    from oracle import beat_this_oracle as O

    g = np.load(os.path.join(GOLDEN, "logmel.npz"))
    lm = LogMelSpect(device="cuda:0")
    s2 = lm(torch.from_numpy(W.synthetic_audio(2.0, seed=11)).to(dev())).cpu().numpy()
    assert s2.shape == (101, 128)
    e2 = float(np.abs(s2 - g["s2"]).max())
    a30 = W.synthetic_audio(30.0, seed=12)
    s30 = lm(torch.from_numpy(a30).to(dev())).cpu().numpy()
    assert s30.shape == (1501, 128)
    e30 = float(np.abs(s30[g["s30_rows"]] - g["s30_sel"]).max())
    o64 = O.logmel(torch.from_numpy(a30), torch.float64).numpy()
    e64 = float(np.abs(s30 - o64).max())
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * np.arange(22050 * 3) / 22050.0)).astype(np.float32)
    st = lm(torch.from_numpy(tone).to(dev())).cpu().numpy()
    et = float(np.abs(st[g["tone_rows"]] - g["tone_sel"]).max())
    report("logmel", err_2s=e2, err_30s=e30, err_vs_f64=e64, err_tone=et)
    assert e2 < 1e-4 and e30 < 1e-4 and e64 < 1e-4   # SURVEY.md 7 step 2 tolerance
    assert et < 2e-3  # empty bands of a pure tone: log1p(1000 x) amplifies fp32 FFT noise
    with pytest.raises(ValueError):
        lm(torch.zeros(100, device=dev()))


def test_postprocessor_bit_exact():
    from beat_this_amd.postprocessor import Postprocessor
    from oracle.cases import POSTP_CASES

    post = json.load(open(os.path.join(GOLDEN, "postp_minimal.json")))
    pp = Postprocessor("minimal", fps=50)
    for name, (bs, ds) in POSTP_CASES.items():
        b = torch.full((100,), -5.0)
        d = torch.full((100,), -5.0)
        for f, v in bs:
            b[f] = v
        for f, v in ds:
            d[f] = v
        bt, dt = pp(b.to(dev()), d.to(dev()))
        assert isinstance(bt, np.ndarray) and bt.dtype == np.float64
        assert bt.tolist() == post[name]["beats"], name
        assert dt.tolist() == post[name]["downbeats"], name
    rng = np.random.default_rng(post["random3000"]["seed"])
    rb = torch.from_numpy(rng.normal(-1.0, 1.5, 3000).astype(np.float32)).to(dev())
    rd = torch.from_numpy(rng.normal(-2.0, 1.5, 3000).astype(np.float32)).to(dev())
    bt, dt = pp(rb, rd)
    assert bt.tolist() == post["random3000"]["beats"] and dt.tolist() == post["random3000"]["downbeats"]
    # batched call returns tuples, like the reference (postprocessor.py:58-83)
    bb, dd = pp(torch.stack([rb, rb]), torch.stack([rd, rd]))
    assert isinstance(bb, tuple) and len(bb) == 2 and bb[1].tolist() == post["random3000"]["beats"]


def test_postprocessor_padding_mask_matches_reference():
    """The ``padding_mask`` argument (postprocessor.py:85-136) on device logits: bit-exact against the reference's outputs."""
    from test_cabi import _check_padding_mask_cases

    _check_padding_mask_cases(dev())


def test_audio2beats_end_to_end_golden():
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats

    g = np.load(os.path.join(GOLDEN, "e2e_small0.npz"))
    a2b = Audio2Beats(checkpoint_path=None, device="cuda:0", float16=False, dbn=False)
    a2b.model = _model("small0", 1, "lively")
    sig = W.synthetic_audio(40.0, seed=13)
    beats, downbeats = a2b(sig, 22050)
    bl, dl = a2b.spect2frames(a2b.signal2spect(sig, 22050))
    eb = float(np.abs(bl.cpu().numpy() - g["beat_logits"]).max())
    flips_b = len(set(np.round(beats * 50, 1)) ^ set(np.round(g["beats"] * 50, 1)))
    flips_d = len(set(np.round(downbeats * 50, 1)) ^ set(np.round(g["downbeats"] * 50, 1)))
    report("audio2beats", err_logits=eb, n_beats=len(beats), flips_beat=flips_b, flips_downbeat=flips_d)
    assert eb < LOGIT_TOL_F32
    assert np.array_equal(beats, g["beats"]) and np.array_equal(downbeats, g["downbeats"])
    with pytest.raises(ValueError):
        a2b.signal2spect(np.zeros((4, 4, 4)), 22050)


FINAL0_MODES = [("fp32", "exact", LOGIT_TOL_F32), ("f32x3", False, X3_TOL), ("half", True, 2.5e-2)]


@pytest.mark.parametrize("mode,float16,tol", FINAL0_MODES)
def test_final0_piece_matches_reference_golden(mode, float16, tol):
    """final0, a 3100-frame piece through Spect2Frames (three chunks, the last one moved back; keep_first aggregation) against
    the UNMODIFIED reference's split_predict_aggregate output (oracle/make_golden_final0.py), in the three precisions."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Spect2Frames
    from oracle.cases import FINAL0_CASES as C

    g = np.load(os.path.join(GOLDEN, "final0_piece_e2e.npz"))
    s2f = Spect2Frames(checkpoint_path=None, device="cuda:0", float16=float16)
    s2f.model = _model("final0", C["piece"]["weight_seed"], C["piece"]["style"])
    piece = torch.from_numpy(W.synthetic_spect(C["piece"]["frames"], seed=C["piece"]["input_seed"])).to(dev())
    beat, down = s2f(piece)
    assert beat.dtype == torch.float32 and beat.shape == (3100,)
    eb = float(np.abs(beat.cpu().numpy() - g["piece_beat"]).max())
    ed = float(np.abs(down.cpu().numpy() - g["piece_downbeat"]).max())
    report("final0_piece_golden", mode=mode, err_beat=eb, err_downbeat=ed)
    assert eb < tol and ed < tol
    if mode == "f32x3":
        assert s2f.model.engine().last_fallbacks == 0


@pytest.mark.parametrize("mode,float16,tol", FINAL0_MODES)
def test_final0_audio2beats_matches_reference_golden(mode, float16, tol):
    """final0, Audio2Beats on 40 s of 22.05 kHz audio against the unmodified reference's logits and beat / downbeat TIMES:
    identical beats in the two fp32-class precisions (north_star's bar), reported for the fp16 path."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from oracle.cases import FINAL0_CASES as C

    g = np.load(os.path.join(GOLDEN, "final0_piece_e2e.npz"))
    a2b = Audio2Beats(checkpoint_path=None, device="cuda:0", float16=float16, dbn=False)
    a2b.model = _model("final0", C["e2e"]["weight_seed"], C["e2e"]["style"])
    sig = W.synthetic_audio(C["e2e"]["seconds"], seed=C["e2e"]["audio_seed"])
    beats, downbeats = a2b(sig, 22050)
    bl, dl = a2b.spect2frames(a2b.signal2spect(sig, 22050))
    eb = max(float(np.abs(bl.cpu().numpy() - g["e2e_beat_logits"]).max()), float(np.abs(dl.cpu().numpy() - g["e2e_downbeat_logits"]).max()))
    flips_b = len(set(np.round(beats * 50, 1)) ^ set(np.round(g["e2e_beats"] * 50, 1)))
    flips_d = len(set(np.round(downbeats * 50, 1)) ^ set(np.round(g["e2e_downbeats"] * 50, 1)))
    report("final0_audio2beats_golden", mode=mode, err_logits=eb, n_beats=len(g["e2e_beats"]), flips_beat=flips_b, flips_downbeat=flips_d)
    assert eb < tol
    if mode != "half":
        assert np.array_equal(beats, g["e2e_beats"]) and np.array_equal(downbeats, g["e2e_downbeats"])


@pytest.mark.parametrize("mode,float16,tol", [("fp32", "exact", LOGIT_TOL_F32), ("f32x3", False, 3e-4), ("half", True, 5e-2)])
def test_final0_outlier_chunk_matches_reference_golden(mode, float16, tol):
    """final0 on the trained-like "outlier" weight style (residual outlier channels of ~1e3, heavy-tailed matrices, sharp
    attention): one chunk against the unmodified reference's forward.  (The default precision is held to 3e-4 here -- the
    style's logit error in the soak is 9e-5 at this arithmetic; the range guard must not fire.)"""
    from beat_this_amd import weights as W
    from oracle.cases import FINAL0_CASES as C

    g = np.load(os.path.join(GOLDEN, "final0_piece_e2e.npz"))
    m = _model("final0", C["outlier"]["weight_seed"], "outlier")
    m.fp32_split_gemms = mode == "f32x3"
    x = torch.from_numpy(W.synthetic_spect(C["outlier"]["frames"], seed=C["outlier"]["input_seed"]))[None].to(dev())
    with torch.inference_mode(), torch.autocast("cuda", enabled=float16 is True):
        r = m(x)
    eb = float(np.abs(r["beat"][0].float().cpu().numpy() - g["outlier_beat"]).max())
    ed = float(np.abs(r["downbeat"][0].float().cpu().numpy() - g["outlier_downbeat"]).max())
    report("final0_outlier_golden", mode=mode, err_beat=eb, err_downbeat=ed)
    assert eb < tol and ed < tol
    if mode == "f32x3":
        assert m.engine().last_fallbacks == 0


def test_split_and_aggregate_kernels_match_oracle():
    from beat_this_amd.inference import split_predict_aggregate
    from oracle import beat_this_oracle as O

    for n in (200, 1488, 1489, 1500, 1501, 2977, 4465):
        spect = torch.randn(n, 128, generator=torch.Generator().manual_seed(n))
        # a "model" that returns identifiable values: mean over mel bins, and its negative
        fake = lambda c: {"beat": c.mean(-1), "downbeat": -c.mean(-1) + 0.5}  # noqa: E731
        r = split_predict_aggregate(spect.to(dev()), 1500, 6, "keep_first", fake)
        chunks, starts = O.split_chunks(spect)
        preds = [(c.mean(-1), -c.mean(-1) + 0.5) for c in chunks]
        ob, od = O.aggregate(preds, starts, n)
        assert torch.allclose(r["beat"].cpu(), ob, atol=1e-6), n
        assert torch.allclose(r["downbeat"].cpu(), od, atol=1e-6), n
        rl = split_predict_aggregate(spect.to(dev()), 1500, 6, "keep_last", fake)
        assert rl["beat"].shape == (n,)


def test_cpu_inputs_fail_loudly():
    from beat_this_amd.inference import Spect2Frames
    from beat_this_amd.model import BeatThis

    with pytest.raises(RuntimeError):
        BeatThis()(torch.zeros(1, 100, 128))
    s2f = Spect2Frames(checkpoint_path=None, device="cuda:0")
    with pytest.raises(RuntimeError):
        s2f(torch.zeros(100, 128))


def test_audio2beats_44k1_input_resampled_on_gpu():
    """44.1 kHz input: mono mix (host) -> GPU polyphase resampler -> GPU log-mel -> model, against the oracle's own
    signal2spect (float64 resampling by its soxr stand-in, oracle/shims/soxr): same beats, logits < 1e-3."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams("small0")
    sd = W.random_state_dict(hp, seed=4, style="lively")
    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=False, dbn=False)
    from beat_this_amd.model import BeatThis
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(sd)
    a2b.model = m.to(dev())
    mono = W.synthetic_audio(12.0, seed=9, sr=44100)
    stereo = np.stack([mono, 0.5 * mono], 1).astype(np.float64)          # (N, 2): exercises the mono mix too
    beats, downbeats = a2b(stereo, 44100)
    bl, dl = a2b.spect2frames(a2b.signal2spect(stereo, 44100))
    with torch.inference_mode():
        ob, od = O.audio2frames(sd, stereo, 44100)
    err = float((bl.cpu() - ob).abs().max())
    obeats, odown = O.postp_minimal(ob, od)
    report("a2b_44k1", err=err, beats=len(beats), downbeats=len(downbeats))
    assert bl.shape == ob.shape and err < 1e-3
    assert np.array_equal(beats, obeats) and np.array_equal(downbeats, odown)


@pytest.mark.parametrize("prec_half", [False, True, "f32x3"])
@pytest.mark.parametrize("variant", ["no_sum_head", "no_partial", "three_layers_d256", "d64_ffmult2", "d192"])
def test_ablation_variants_against_oracle(variant, prec_half):
    """Head instead of SumHead (beat_tracker.py:333-346), frontend blocks without partial transformers
    (:143-153), other transformer_dim / n_layers -- against the oracle on seeded weights."""
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis
    from oracle import beat_this_oracle as O

    hp = dict(W.resolve_hparams("small0"))
    if variant == "no_sum_head":
        hp["sum_head"] = False
    elif variant == "no_partial":
        hp["partial_transformers"] = False
    elif variant == "d64_ffmult2":   # (not a multiple of 128: the main layers run on the register-staged GEMM / flash kernels)
        hp.update(transformer_dim=64, n_layers=2, ff_mult=2)
    elif variant == "d192":
        hp.update(transformer_dim=192, n_layers=2)
    else:
        hp.update(transformer_dim=256, n_layers=3)   # (half precision: the fused layer tail's C = 256 instantiation)
    sd = W.random_state_dict(hp, seed=21, style="lively")
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim",
                                       "sum_head", "partial_transformers")})
    m.load_state_dict(sd)
    m = m.to(dev())
    m.fp32_split_gemms = prec_half == "f32x3"
    x = torch.from_numpy(np.stack([W.synthetic_spect(700, seed=31), W.synthetic_spect(700, seed=32)]))
    with torch.inference_mode(), torch.autocast("cuda", enabled=prec_half is True):
        r = m(x.to(dev()))
    with torch.inference_mode():
        ob, od = O.model_forward(sd, x, sum_head=hp["sum_head"])
    eb = float((r["beat"].cpu() - ob).abs().max())
    ed = float((r["downbeat"].cpu() - od).abs().max())
    report("ablation", variant=variant, half=prec_half, err_beat=eb, err_downbeat=ed, spread=float(ob.std()))
    # (half: the reference's fp16-autocast scale.  f32x3 = the default precision, P16 attention in the main layers only: 1.5e-4;
    # the exact path: the gate.  The opt-in level 2 -- P16 in the frontend as well, round 5's default -- is held below the gate too:
    # these small / narrow variants amplify the frontend's share more than final0 / small0 do, 3e-4 .. 6e-4 measured, 7.5e-4 asserted)
    tol = 2.5e-2 if prec_half is True else X3_TOL if prec_half == "f32x3" else 1e-3
    assert eb < tol and ed < tol
    if prec_half == "f32x3":
        assert m.engine().last_fallbacks == 0
        m.engine().set_options({"x3_attn_p16": 2})
        with torch.inference_mode():
            r2 = m(x.to(dev()))
        e2 = max(float((r2["beat"].cpu() - ob).abs().max()), float((r2["downbeat"].cpu() - od).abs().max()))
        report("ablation_p16_frontend_too", variant=variant, err=e2)
        assert e2 < 7.5e-4


@pytest.mark.parametrize("B,T", [(1, 37), (2, 1), (5, 333), (33, 64)])
def test_forward_half_odd_batches_match_fp32_path(B, T):
    """Ragged shapes through the half-precision kernels (partial 32-token blocks, partial GEMM tiles, more chunks than one tile
    row): bounded against the exact-fp32 path of the same engine."""
    from beat_this_amd import weights as W

    m = _model("final0", 1, "lively")
    x = torch.from_numpy(np.stack([W.synthetic_spect(T, seed=300 + i) for i in range(B)])).to(dev())
    with torch.inference_mode():
        ref = m(x)
        with torch.autocast("cuda", enabled=True):
            got = m(x)
    assert got["beat"].shape == (B, T) and torch.isfinite(got["beat"]).all() and torch.isfinite(got["downbeat"]).all()
    err = float((got["beat"] - ref["beat"]).abs().max())
    spread = float(ref["beat"].std()) if B * T > 1 else 0.0
    report("forward_half_odd", B=B, T=T, err_beat=err, spread=spread)
    assert err < 0.25 * max(spread, 0.4)


def test_audio2beats_many_matches_single_track_calls_and_oracle():
    """The batched track API (one launch per stage for a list of tracks, async device-to-host copy) against one
    Audio2Beats.__call__ per track: identical logits and beats; ragged lengths incl. a short piece (one odd-length chunk),
    a piece of exactly 1488 frames + 1, 44.1 kHz stereo input through the batched resampler; and against the oracle."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams("small0")
    sd = W.random_state_dict(hp, seed=4, style="lively")
    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=False, dbn=False)
    a2b.model = _model("small0", 4, "lively")
    secs = (7.0, 29.77, 65.3, 31.0, 2.0)
    sigs = [W.synthetic_audio(s, seed=40 + i) for i, s in enumerate(secs)]
    many = a2b.many(sigs, 22050)
    assert len(many) == len(sigs)
    spect, foff = a2b.signal2spect_many(sigs, 22050)
    bl, dl = a2b.spect2frames_batch(spect, foff)
    worst = 0.0
    for i, sig in enumerate(sigs):
        b1, d1 = a2b(sig, 22050)
        assert np.array_equal(many[i][0], b1) and np.array_equal(many[i][1], d1), i
        s1 = a2b.signal2spect(sig, 22050)
        assert torch.equal(spect[foff[i]: foff[i + 1]], s1), i
        lb, ld = a2b.spect2frames(s1)
        worst = max(worst, float((bl[foff[i]: foff[i + 1]] - lb).abs().max()))
    assert worst <= 1e-5
    with torch.inference_mode():
        ob, od = O.spect2frames(sd, O.logmel(torch.from_numpy(sigs[2])))
    err = float((bl[foff[2]: foff[3]].cpu() - ob).abs().max())
    obeats, odown = O.postp_minimal(ob, od)
    report("a2b_many", worst_vs_single=worst, err_vs_oracle=err, beats=len(obeats))
    assert err < LOGIT_TOL_F32 and np.array_equal(many[2][0], obeats) and np.array_equal(many[2][1], odown)
    # 44.1 kHz stereo numpy + torch device tensors, mixed in one batch
    s44 = [np.stack([W.synthetic_audio(9.0, seed=60, sr=44100)] * 2, 1).astype(np.float64),
           torch.from_numpy(W.synthetic_audio(40.0, seed=61, sr=44100)).to(dev())]
    m44 = a2b.many(s44, 44100)
    for i, sig in enumerate(s44):
        ref = a2b(sig.cpu().numpy() if isinstance(sig, torch.Tensor) else sig, 44100)
        assert np.array_equal(m44[i][0], ref[0]) and np.array_equal(m44[i][1], ref[1]), i
    ob, od = O.audio2frames(sd, s44[1].cpu().numpy(), 44100)
    fr = a2b.many_async(s44[1:], 44100)
    fr.result()
    e44 = float((fr.logits[0].cpu() - ob).abs().max())
    report("a2b_many_44k1", err_vs_oracle=e44)
    assert e44 < LOGIT_TOL_F32
    assert a2b.many([], 22050) == []
    # host buffers: pinned tensors (uploaded on the copy stream, two batches in flight), plain CPU tensors and numpy in
    # one batch give what the device-resident waveforms give
    host = [torch.from_numpy(sigs[2]).pin_memory(), torch.from_numpy(sigs[3]), sigs[0], torch.from_numpy(sigs[1]).pin_memory()]
    order = (2, 3, 0, 1)
    first, second = a2b.many_async(host, 22050), a2b.many_async(host, 22050)
    for res in (first.result(), second.result()):
        for k, i in enumerate(order):
            assert np.array_equal(res[k][0], many[i][0]) and np.array_equal(res[k][1], many[i][1]), (k, i)


@pytest.mark.parametrize("name", ["small0", "final0"])
def test_submodules_are_callable_like_the_reference(name):
    """BeatThis.frontend / .transformer_blocks / .task_heads called on their own (beat_tracker.py:188-192), each against
    the oracle's stage on the oracle's input, their composition against the one-call forward, hooks, half precision."""
    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=5, style="lively")
    m = _model(name, 5, "lively")
    D = hp["transformer_dim"]
    x = torch.from_numpy(np.stack([W.synthetic_spect(333, seed=70 + i) for i in range(2)]))
    with torch.inference_mode():
        o_front = O.frontend(x, sd)
        o_tr = O.transformer(o_front, sd, hp["n_layers"], D // 32)
        bd = o_tr @ sd["task_heads.beat_downbeat_lin.weight"].T + sd["task_heads.beat_downbeat_lin.bias"]
        o_beat, o_down = bd[..., 0] + bd[..., 1], bd[..., 1]
        g_front = m.frontend(x.to(dev()))
        g_tr = m.transformer_blocks(o_front.to(dev()))
        g_head = m.task_heads(o_tr.to(dev()))
        whole = m(x.to(dev()))
        chain = m.task_heads(m.transformer_blocks(m.frontend(x.to(dev()))))
    assert g_front.shape == (2, 333, D) and g_tr.shape == (2, 333, D) and set(g_head) == {"beat", "downbeat"}
    e_front = float((g_front.cpu() - o_front).abs().max()) / float(o_front.abs().max())
    e_tr = float((g_tr.cpu() - o_tr).abs().max()) / float(o_tr.abs().max())
    e_head = max(float((g_head["beat"].cpu() - o_beat).abs().max()), float((g_head["downbeat"].cpu() - o_down).abs().max()))
    e_chain = max(float((chain[k] - whole[k]).abs().max()) for k in ("beat", "downbeat"))
    report("stages", model=name, frontend_rel=e_front, transformer_rel=e_tr, head_abs=e_head, chain_vs_forward=e_chain)
    assert e_front < 2e-5 and e_tr < 2e-5 and e_head < 1e-4 and e_chain < 1e-4
    # hooks on a sub-module fire in the whole forward too (it then runs stage by stage) and see the stage's output
    seen = []
    h = m.transformer_blocks.register_forward_hook(lambda mod, inp, out: seen.append((tuple(inp[0].shape), tuple(out.shape))))
    with torch.inference_mode():
        hooked = m(x.to(dev()))
    h.remove()
    assert seen == [((2, 333, D), (2, 333, D))]
    assert max(float((hooked[k] - whole[k]).abs().max()) for k in ("beat", "downbeat")) < 1e-4
    # half precision: the stage entry rebuilds what the fused producer leaves (half shadow, partial sums of squares)
    with torch.inference_mode(), torch.autocast("cuda", enabled=True):
        whole_h = m(x.to(dev()))
        chain_h = m.task_heads(m.transformer_blocks(m.frontend(x.to(dev()))))
    e_half = max(float((chain_h[k] - whole_h[k]).abs().max()) for k in ("beat", "downbeat"))
    report("stages_half", model=name, chain_vs_forward=e_half)
    # (the partial sums of squares are added in another order than the fused producer's: a last-bit difference that the
    # fp16 roundings of six layers turn into ~2e-3 on these noise-like weights -- the half path's own error is 6e-3)
    assert e_half < 6e-3


@pytest.mark.parametrize("name", ["small0", "final0"])
def test_fp32_split_gemms_stay_within_the_fp32_gate(name):
    """BT_PREC_F32X3 (the fp32 path with its plain GEMMs on three half MFMAs per product, operands split into hi + lo):
    logits against the oracle and against the exact fp32 path, beats identical."""
    from beat_this_amd import weights as W
    from beat_this_amd.postprocessor import Postprocessor
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=6, style="lively")
    m = _model(name, 6, "lively")
    x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=75 + i) for i in range(2)]))
    assert m.fp32_split_gemms          # (the module's own default since round 5)
    with torch.inference_mode():
        ob, od = O.model_forward(sd, x)
        m.fp32_split_gemms = False
        exact = m(x.to(dev()))
        m.fp32_split_gemms = True
        split = m(x.to(dev()))
        m.engine().set_options({"x3_attn_p16": 0})     # three-term P.V (rounds 3 - 4): the tighter variant of the same path
        split3 = m(x.to(dev()))
        m.engine().set_options({"x3_attn_p16": 1})
        m.fp32_split_gemms = False
    e3 = max(float((split3["beat"].cpu() - ob).abs().max()), float((split3["downbeat"].cpu() - od).abs().max()))
    assert e3 < 1e-4 and not torch.equal(split3["beat"], split["beat"])
    e_oracle = max(float((split["beat"].cpu() - ob).abs().max()), float((split["downbeat"].cpu() - od).abs().max()))
    e_exact = max(float((split[k] - exact[k]).abs().max()) for k in ("beat", "downbeat"))
    post = Postprocessor()
    flips = 0
    for i in range(2):
        b0, d0 = post(exact["beat"][i], exact["downbeat"][i])
        b1, d1 = post(split["beat"][i], split["downbeat"][i])
        flips += len(set(np.round(b0 * 100).astype(int)) ^ set(np.round(b1 * 100).astype(int)))
        flips += len(set(np.round(d0 * 100).astype(int)) ^ set(np.round(d1 * 100).astype(int)))
    report("f32x3", model=name, err_vs_oracle=e_oracle, err_vs_exact_fp32=e_exact, flips_vs_exact=flips, err_three_term_vs_oracle=e3)
    assert e_oracle < X3_TOL and flips == 0
    # the user-facing switch: float16="f32x3" selects it without autocast
    from beat_this_amd.inference import Spect2Frames

    s2f = Spect2Frames(checkpoint_path={"hyper_parameters": dict(hp), "state_dict": {"model." + k: v for k, v in sd.items()}},
                       device=dev(), float16="f32x3")
    assert s2f.float16 is False and s2f.model.fp32_split_gemms
    b2, d2 = s2f.spect2frames(x[0].to(dev()))
    s2f.model.fp32_split_gemms = False
    b3, d3 = s2f.spect2frames(x[0].to(dev()))   # the exact fp32 path through the same chunking
    assert float((b2 - b3).abs().max()) < X3_TOL and float((d2 - d3).abs().max()) < X3_TOL and not torch.equal(b2, b3)


@pytest.mark.parametrize("name", ["small0", "final0"])
def test_f32x3_range_guard_falls_back_to_exact_fp32(name):
    """Operands beyond the fp16 range of a hi part (here: a residual stream of ~1e6, which fp32 arithmetic -- and the
    reference on a CPU -- handles fine) must not produce wrong numbers silently: the forward's range flag fires, the
    engine repeats the batch on the exact fp32 MFMA path (bit-identical to calling that path directly), and the batched
    track API does the same from its deferred flags."""
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from beat_this_amd.model import BeatThis

    hp = W.resolve_hparams(name)
    sd = W.random_state_dict(hp, seed=6, style="lively")
    sd["frontend.linear.weight"] = sd["frontend.linear.weight"] * 3.0e5
    sd["frontend.linear.bias"] = sd["frontend.linear.bias"] * 3.0e5
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(sd)
    m = m.to(dev())
    x = torch.from_numpy(np.stack([W.synthetic_spect(900, seed=80 + i) for i in range(2)])).to(dev())
    with torch.inference_mode():
        m.fp32_split_gemms = False
        exact = m(x)
        assert torch.isfinite(exact["beat"]).all()
        m.fp32_split_gemms = True
        before = m.engine().last_fallbacks
        split = m(x)
    assert m.engine().last_fallbacks == before + 1
    assert torch.equal(split["beat"], exact["beat"]) and torch.equal(split["downbeat"], exact["downbeat"])
    # ordinary weights: the flag stays down and the result is NOT the exact path's bit pattern (the split path really ran)
    m2 = _model(name, 6, "lively")
    with torch.inference_mode():
        m2.fp32_split_gemms = False
        e2 = m2(x)
        m2.fp32_split_gemms = True
        s2 = m2(x)
    assert m2.engine().last_fallbacks == 0 and not torch.equal(s2["beat"], e2["beat"])
    assert float((s2["beat"] - e2["beat"]).abs().max()) < X3_TOL
    # batched track API (deferred flags): same beats as the exact path
    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16="f32x3")
    a2b.model = m
    sigs = [W.synthetic_audio(40.0, seed=90), W.synthetic_audio(12.0, seed=91)]
    before = m.engine().last_fallbacks
    got = a2b.many(sigs, 22050)
    assert m.engine().last_fallbacks > before
    # ... the handle's logits are those of the repeat, not the overflowed ones
    h = a2b.many_async(sigs, 22050)
    h.result()
    assert all(torch.isfinite(t).all() for t in h.logits[:2])
    # a DBN-type post-processor (madmom is not installed: a stand-in that only looks at what it is given) runs on the host
    # right away -- it must see the repeated, finite logits, never the overflowed ones
    seen = []

    class _FakeDBN:
        type = "dbn"

        def __call__(self, beat, downbeat):
            seen.append(bool(torch.isfinite(beat).all() and torch.isfinite(downbeat).all()))
            return np.zeros(0), np.zeros(0)
    minimal = a2b.frames2beats
    a2b.frames2beats = _FakeDBN()
    before = m.engine().last_fallbacks
    hd = a2b.many_async(sigs, 22050)
    hd.result()
    assert m.engine().last_fallbacks > before and seen == [True, True]
    dbn_logits = hd.logits
    a2b.frames2beats = minimal
    m.fp32_split_gemms = False
    want = a2b.many(sigs, 22050)
    for (gb, gd), (wb, wd) in zip(got, want):
        assert np.array_equal(gb, wb) and np.array_equal(gd, wd)
    hw = a2b.many_async(sigs, 22050)
    hw.result()
    assert torch.equal(h.logits[0], hw.logits[0]) and torch.equal(dbn_logits[0], hw.logits[0])
    # stage calls in BT_PREC_F32X3 (bt_forward_stages with last < 2) are guarded as well: frontend / transformer_blocks
    # called on their own return the exact path's result, not inf / NaN
    with torch.inference_mode():
        fe = m.frontend(x)
        te = m.transformer_blocks(fe)
        m.fp32_split_gemms = True
        before = m.engine().last_fallbacks
        fs = m.frontend(x)
        ts = m.transformer_blocks(fe)
    assert m.engine().last_fallbacks == before + 2
    assert torch.equal(fs, fe) and torch.equal(ts, te) and torch.isfinite(ts).all()


@pytest.mark.parametrize("mode", ["fp32", "half", "f32x3"])
def test_forward_takes_sequences_longer_than_a_chunk(mode):
    """BeatThis.forward on 2100 frames in one item (the reference's module takes any length, beat_tracker.py:188-192; its
    inference classes only ever feed 1500): the rotary table grows on demand, every kernel is length-agnostic."""
    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    hp = W.resolve_hparams("small0")
    sd = W.random_state_dict(hp, seed=2, style="lively")
    m = _model("small0", 2, "lively")
    m.fp32_split_gemms = mode == "f32x3"
    x = torch.from_numpy(W.synthetic_spect(2100, seed=41))[None]
    with torch.inference_mode(), torch.autocast("cuda", enabled=mode == "half"):
        r = m(x.to(dev()))
        r1500 = m(x[:, :1500].to(dev()))   # (and the ordinary length still works on the grown table)
    with torch.inference_mode():
        ob, od = O.model_forward(sd, x)
    err = max(float((r["beat"].cpu() - ob).abs().max()), float((r["downbeat"].float().cpu() - od).abs().max()))
    report("long_sequence", mode=mode, T=2100, max_abs_logit=err)
    assert r["beat"].shape == (1, 2100) and r1500["beat"].shape == (1, 1500)
    assert err < (2e-2 if mode == "half" else LOGIT_TOL_F32 if mode == "fp32" else X3_TOL)


def test_empty_inputs_and_cpu_device():
    from beat_this_amd.inference import Spect2Frames

    s2f = Spect2Frames(checkpoint_path=None, device="cuda:0")
    b, d = s2f(torch.zeros((0, 128), device=dev()))
    assert b.shape == (0,) and d.shape == (0,) and b.dtype == torch.float32
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        Spect2Frames(checkpoint_path=None, device="cpu")


def test_ff_mult_other_than_four_loads_and_matches_oracle():
    """A checkpoint with ff_mult != 4: the frontend's FeedForward stays at 4 x dim (beat_tracker.py:279,288),
    the main layers use ff_mult."""
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis
    from oracle import beat_this_oracle as O

    hp = dict(W.resolve_hparams("small0"), ff_mult=2)
    sd = W.random_state_dict(hp, seed=9, style="lively")
    assert sd["transformer_blocks.layers.0.1.net.1.weight"].shape == (256, 128)
    assert sd["frontend.blocks.0.partial.ffF.net.1.weight"].shape == (128, 32)
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(sd)
    m = m.to(dev())
    x = torch.from_numpy(W.synthetic_spect(600, seed=2))[None]
    with torch.inference_mode():
        r = m(x.to(dev()))
        ob, od = O.model_forward(sd, x)
        with torch.autocast("cuda", enabled=True):
            rh = m(x.to(dev()))
    err = float((r["beat"].cpu() - ob).abs().max())
    errh = float((rh["beat"].cpu() - ob).abs().max())
    report("ff_mult2", err_f32=err, err_half=errh)
    assert err < LOGIT_TOL_F32 and errh < 0.1


def test_bare_load_model_runs_the_default_precision_with_its_range_guard():
    """what hubconf.py exports (``load_model``) and what pl_module.py:264 does with it
    (``split_predict_aggregate(spect, 1500, 6, "keep_first", model)``) run BT_PREC_F32X3 -- not the three times slower exact
    path -- and keep its guard: a residual stream of 1e6 is repeated on the exact path (inference.py:56-87, 188-230)."""
    from beat_this_amd import _lib
    from beat_this_amd import weights as W
    from beat_this_amd.inference import load_model, split_predict_aggregate

    hp = W.resolve_hparams("small0")
    sd = W.random_state_dict(hp, seed=6, style="lively")

    def ckpt(state):
        return {"hyper_parameters": dict(hp), "state_dict": {"model." + k: v for k, v in state.items()}}
    m = load_model(ckpt(sd), dev())
    assert m.fp32_split_gemms and (m._precision() == _lib.PREC_F32X3 or _lib.lib().bt_half_is_bf16())
    spect = torch.from_numpy(W.synthetic_spect(3100, seed=9)).to(dev())
    with torch.inference_mode():
        r = split_predict_aggregate(spect, 1500, 6, "keep_first", m)
        m.fp32_split_gemms = False
        e = split_predict_aggregate(spect, 1500, 6, "keep_first", m)
        m.fp32_split_gemms = True
    assert m.engine().last_fallbacks == 0
    assert not torch.equal(r["beat"], e["beat"]) and float((r["beat"] - e["beat"]).abs().max()) < X3_TOL
    sd2 = dict(sd)
    sd2["frontend.linear.bias"] = sd["frontend.linear.bias"] * 3.0e5
    m2 = load_model(ckpt(sd2), dev())
    with torch.inference_mode():
        r2 = split_predict_aggregate(spect, 1500, 6, "keep_first", m2)
        assert m2.engine().last_fallbacks >= 1
        m2.fp32_split_gemms = False
        e2 = split_predict_aggregate(spect, 1500, 6, "keep_first", m2)
    assert torch.isfinite(r2["beat"]).all() and torch.equal(r2["beat"], e2["beat"]) and torch.equal(r2["downbeat"], e2["downbeat"])


@pytest.mark.parametrize("mode", [False, True, "exact"])
def test_single_file_path_on_a_captured_forward_matches_plain_launches(mode):
    """Pieces of up to 11 chunks run their forward as one hipGraph (pack.Engine.graph_forward: chunk gather into the graph's
    input, replay, aggregation from its outputs).  Same kernels, same arguments: the logits must equal the plain launches'
    bit for bit, on the first call (capture) and on replays, for several piece lengths sharing / not sharing an entry, and
    the result must not be disturbed by a later call of another length (entries own their buffers)."""
    from beat_this_amd import inference as inf
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Spect2Frames

    s2f = Spect2Frames(checkpoint_path=None, device=dev(), float16=mode)
    s2f.model = _model("small0", 1, "lively")
    pieces = [torch.from_numpy(W.synthetic_spect(n, seed=70 + i)).to(dev()) for i, n in enumerate((3100, 1501, 2000, 16000, 700))]
    assert inf.USE_GRAPHS
    got = [s2f(p) for p in pieces]            # captures (3 chunks, 2, 2 again = replay of the same entry, 11; the odd-length chunk runs plain)
    again = [s2f(p) for p in pieces]          # replays
    eng = s2f.model.engine()
    assert getattr(eng, "_graphs_ok", True), getattr(eng, "_graph_error", "")
    assert len(eng.__dict__.get("_graphs", {})) == 3     # (only full-length chunks are captured: Engine.GRAPH_T)
    assert len({id(e.ws) for e in eng.__dict__["_graphs"].values()}) <= 2   # entries of a stream share a workspace (a larger need replaces it)
    inf.USE_GRAPHS = False
    try:
        plain = [s2f(p) for p in pieces]
    finally:
        inf.USE_GRAPHS = True
    for (b1, d1), (b2, d2), (b0, d0) in zip(got, again, plain):
        assert torch.equal(b1, b0) and torch.equal(d1, d0) and torch.equal(b2, b0) and torch.equal(d2, d0)


@pytest.mark.parametrize("mode", [False, True, "exact"])
def test_audio2beats_one_call_is_bit_identical_to_the_stage_by_stage_path(mode):
    """Audio2Beats.__call__ as ONE library call (bt_audio2beats_enqueue: resample -> log-mel -> split -> forward, replayed as a
    hipGraph the library captures itself -> aggregate -> peaks -> one D2H copy) against the stage-by-stage path of rounds 1 - 5
    (one library call per stage from Python): identical beat / downbeat times AND identical framewise logits (read back from the
    call's workspace), for short clips (one odd-length chunk), pieces around the 1488 / 1489-frame edge, multi-chunk pieces,
    44.1 kHz / 48 kHz / stereo float64 input, repeated calls (graph replays) and interleaved lengths (same chunk count, other
    length: the captured forward is reused; other chunk count: a new capture)."""
    import ctypes as C

    from beat_this_amd import _lib as Lb
    from beat_this_amd import inference as inf
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats

    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=mode, dbn=False)
    a2b.model = _model("small0", 4, "lively")
    cases = [(7.0, 22050, 1), (29.75, 22050, 1), (29.77, 22050, 1), (40.0, 22050, 1), (41.3, 22050, 1), (65.3, 44100, 1),
             (12.0, 48000, 2), (40.0, 22050, 1), (100.0, 44100, 1), (7.0, 22050, 1)]
    n_fast = 0
    for secs, sr, ch in cases:
        sig = W.synthetic_audio(secs, seed=int(secs * 10) + sr // 1000, sr=sr)
        if ch == 2:
            sig = np.stack([sig, 0.5 * sig[::-1]], 1).astype(np.float64)
        inf.USE_ONE_CALL = False
        try:
            want_b, want_d = a2b(sig, sr)
            lb, ld = a2b.spect2frames(a2b.signal2spect(sig, sr))
        finally:
            inf.USE_ONE_CALL = True
        fast = a2b._one_call(sig, sr)
        assert fast is not None
        n_fast += 1
        assert np.array_equal(fast[0], want_b) and np.array_equal(fast[1], want_d), (secs, sr, ch)
        got_b, got_d = a2b(sig, sr)            # (the public call takes the same route)
        assert np.array_equal(got_b, want_b) and np.array_equal(got_d, want_d)
        # framewise logits of the call, from its workspace
        eng = a2b.model.engine()
        with torch.autocast("cuda", enabled=a2b.float16):
            prec = a2b.model._precision()
        plan = Lb.A2BPlan()
        from math import gcd
        g = gcd(sr, 22050)
        Lb.check(Lb.lib().bt_audio2beats_plan(eng._h, sig.shape[0], 22050 // g, sr // g, prec, C.byref(plan)))
        ws = eng._a2b_workspace(plan.ws_bytes)
        logits = ws[plan.off_logits: plan.off_logits + 8 * plan.n_frames].view(torch.float32)
        assert plan.n_frames == lb.shape[0]
        assert torch.equal(logits[: plan.n_frames], lb) and torch.equal(logits[plan.n_frames:], ld), (secs, sr, ch)
    report("a2b_one_call", mode=str(mode), cases=n_fast)


def test_audio2beats_one_call_repeats_an_overflowing_track_on_the_exact_path():
    """The range flag of a BT_PREC_F32X3 forward travels with the peak frames; a call whose flag is up is repeated on the exact
    fp32 path (same result as float16="exact"), and the fallback is counted."""
    from beat_this_amd import inference as inf
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from beat_this_amd.model import BeatThis

    hp = W.resolve_hparams("small0")
    sd = W.random_state_dict(hp, seed=6, style="lively")
    sd["frontend.linear.weight"] = sd["frontend.linear.weight"] * 3.0e5
    sd["frontend.linear.bias"] = sd["frontend.linear.bias"] * 3.0e5
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(sd)
    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=False)
    a2b.model = m.to(dev())
    sig = W.synthetic_audio(40.0, seed=90)
    before = m.engine().last_fallbacks
    got = a2b(sig, 22050)
    assert m.engine().last_fallbacks == before + 1
    ex = Audio2Beats(checkpoint_path=None, device=dev(), float16="exact")
    ex.model = m
    inf.USE_ONE_CALL = False
    try:
        want = ex(sig, 22050)
    finally:
        inf.USE_ONE_CALL = True
        m.fp32_split_gemms = True
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.timeout(300)
def test_interleaved_graph_replays_on_the_default_stream_never_raise_the_range_flag():
    """Round 6, found by tools/length_fuzz.py: with the range flag of a BT_PREC_F32X3 forward cleared by hipMemsetAsync, the memset NODE at
    the head of a captured forward misbehaved on the legacy default stream once replays of OTHER captured graphs (another engine's, the
    stage-by-stage route's) had run in between -- the flag word read 0x01010101, the logits were those of a damaged forward, and the call
    was (correctly, but needlessly) repeated on the exact path: 3 of 40 files in this sequence.  The flag is cleared by a one-block launch
    since (bt_forward_stages); the sequence must run without a single fallback, on the default stream, with both engines interleaved."""
    from beat_this_amd import inference as inf
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats
    from beat_this_amd.model import BeatThis

    hp = W.resolve_hparams("final0")
    sd = W.random_state_dict(hp, seed=1, style="outlier")

    def make(f16):
        a = Audio2Beats(checkpoint_path=None, device=dev(), float16=f16, dbn=False)
        m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
        m.load_state_dict(sd)
        a.model = m.to(dev())
        return a

    assert torch.cuda.current_stream(dev()).cuda_stream == 0, "this test is about the legacy default stream"
    fast, exact = make(False), make("exact")
    eng = fast.model.engine()
    rng = np.random.default_rng(1)
    n_calls = 0
    for i in range(36):   # (the sequence of tools/length_fuzz.py 40 outlier 1: files 28, 30 and 34 fell back)
        secs = float(rng.uniform(0.3, 60.0) if i % 2 == 0 else rng.uniform(60.0, 400.0))
        sr = int(rng.choice([22050, 44100, 44100, 48000, 16000]))
        sig = W.synthetic_audio(secs, seed=1000 + i, sr=sr)
        inf.Audio2Frames.__call__(exact, sig, sr)     # the other engine's stage-by-stage route (a torch-level graph replay)
        b1, d1 = fast(sig, sr)                        # one library call: the replay under test
        b2, d2 = exact(sig, sr)                       # the other engine's one-call route
        n_calls += 1
        assert eng.last_fallbacks == 0, f"file {i} ({secs:.2f} s at {sr} Hz) raised the range flag"
        assert abs(len(b1) - len(b2)) <= 2 and abs(len(d1) - len(d2)) <= 2
    report("interleaved_default_stream", calls=n_calls, fallbacks=eng.last_fallbacks)


@pytest.mark.timeout(300)
def test_one_audio2beats_shared_by_host_threads():
    """One Audio2Beats (one engine) called from several host threads, each on its own stream, while files of different chunk counts make
    the one-call path capture and replay forwards: every result equals the single-threaded one.  (Round 6: the graph cache, the capture
    stream and the pinned-buffer pool were unguarded -- `hipErrorIllegalState` from a launch into another thread's capture.)"""
    import threading

    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats

    a2b = Audio2Beats(checkpoint_path=None, device=dev(), float16=False, dbn=False)
    a2b.model = _model("small0", 4, "lively")
    rng = np.random.default_rng(3)
    files = []
    for i in range(10):
        secs = float(rng.uniform(0.5, 45.0) if i % 2 else rng.uniform(45.0, 150.0))
        sr = int(rng.choice([22050, 44100]))
        files.append((W.synthetic_audio(secs, seed=3000 + i, sr=sr), sr))
    want = [a2b(sig, sr) for sig, sr in files]
    n_threads, errors, results = 3, [], [None] * 3

    def work(t):
        try:
            out = []
            with torch.cuda.stream(torch.cuda.Stream(device=dev())):
                order = list(range(len(files)))
                np.random.default_rng(t).shuffle(order)
                for j in order:
                    out.append((j, a2b(*files[j])))
            results[t] = out
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for out in results:
        for j, (b, d) in out:
            assert np.array_equal(b, want[j][0]) and np.array_equal(d, want[j][1]), j
    report("shared_engine_threads", threads=n_threads, files=len(files))

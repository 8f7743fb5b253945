"""CPU oracle for the beat_this inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (functional, state-dict driven, torch-CPU, fp32 or fp64)
of the arithmetic of the reference's hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``beat_this_amd`` never does (it fails loudly if its HIP
library is missing).

Pinning status: the reference's own tests hold no numerical fixtures for this path
(tests/test_inference.py:14-15,24-25 assert types only), so the pins are outputs of
the reference code itself, run in the build container by ``oracle/make_golden.py``
(unmodified ``/root/reference`` + the three third-party stand-ins in
``oracle/shims``) and committed under ``tests/golden/``.  ``tests/test_oracle.py``
checks this restatement against those fixtures everywhere, and against the live
reference whenever ``/root/reference`` exists.  The third-party leaves (torchaudio
2.3.1 MelSpectrogram, rotary-embedding-torch 0.6.4, soxr 0.3.7) are absent from the
image: their restatement is "parity unpinned" (SURVEY.md 8c).

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 22050
N_FFT = 1024
HOP = 441
N_MELS = 128
F_MIN = 30.0
F_MAX = 11000.0
LOG_MULT = 1000.0
FPS = 50
CHUNK = 1500
BORDER = 6


# --------------------------------------------------------------------------------------
# log-mel front end: beat_this/preprocessing.py:27-59 (-> torchaudio MelSpectrogram)
# --------------------------------------------------------------------------------------
def mel_filterbank(dtype=torch.float32) -> torch.Tensor:
    """(513,128) slaney triangular filterbank, norm=None.

    torchaudio.functional.melscale_fbanks as called through preprocessing.py:43-53
    (n_freqs=513, f_min=30, f_max=11000, n_mels=128, sr=22050, mel_scale="slaney").
    Always evaluated in fp32 like torchaudio does, then cast.
    """
    f_sp = 200.0 / 3.0
    logstep = math.log(6.4) / 27.0

    def hz2mel(f):
        return 15.0 + math.log(f / 1000.0) / logstep if f >= 1000.0 else f / f_sp

    all_freqs = torch.linspace(0, SAMPLE_RATE // 2, N_FFT // 2 + 1)
    m_pts = torch.linspace(hz2mel(F_MIN), hz2mel(F_MAX), N_MELS + 2)
    f_pts = f_sp * m_pts
    hi = m_pts >= 15.0
    f_pts[hi] = 1000.0 * torch.exp(logstep * (m_pts[hi] - 15.0))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp_min(torch.minimum(down, up), 0.0).to(dtype)


def n_frames(n_samples: int) -> int:
    """torch.stft(center=True): 1 + floor(N / hop)."""
    return 1 + n_samples // HOP


def logmel(x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """LogMelSpect.forward (preprocessing.py:56-59): (N,) waveform @22.05 kHz -> (frames,128).

    Explicit framing instead of torch.stft: reflect-pad n_fft/2, frames at hop 441,
    periodic Hann, rFFT, magnitude, * 1/sqrt(n_fft) (normalized="frame_length"),
    mel projection, log1p(1000 * .).
    """
    x = x.to(dtype)
    xp = F.pad(x[None, None, :], (N_FFT // 2, N_FFT // 2), mode="reflect")[0, 0]
    frames = xp.unfold(0, N_FFT, HOP)  # (n, 1024)
    win = torch.hann_window(N_FFT, periodic=True, dtype=dtype)
    mag = torch.fft.rfft(frames * win, dim=-1).abs() * (1.0 / math.sqrt(N_FFT))
    mel = mag @ mel_filterbank(dtype)
    return torch.log1p(LOG_MULT * mel)


# --------------------------------------------------------------------------------------
# chunking glue: beat_this/inference.py:90-185
# --------------------------------------------------------------------------------------
def split_starts(n: int, chunk: int = CHUNK, border: int = BORDER) -> np.ndarray:
    """split_piece start indices (inference.py:120-125, avoid_short_end=True)."""
    starts = np.arange(-border, n - border, chunk - 2 * border)
    if n > chunk - 2 * border:
        starts[-1] = n - (chunk - border)
    return starts


def split_chunks(spect: torch.Tensor, chunk: int = CHUNK, border: int = BORDER):
    """split_piece (inference.py:100-135): list of zero-padded chunks + starts."""
    n = spect.shape[0]
    starts = split_starts(n, chunk, border)
    chunks = []
    for s in starts:
        s = int(s)
        piece = spect[max(s, 0): min(s + chunk, n)]
        left = max(0, -s)
        right = max(0, min(border, s + chunk - n))
        chunks.append(F.pad(piece, (0, 0, left, right)))
    return chunks, starts


def aggregate(pred_chunks, starts, full_size: int, chunk: int = CHUNK, border: int = BORDER):
    """aggregate_prediction, overlap_mode="keep_first" (inference.py:138-185)."""
    beat = torch.full((full_size,), -1000.0)
    down = torch.full((full_size,), -1000.0)
    for s, (b, d) in reversed(list(zip(starts, pred_chunks))):
        s = int(s)
        b = b[border:-border]
        d = d[border:-border]
        beat[s + border: s + chunk - border] = b.float()
        down[s + border: s + chunk - border] = d.float()
    return beat, down


# --------------------------------------------------------------------------------------
# model: beat_this/model/roformer.py, beat_tracker.py
# --------------------------------------------------------------------------------------
def rmsnorm(x, gamma):
    """RMSNorm.forward (roformer.py:22-32): F.normalize(x, dim=-1) * sqrt(dim) * gamma."""
    nrm = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return x / nrm * math.sqrt(x.shape[-1]) * gamma


def rope(t, freqs):
    """rotary-embedding-torch 0.6.4 rotate_queries_or_keys on (b,h,n,d), seq dim -2.

    Interleaved pairs, theta from the stored ``freqs`` (beat_tracker.py:52,
    roformer.py:121-123); angles are computed in fp32 then cast, as the library
    does under autocast(enabled=False).
    """
    n = t.shape[-2]
    ang = torch.arange(n, dtype=torch.float32)[:, None] * freqs.float()[None, :]
    cos = ang.cos().repeat_interleave(2, -1).to(t.dtype)
    sin = ang.sin().repeat_interleave(2, -1).to(t.dtype)
    te, to = t[..., 0::2], t[..., 1::2]
    rot = torch.stack((-to, te), dim=-1).flatten(-2)
    return t * cos + rot * sin


def _linear(x, w, b=None):
    """nn.Linear as the reference's modules run it on CPU: ONE mm / addmm on the batch folded into rows.  The reference's
    weights are Parameters with requires_grad, for which torch.matmul always folds a 3-D input (copying a non-contiguous
    one -- the rearranged views of PartialFTTransformer are, beat_tracker.py:293-300); the plain tensors of a state dict take
    the batched-matmul route for such inputs instead, which is 1.4 x slower per chunk.  Same arithmetic either way."""
    return F.linear(x if x.is_contiguous() else x.contiguous(), w, b)


def attention(x, sd, pfx, heads):
    """Attention.forward (roformer.py:114-132) on x: (b, n, dim)."""
    b, n, dim = x.shape
    xn = rmsnorm(x, sd[pfx + "norm.gamma"])
    qkv = _linear(xn, sd[pfx + "to_qkv.weight"])  # (b,n,3*h*d), split "(qkv h d)"
    d = qkv.shape[-1] // (3 * heads)
    qkv = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)  # qkv b h n d
    q, k, v = qkv[0], qkv[1], qkv[2]
    fr = sd[pfx + "rotary_embed.freqs"]
    q, k = rope(q, fr), rope(k, fr)
    if q.dtype == torch.float32:
        # Attend.forward (roformer.py:67-80) calls F.scaled_dot_product_attention (default scale d^-0.5, no mask): the same
        # fused CPU kernel the reference runs, so that the cpu_baseline timing of bench.py is the reference's own cost
        out = F.scaled_dot_product_attention(q, k, v)
    else:  # float64 evaluation (accuracy reference): the explicit definition of the same operator
        att = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
        out = att @ v
    gates = _linear(xn, sd[pfx + "to_gates.weight"], sd[pfx + "to_gates.bias"])  # (b,n,h)
    out = out * torch.sigmoid(gates).permute(0, 2, 1)[..., None]
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return _linear(out, sd[pfx + "to_out.0.weight"])


def feedforward(x, sd, pfx):
    """FeedForward.forward (roformer.py:38-61): RMSNorm, Linear, GELU(erf), Linear."""
    h = rmsnorm(x, sd[pfx + "net.0.gamma"])
    h = F.gelu(_linear(h, sd[pfx + "net.1.weight"], sd[pfx + "net.1.bias"]))
    return _linear(h, sd[pfx + "net.4.weight"], sd[pfx + "net.4.bias"])


def batchnorm(x, sd, pfx, ch_dim):
    """eval-mode BatchNorm (running stats, eps 1e-5) along ``ch_dim``."""
    shape = [1] * x.dim()
    shape[ch_dim] = -1
    mu = sd[pfx + "running_mean"].view(shape)
    var = sd[pfx + "running_var"].view(shape)
    return (x - mu) / torch.sqrt(var + 1e-5) * sd[pfx + "weight"].view(shape) + sd[pfx + "bias"].view(shape)


def stem(x, sd):
    """BeatThis.make_stem (beat_tracker.py:108-126): (b,t,128) -> (b,32,32,t)."""
    x = batchnorm(x.transpose(1, 2), sd, "frontend.stem.bn1d.", 1)  # b f t
    x = F.conv2d(x[:, None], sd["frontend.stem.conv2d.weight"], stride=(4, 1), padding=(0, 1))
    x = batchnorm(x, sd, "frontend.stem.bn2d.", 1)
    return F.gelu(x)


def partial_ft(x, sd, pfx):
    """PartialFTTransformer.forward (beat_tracker.py:290-301) on (b,c,f,t)."""
    b, c, f, t = x.shape
    heads = c // 32
    y = x.permute(0, 3, 2, 1).reshape(b * t, f, c)  # (b t) f c
    y = y + attention(y, sd, pfx + "attnF.", heads)
    y = y + feedforward(y, sd, pfx + "ffF.")
    y = y.view(b, t, f, c).permute(0, 2, 1, 3).reshape(b * f, t, c)  # (b f) t c
    y = y + attention(y, sd, pfx + "attnT.", heads)
    y = y + feedforward(y, sd, pfx + "ffT.")
    return y.view(b, f, t, c).permute(0, 3, 1, 2)  # b c f t


def frontend(x, sd, taps=None):
    """BeatThis.frontend (beat_tracker.py:54-80): stem, 3 x (partial, conv, BN, GELU), concat, linear."""
    x = stem(x, sd)
    if taps is not None:
        taps["stem"] = x
    for i in range(3):
        p = f"frontend.blocks.{i}."
        if p + "partial.attnF.norm.gamma" in sd:
            x = partial_ft(x, sd, p + "partial.")
        if taps is not None:
            taps[f"partial{i}"] = x
        x = F.conv2d(x, sd[p + "conv2d.weight"], stride=(2, 1), padding=(0, 1))
        x = F.gelu(batchnorm(x, sd, p + "norm.", 1))
        if taps is not None:
            taps[f"block{i}"] = x
    b, c, f, t = x.shape
    x = x.permute(0, 3, 1, 2).reshape(b, t, c * f)  # "b c f t -> b t (c f)"
    return _linear(x, sd["frontend.linear.weight"], sd["frontend.linear.bias"])


def transformer(x, sd, n_layers, heads, taps=None):
    """roformer.Transformer.forward (roformer.py:176-181)."""
    for l in range(n_layers):
        p = f"transformer_blocks.layers.{l}."
        x = attention(x, sd, p + "0.", heads) + x
        x = feedforward(x, sd, p + "1.") + x
        if taps is not None:
            taps[f"layer{l}"] = x
    return rmsnorm(x, sd["transformer_blocks.norm.gamma"])


def model_forward(sd: dict, x: torch.Tensor, dtype=torch.float32, taps=None, sum_head: bool = True):
    """BeatThis.forward (beat_tracker.py:188-192) with SumHead (:304-330), or Head (:333-346) for
    ``sum_head=False``; blocks without partial transformers (beat_tracker.py:143-153) are recognised by their keys.

    sd: reference-layout state dict; x: (B,T,128).
    Returns (beat, downbeat) each (B,T) in ``dtype``.
    """
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.to(dtype)
    dim = sd["transformer_blocks.norm.gamma"].shape[0]
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer_blocks.layers."))
    x = frontend(x, sd, taps)
    if taps is not None:
        taps["frontend"] = x
    x = transformer(x, sd, n_layers, dim // 32, taps)
    bd = _linear(x, sd["task_heads.beat_downbeat_lin.weight"], sd["task_heads.beat_downbeat_lin.bias"])
    beat, down = bd[..., 0], bd[..., 1]
    return (beat + down if sum_head else beat), down


def spect2frames(sd, spect, dtype=torch.float32):
    """Spect2Frames.spect2frames (inference.py:244-254): chunk, predict one by one, aggregate."""
    chunks, starts = split_chunks(spect)
    preds = []
    for c in chunks:
        b, d = model_forward(sd, c[None], dtype)
        preds.append((b[0], d[0]))
    return aggregate(preds, starts, spect.shape[0])


# --------------------------------------------------------------------------------------
# minimal post-processor: beat_this/model/postprocessor.py:85-136,176-197
# --------------------------------------------------------------------------------------
def deduplicate_peaks(peaks, width=1) -> np.ndarray:
    """postprocessor.py:176-197 -- merge runs against the *running mean*."""
    out = []
    it = iter(int(p) for p in peaks)
    try:
        mean = next(it)
    except StopIteration:
        return np.array(out)
    count = 1
    for nxt in it:
        if nxt - mean <= width:
            count += 1
            mean += (nxt - mean) / count
        else:
            out.append(mean)
            mean, count = nxt, 1
    out.append(mean)
    return np.array(out)


def peak_frames(logits: torch.Tensor) -> np.ndarray:
    """postp_minimal (postprocessor.py:93-99): x == maxpool7(x) and x > 0, as frame indices."""
    x = logits.float()[None, None]
    pooled = F.max_pool1d(x, 7, 1, 3)
    keep = (x == pooled) & (x > 0)
    return torch.nonzero(keep[0, 0])[:, 0].numpy()


def postp_minimal(beat: torch.Tensor, downbeat: torch.Tensor, fps: int = FPS):
    """Postprocessor("minimal") on unbatched logits (postprocessor.py:85-136)."""
    bt = deduplicate_peaks(peak_frames(beat)) / fps
    dt = deduplicate_peaks(peak_frames(downbeat)) / fps
    if len(bt) > 0:
        for i, d in enumerate(dt):
            dt[i] = bt[np.argmin(np.abs(bt - d))]
    return bt, np.unique(dt)


def audio2beats(sd, signal22k: np.ndarray, dtype=torch.float32):
    """Audio2Beats.__call__ (inference.py:301-303) from the 22.05 kHz waveform on."""
    spect = logmel(torch.as_tensor(signal22k, dtype=torch.float32))
    b, d = spect2frames(sd, spect, dtype)
    return postp_minimal(b, d)


def signal2spect(signal: np.ndarray, sr: int):
    """Audio2Frames.signal2spect (inference.py:269-277): mono mix, resample to 22.05 kHz (the soxr stand-in of
    oracle/shims -- parity with libsoxr itself is unpinned), float32, log-mel."""
    import importlib.util
    import os

    if signal.ndim == 2:
        signal = signal.mean(1)
    elif signal.ndim != 1:
        raise ValueError(f"Expected 1D or 2D signal, got shape {signal.shape}")
    if sr != SAMPLE_RATE:
        spec = importlib.util.spec_from_file_location(
            "_oracle_soxr_shim", os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims", "soxr", "__init__.py"))
        shim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shim)
        signal = shim.resample(signal, in_rate=sr, out_rate=SAMPLE_RATE)
    return logmel(torch.tensor(signal, dtype=torch.float32))


def audio2frames(sd, signal: np.ndarray, sr: int, dtype=torch.float32):
    """Audio2Frames.__call__ (inference.py:279-281)."""
    return spect2frames(sd, signal2spect(signal, sr), dtype)

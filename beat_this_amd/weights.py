"""Hyper-parameters and seeded synthetic state dicts in the reference checkpoint layout.

The checkpoint layout (166 entries for the default model) is the one
``BeatThis.state_dict()`` produces in the reference (beat_this/model/beat_tracker.py:38-106,
SURVEY.md Appendix A).  No checkpoint can be downloaded in this environment, so
benchmarks and tests use weights drawn from ``numpy.random.default_rng`` (PCG64 --
identical on every machine, unlike torch's generators) with the reference's
initialisation statistics (beat_tracker.py:170-186) or a "lively" variant whose
activations have O(1) dynamic range (sharper softmaxes, logits crossing 0).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

HPARAMS = {
    # README.md:231-243 / launch_scripts/train.py:81-100 of the reference
    "final0": dict(spect_dim=128, transformer_dim=512, ff_mult=4, n_layers=6, head_dim=32, stem_dim=32),
    "small0": dict(spect_dim=128, transformer_dim=128, ff_mult=4, n_layers=6, head_dim=32, stem_dim=32),
}


def resolve_hparams(hp=None) -> dict:
    base = dict(HPARAMS["final0"], sum_head=True, partial_transformers=True)
    if isinstance(hp, str):
        base.update(HPARAMS["small0"] if hp.startswith("small") else HPARAMS["final0"])
    elif hp:
        base.update({k: v for k, v in hp.items() if k in base})
    return base


def state_dict_shapes(hp: dict) -> "OrderedDict[str, tuple]":
    """Ordered {key: shape} of BeatThis.state_dict() for ``hp``."""
    hp = resolve_hparams(hp)
    D, L, hd, S, mel = hp["transformer_dim"], hp["n_layers"], hp["head_dim"], hp["stem_dim"], hp["spect_dim"]
    mult = hp["ff_mult"]
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def bn(p, n):
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[p + k] = (n,)
        out[p + "num_batches_tracked"] = ()

    def attn(p, dim):
        h = dim // hd
        out[p + "rotary_embed.freqs"] = (hd // 2,)
        out[p + "norm.gamma"] = (dim,)
        out[p + "to_qkv.weight"] = (3 * dim, dim)
        out[p + "to_gates.weight"] = (h, dim)
        out[p + "to_gates.bias"] = (h,)
        out[p + "to_out.0.weight"] = (dim, dim)

    def ff(p, dim, mult=4):  # (the frontend's FeedForward(dim) is always mult = 4: beat_tracker.py:279,288)
        out[p + "net.0.gamma"] = (dim,)
        out[p + "net.1.weight"] = (mult * dim, dim)
        out[p + "net.1.bias"] = (mult * dim,)
        out[p + "net.4.weight"] = (dim, mult * dim)
        out[p + "net.4.bias"] = (dim,)

    bn("frontend.stem.bn1d.", mel)
    out["frontend.stem.conv2d.weight"] = (S, 1, 4, 3)
    bn("frontend.stem.bn2d.", S)
    dim, f = S, mel // 4
    for i in range(3):
        p = f"frontend.blocks.{i}."
        if hp["partial_transformers"]:
            attn(p + "partial.attnF.", dim)
            ff(p + "partial.ffF.", dim)
            attn(p + "partial.attnT.", dim)
            ff(p + "partial.ffT.", dim)
        out[p + "conv2d.weight"] = (2 * dim, dim, 2, 3)
        bn(p + "norm.", 2 * dim)
        dim, f = 2 * dim, f // 2
    out["frontend.linear.weight"] = (D, dim * f)
    out["frontend.linear.bias"] = (D,)
    for l in range(L):
        attn(f"transformer_blocks.layers.{l}.0.", D)
        ff(f"transformer_blocks.layers.{l}.1.", D, mult)
    out["transformer_blocks.norm.gamma"] = (D,)
    out["task_heads.beat_downbeat_lin.weight"] = (2, D)
    out["task_heads.beat_downbeat_lin.bias"] = (2,)
    return out


def random_state_dict(hp=None, seed: int = 0, style: str = "lively") -> "OrderedDict[str, torch.Tensor]":
    """Seeded synthetic weights in checkpoint layout.

    style="init":   the reference's init statistics (Linear N(0,0.02^2), zero bias,
                    Conv2d Kaiming-normal fan_out, gamma=1, BN identity) plus a seeded
                    perturbation of every 1-D parameter and BN running statistic
                     so that no affine term is trivially 0/1.
    style="init0":  the reference's init statistics exactly (no perturbation) -- what
                    ``BeatThis()`` holds before a checkpoint is loaded.
    style="lively": fan-in scaled weights so that attention logits and the output
                    logits have O(1) spread (peaks, sign changes) -- a stricter
                    numerical test than "init", same FLOPs.
    style="outlier": "lively" with what trained transformers add to it (no checkpoint can be
                    fetched here, so the stress is synthesised): a few OUTLIER CHANNELS of the
                    residual stream carrying activations of 10^2 .. 10^3 (rows of
                    frontend.linear / to_out / net.4 scaled up, the RMSNorm gammas of those
                    channels scaled down, as trained models do), heavy-tailed (Student-t, 4
                    degrees of freedom) matrix entries, sharper attention (q / k rows x 1.25), and
                    frontend BatchNorm scales that push some channels to |a| ~ 10^2.  Exercises
                    the fp16 range and the hi + lo representation of BT_PREC_F32X3.  Kept WELL
                    CONDITIONED (fp32 and fp64 CPU forwards agree to 2e-6 at the logits), so
                    that an error against the oracle is the kernels', not the network's.
    """
    hp = resolve_hparams(hp)
    if style == "outlier":
        return _outlier_state_dict(hp, seed)
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    hd = hp["head_dim"]
    for key, shape in state_dict_shapes(hp).items():
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("rotary_embed.freqs"):
            v = (1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))).astype(np.float32)
        elif leaf == "num_batches_tracked":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        elif style == "init0" and (len(shape) == 1 or leaf in ("running_mean", "running_var")):
            v = np.ones(shape) if leaf in ("gamma", "weight", "running_var") else np.zeros(shape)
        elif leaf == "running_mean":
            v = 0.5 * rng.standard_normal(shape)
            if key.startswith("frontend.stem.bn1d"):
                v = 3.0 + v  # log-mel values live around 3..5
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
            if key.startswith("frontend.stem.bn1d"):
                v = v * 2.0
        elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf == "bias":
            v = 0.1 * rng.standard_normal(shape)
        elif len(shape) == 4:  # conv, Kaiming normal fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt(2.0 / fan_out) if style.startswith("init") else 1.3 / math.sqrt(fan_in)
            v = std * rng.standard_normal(shape)
        else:  # linear
            if style.startswith("init"):
                std = 0.02
            elif "to_qkv" in key:
                std = 1.6 / math.sqrt(shape[1])
            elif "task_heads" in key:
                std = 2.0 / math.sqrt(shape[1])
            else:
                std = 1.0 / math.sqrt(shape[1])
            v = std * rng.standard_normal(shape)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return sd


OUTLIER_GAIN = 1000.0   # residual-stream magnitude of the outlier channels of style="outlier" (ordinary channels: O(1 .. 10))


def _outlier_state_dict(hp: dict, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """style="outlier" of ``random_state_dict`` (see there)."""
    sd = random_state_dict(hp, seed=seed, style="lively")
    rng = np.random.default_rng(seed + 7919)
    D, L = hp["transformer_dim"], hp["n_layers"]
    # heavy-tailed matrix entries: every 2-D weight of the transformer parts times |t_4| / sqrt(E t_4^2) element by element
    # (same RMS -- a larger overall gain makes the network chaotic: fp32 and fp64 CPU forwards then differ by 0.1 at the
    # logits and no 1e-3 gate means anything -- with occasional entries x4 .. x8)
    for key, v in sd.items():
        if v.dim() == 2 and "task_heads" not in key:
            t = np.abs(rng.standard_t(4, size=tuple(v.shape))).astype(np.float32)
            sd[key] = v * torch.from_numpy(t / np.float32(np.sqrt(2.0)))
    # sharper attention: q and k rows of every to_qkv x 1.25 (scores x 1.56)
    for key, v in sd.items():
        if key.endswith("to_qkv.weight"):
            v[: 2 * v.shape[1]] *= 1.25
    # outlier channels of the main residual stream: a large, nearly token-independent component (what "massive activations"
    # of trained transformers look like) that the layers keep feeding, and RMSNorm gammas that undo what it does to the norm
    n_out = max(2, D // 128)
    out_ch = rng.choice(D, size=n_out, replace=False)
    gain = OUTLIER_GAIN
    sd["frontend.linear.weight"][out_ch] *= 4.0
    sd["frontend.linear.bias"][out_ch] = torch.from_numpy((gain * rng.choice([-1.0, 1.0], n_out)).astype(np.float32))
    comp = gain * float(np.sqrt(n_out / D)) / 4     # ~ (row norm with outliers) / (row norm without)
    gammas = ["transformer_blocks.norm.gamma"]
    for l in range(L):
        p = f"transformer_blocks.layers.{l}."
        sd[p + "0.to_out.0.weight"][out_ch] *= 4.0
        sd[p + "1.net.4.weight"][out_ch] *= 4.0
        gammas += [p + "0.norm.gamma", p + "1.net.0.gamma"]
    for g in gammas:
        sd[g] *= comp
        sd[g][out_ch] *= 1.0 / (comp * gain / 8)
    # frontend: two BatchNorm channels per block with a large scale (activations of ~10^2 inside the partial transformers)
    for i in range(3):
        w = sd[f"frontend.blocks.{i}.norm.weight"]
        w[rng.choice(w.shape[0], size=2, replace=False)] *= 6.0
    return sd


def synthetic_audio(seconds: float, seed: int = 0, sr: int = 22050) -> np.ndarray:
    """White noise N(0,0.1^2) + 120 BPM click train (every 4th click doubled) -- SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sr))
    x = rng.normal(0.0, 0.1, n).astype(np.float32)
    step = sr // 2
    for i, pos in enumerate(range(0, n, step)):
        x[pos] += 2.0 if i % 4 == 0 else 1.0
    return x


def synthetic_spect(n_frames: int, seed: int = 0) -> np.ndarray:
    """Cheap seeded stand-in for a log-mel spectrogram: smooth-ish values in [0, 7]."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(0.0, 1.0, (n_frames, 128)).astype(np.float32)
    env = (1.0 + np.sin(np.arange(n_frames, dtype=np.float32)[:, None] * (2 * np.pi / 25.0))) * 0.5
    tilt = np.linspace(1.0, 0.3, 128, dtype=np.float32)[None, :]
    return np.log1p(1000.0 * 0.05 * base * (0.2 + env) * tilt).astype(np.float32)

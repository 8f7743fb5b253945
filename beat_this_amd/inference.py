"""Inference API -- drop-in for beat_this/inference.py (same names, signatures, return types,
exceptions), backed by the HIP kernels of libbeat_this_amd.so.

Differences that matter for speed, not for results:
  * ``split_predict_aggregate`` runs ALL chunks of a piece as one batch (the reference loops
    batch-1, inference.py:215); chunk gathering and keep_first aggregation are GPU kernels.
  * ``Spect2Frames.spect2frames_many`` (extension) batches the chunks of several pieces, and
    shards them over torch.distributed ranks (RCCL) when a process group is initialised.
"""
from __future__ import annotations

import inspect

import numpy as np
import torch

from . import _lib
from .model import BeatThis
from .postprocessor import Postprocessor
from .preprocessing import LogMelSpect, load_audio
from .utils import replace_state_dict_key, save_beat_tsv

CHECKPOINT_URL = "https://cloud.cp.jku.at/public.php/dav/files/7ik4RrBKTS273gp"
MAX_CHUNKS_PER_LAUNCH = 64  # workspace is ~70 MB (fp32) per chunk; larger batches are split


def load_checkpoint(checkpoint_path, device="cpu") -> dict:
    """Local file, short name or URL -> checkpoint dict (inference.py:16-53)."""
    try:
        return torch.load(checkpoint_path, map_location=device, weights_only=True)
    except FileNotFoundError:
        try:
            if str(checkpoint_path).startswith(("https://", "http://")):
                url, file_name = checkpoint_path, None
            else:
                url, file_name = f"{CHECKPOINT_URL}/{checkpoint_path}.ckpt", f"beat_this-{checkpoint_path}.ckpt"
            return torch.hub.load_state_dict_from_url(url, file_name=file_name, map_location=device)
        except Exception:
            raise ValueError("Could not load the checkpoint given the provided name", checkpoint_path)


def load_model(checkpoint_path="final0", device="cpu") -> BeatThis:
    """BeatThis in eval mode on ``device`` (inference.py:56-87).  ``checkpoint_path`` may also be
    an already loaded checkpoint dict, or None for a randomly initialised model."""
    if checkpoint_path is not None:
        ckpt = checkpoint_path if isinstance(checkpoint_path, dict) else load_checkpoint(checkpoint_path, "cpu")
        accepted = set(inspect.signature(BeatThis).parameters)
        hparams = {k: v for k, v in ckpt["hyper_parameters"].items() if k in accepted}
        model = BeatThis(**hparams)
        model.load_state_dict(replace_state_dict_key(dict(ckpt["state_dict"]), "model.", ""))
    else:
        model = BeatThis()
    return model.to(device).eval()


def zeropad(spect: torch.Tensor, left: int = 0, right: int = 0):
    if left == 0 and right == 0:
        return spect
    return torch.nn.functional.pad(spect, (0, 0, left, right), "constant", 0)


def chunk_starts(n_frames: int, chunk_size: int, border_size: int = 6, avoid_short_end: bool = True) -> np.ndarray:
    """Start frames of split_piece (inference.py:120-125)."""
    starts = np.arange(-border_size, n_frames - border_size, chunk_size - 2 * border_size)
    if avoid_short_end and n_frames > chunk_size - 2 * border_size:
        starts[-1] = n_frames - (chunk_size - border_size)
    return starts


def chunk_length(n_frames: int, chunk_size: int, border_size: int) -> int:
    """All chunks of a piece have this many frames (short pieces give one shorter chunk)."""
    return chunk_size if n_frames > chunk_size - 2 * border_size else n_frames + 2 * border_size


def split_piece(spect: torch.Tensor, chunk_size: int, border_size: int = 6, avoid_short_end: bool = True):
    """List of zero-padded chunks + their starts (inference.py:100-135); torch glue for callers
    that want the chunks themselves -- the fast path below gathers them on the GPU instead."""
    n = len(spect)
    starts = chunk_starts(n, chunk_size, border_size, avoid_short_end)
    chunks = [zeropad(spect[max(s, 0): min(s + chunk_size, n)], left=max(0, -s),
                      right=max(0, min(border_size, s + chunk_size - n))) for s in map(int, starts)]
    return chunks, starts


def aggregate_prediction(pred_chunks, starts, full_size, chunk_size, border_size, overlap_mode, device):
    """Reference-compatible aggregation of per-chunk dicts (inference.py:138-185)."""
    if border_size > 0:
        pred_chunks = [{k: p[k][border_size:-border_size] for k in ("beat", "downbeat")} for p in pred_chunks]
    beat = torch.full((full_size,), -1000.0, device=device)
    down = torch.full((full_size,), -1000.0, device=device)
    order = list(zip(starts, pred_chunks))
    if overlap_mode == "keep_first":
        order.reverse()
    for start, p in order:
        beat[start + border_size: start + chunk_size - border_size] = p["beat"]
        down[start + border_size: start + chunk_size - border_size] = p["downbeat"]
    return beat, down


def _gather_chunks(spect: torch.Tensor, starts: np.ndarray, T: int):
    B = len(starts)
    d_starts = torch.as_tensor(np.asarray(starts, dtype=np.int32), device=spect.device)
    chunks = torch.empty((B, T, 128), dtype=torch.float32, device=spect.device)
    with torch.cuda.device(spect.device):
        _lib.check(_lib.lib().bt_split_chunks(_lib.stream_ptr(spect.device), spect.data_ptr(), spect.shape[0],
                                              d_starts.data_ptr(), B, T, chunks.data_ptr()))
    return chunks, d_starts


def _run_batched(model, chunks: torch.Tensor):
    """model over (B,T,128) in slices of MAX_CHUNKS_PER_LAUNCH -> beat, downbeat (B,T)."""
    outs = [model(chunks[i: i + MAX_CHUNKS_PER_LAUNCH]) for i in range(0, chunks.shape[0], MAX_CHUNKS_PER_LAUNCH)]
    if len(outs) == 1:
        return outs[0]["beat"], outs[0]["downbeat"]
    return torch.cat([o["beat"] for o in outs]), torch.cat([o["downbeat"] for o in outs])


def split_predict_aggregate(spect: torch.Tensor, chunk_size: int, border_size: int, overlap_mode: str,
                            model) -> dict:
    """Chunk a (T,128) piece, predict, aggregate (inference.py:188-230).  ``model`` is any callable
    (B,T,128) -> {"beat": (B,T), "downbeat": (B,T)}; all chunks go through it as ONE batch."""
    _lib.require_gpu(spect, "spectrogram")
    if spect.dim() != 2 or spect.shape[1] != 128:
        raise ValueError(f"expected a (frames, 128) spectrogram, got {tuple(spect.shape)}")
    spect = spect.to(torch.float32).contiguous()
    n = spect.shape[0]
    starts = chunk_starts(n, chunk_size, border_size)
    T = chunk_length(n, chunk_size, border_size)
    chunks, d_starts = _gather_chunks(spect, starts, T)
    cb, cd = _run_batched(model, chunks)
    cb, cd = cb.float().contiguous(), cd.float().contiguous()
    if overlap_mode != "keep_first":  # "keep_last": reference-compatible torch glue
        preds = [{"beat": cb[i], "downbeat": cd[i]} for i in range(len(starts))]
        beat, down = aggregate_prediction(preds, starts, n, chunk_size, border_size, overlap_mode, spect.device)
        return {"beat": beat, "downbeat": down}
    beat = torch.empty((n,), dtype=torch.float32, device=spect.device)
    down = torch.empty((n,), dtype=torch.float32, device=spect.device)
    with torch.cuda.device(spect.device):
        _lib.check(_lib.lib().bt_aggregate(_lib.stream_ptr(spect.device), cb.data_ptr(), cd.data_ptr(),
                                           d_starts.data_ptr(), len(starts), T, border_size, n, beat.data_ptr(),
                                           down.data_ptr()))
    return {"beat": beat, "downbeat": down}


class Spect2Frames:
    """Framewise beat/downbeat logits from a spectrogram (inference.py:233-257)."""

    def __init__(self, checkpoint_path="final0", device="cpu", float16=False):
        super().__init__()
        self.device = torch.device(device)
        self.float16 = bool(float16)
        self.model = load_model(checkpoint_path, self.device)
        if float16 == "fp8":  # extension: float16="fp8" -> autocast + e4m3 feed-forward GEMMs (BT_PREC_FP8)
            self.model.fp8_weights = True

    def spect2frames(self, spect):
        with torch.inference_mode():
            with torch.autocast(enabled=self.float16, device_type=self.device.type):
                pred = split_predict_aggregate(spect=spect, chunk_size=1500, overlap_mode="keep_first",
                                               border_size=6, model=self.model)
        return pred["beat"].float(), pred["downbeat"].float()

    def spect2frames_many(self, spects, group=None):
        """Extension: several (T_i,128) pieces at once.  Chunks of all pieces form one batch; with an
        initialised torch.distributed group the batch is block-partitioned over the ranks, every
        rank computes its share, and ONE all_gather (RCCL over xGMI) returns all chunk logits to
        every rank.  Returns a list of (beat, downbeat) like repeated ``spect2frames`` calls."""
        from .parallel import forward_chunks_sharded

        with torch.inference_mode():
            with torch.autocast(enabled=self.float16, device_type=self.device.type):
                return forward_chunks_sharded(self.model, spects, 1500, 6, group)

    def __call__(self, spect):
        return self.spect2frames(spect)


class Audio2Frames(Spect2Frames):
    """Framewise logits from an audio signal (inference.py:260-281)."""

    def __init__(self, checkpoint_path="final0", device="cpu", float16=False):
        super().__init__(checkpoint_path, device, float16)
        self.spect = LogMelSpect(device=self.device)

    def signal2spect(self, signal, sr):
        if signal.ndim == 2:
            signal = signal.mean(1)
        elif signal.ndim != 1:
            raise ValueError(f"Expected 1D or 2D signal, got shape {signal.shape}")
        signal = torch.tensor(signal, dtype=torch.float32, device=self.device)
        if sr != 22050:
            signal = resample_gpu(signal, sr, 22050)
        return self.spect(signal)

    def __call__(self, signal, sr):
        return self.spect2frames(self.signal2spect(signal, sr))


class Audio2Beats(Audio2Frames):
    """Beat and downbeat times in seconds from an audio signal (inference.py:284-303)."""

    def __init__(self, checkpoint_path="final0", device="cpu", float16=False, dbn=False):
        super().__init__(checkpoint_path, device, float16)
        self.frames2beats = Postprocessor(type="dbn" if dbn else "minimal")

    def __call__(self, signal, sr):
        beat_logits, downbeat_logits = super().__call__(signal, sr)
        return self.frames2beats(beat_logits, downbeat_logits)


class File2Beats(Audio2Beats):
    def __call__(self, audio_path):
        signal, sr = load_audio(audio_path)
        return super().__call__(signal, sr)


class File2File(File2Beats):
    def __call__(self, audio_path, output_path):
        downbeats, beats = super().__call__(audio_path)  # (sic) argument naming as in inference.py:313-315
        save_beat_tsv(downbeats, beats, output_path)


_RESAMPLE_FILTERS: dict = {}


def resample_gpu(signal: torch.Tensor, in_rate: int, out_rate: int) -> torch.Tensor:
    """1-D fp32 device waveform at ``in_rate`` -> ``out_rate`` on the GPU (csrc/frontend.hip: resample_kernel), the
    MI355X replacement of the reference's host ``soxr.resample`` (inference.py:274-275).  Polyphase
    Kaiser-windowed-sinc FIR with scipy.signal.resample_poly's filter design; like every non-libsoxr resampler it is
    not bit-compatible with soxr: parity is defined from the 22.05 kHz waveform onwards."""
    import ctypes as C
    from math import gcd

    from . import tables

    _lib.require_gpu(signal, "waveform")
    if signal.dim() != 1:
        raise ValueError(f"expected a 1-D waveform, got shape {tuple(signal.shape)}")
    in_rate, out_rate = int(in_rate), int(out_rate)
    g = gcd(in_rate, out_rate)
    up, down = out_rate // g, in_rate // g
    if up == down:
        return signal
    key = (up, down, signal.device)
    if key not in _RESAMPLE_FILTERS:
        h, half = tables.resample_filter(up, down)
        _RESAMPLE_FILTERS[key] = (torch.from_numpy(h.astype(np.float32)).to(signal.device), half)
    h, half = _RESAMPLE_FILTERS[key]
    x = signal.to(torch.float32).contiguous()
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    y = torch.empty(n_out, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().bt_resample(_lib.stream_ptr(x.device), x.data_ptr(), n_in, up, down, h.data_ptr(), half,
                                          y.data_ptr(), n_out))
    return y


def resample(signal: np.ndarray, in_rate: int, out_rate: int) -> np.ndarray:
    """Host resampler for sr != 22050 (reference: soxr.resample, inference.py:274-275).
    soxr (libsoxr) if installed, else a polyphase FIR (scipy).  Not bit-compatible with each
    other: parity is defined from the 22.05 kHz waveform onwards."""
    try:
        import soxr

        return soxr.resample(signal, in_rate=in_rate, out_rate=out_rate)
    except ImportError:
        from math import gcd

        from scipy.signal import resample_poly

        g = gcd(int(in_rate), int(out_rate))
        return resample_poly(np.asarray(signal), int(out_rate) // g, int(in_rate) // g, axis=0)

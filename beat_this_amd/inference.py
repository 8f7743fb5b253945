"""Inference API -- drop-in for beat_this/inference.py (same names, signatures, return types,
exceptions), backed by the HIP kernels of libbeat_this_amd.so.

Differences that matter for speed, not for results:
  * ``split_predict_aggregate`` runs ALL chunks of a piece as one batch (the reference loops
    batch-1, inference.py:215); chunk gathering and keep_first aggregation are GPU kernels.
  * ``Spect2Frames.spect2frames_many`` (extension) batches the chunks of several pieces, and
    shards them over torch.distributed ranks (RCCL) when a process group is initialised.
  * ``Audio2Beats.many`` / ``Audio2Frames.signal2spect_many`` / ``Spect2Frames.spect2frames_batch`` (extensions):
    a whole list of tracks per call -- ONE launch per stage (resample, log-mel, chunk gather, forward slices,
    aggregation, peak picking) and ONE device-to-host copy for all tracks.

Precision (the ``float16`` argument of the inference classes):
  * ``False`` (the reference's default): fp32 activations and fp32-class results -- framewise logits within 1e-3 of the
    reference's CPU fp32 path (measured 1e-5 .. 2e-5), identical beats -- with every product of the forward on three fp16
    MFMAs over hi + lo operand halves (BT_PREC_F32X3; ``BeatThis.fp32_split_gemms``).  A batch whose operands leave the
    fp16 range of a hi half is detected and repeated on the exact fp32 MFMA path (``Engine.last_fallbacks``).
  * ``True``: the reference's float16 autocast -- fp16 MFMA operands, fp32 accumulation; not under the 1e-3 gate.
  * ``"exact"``: every product on exact fp32 MFMAs (``v_mfma_f32_32x32x2_f32``), a third of the default's speed.

Limits of the drop-in (there is no CPU implementation in this package):
  * ``device`` must be a ROCm GPU; the default is "cuda" (the reference's default "cpu" raises here, in ``__init__``);
  * the resampler is a Kaiser-windowed-sinc polyphase FIR built to libsoxr's HQ specification, not libsoxr.
"""
from __future__ import annotations

import inspect

import numpy as np
import torch

from . import _lib
from .model import BeatThis
from .pack import Engine
from .postprocessor import Postprocessor
from .preprocessing import LogMelSpect, load_audio
from .utils import replace_state_dict_key, save_beat_tsv

CHECKPOINT_URL = "https://cloud.cp.jku.at/public.php/dav/files/7ik4RrBKTS273gp"
MAX_CHUNKS_PER_LAUNCH = 96  # workspace is ~70 MB (fp32) / ~45 MB (half) per chunk; larger batches are split into equal slices
CONCURRENT_STREAMS = 2      # forward slices of a big batch run on this many streams (1 = everything on the caller's stream)
CONCURRENT_SLICE_CHUNKS = 32  # ... from this many chunks on
_SIDE_STREAMS: dict = {}


def _side_streams(dev, n):
    key = (torch.device(dev).index, n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(dev) for _ in range(n)]
    return _SIDE_STREAMS[key]


def _copy_stream(dev):
    key = (torch.device(dev).index, "copy")
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


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


def _gpu_device(device) -> torch.device:
    """The device argument of the inference classes: a ROCm GPU, or a clear error right away."""
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"beat_this_amd runs on ROCm GPUs only (got device='{dev}'): this package has no CPU implementation. "
            "Pass device='cuda' (the default here) or 'cuda:N'.")
    return dev


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


USE_GRAPHS = True   # single-file path: the forward of up to Engine.GRAPH_MAX_CHUNKS chunks is replayed as one hipGraph
USE_ONE_CALL = True  # Audio2Beats.__call__: one track = ONE library call (bt_audio2beats_enqueue), no Python between the stages
ONE_CALL_MAX_CHUNKS = 96   # longer tracks (> 47 minutes) take the sliced path


def _graphed_forward(model, spect: torch.Tensor, starts: np.ndarray, T: int):
    """The chunks of one piece through a captured forward (pack.Engine.graph_forward): the chunk gather writes straight into
    the graph's input buffer and the aggregation reads the graph's logits -- no launches from the host but three, no copies.
    -> (beat, downbeat, device starts), or None when the model is not a BeatThis on its fused path, the piece is too long,
    graphs are off, or the range guard of BT_PREC_F32X3 fired (the caller then takes the ordinary path, fallback included)."""
    if not USE_GRAPHS or not isinstance(model, BeatThis) or len(starts) > Engine.GRAPH_MAX_CHUNKS or _model_hooked(model):
        return None
    eng = model.engine()
    prec = model._precision()
    entry = eng.graph_forward(len(starts), T, prec)
    if entry is None:
        return None
    dev = spect.device
    with torch.cuda.device(dev):
        d_starts = _lib.upload(np.asarray(starts, dtype=np.int32), dev)
        _lib.check(_lib.lib().bt_split_chunks(_lib.stream_ptr(dev), spect.data_ptr(), spect.shape[0], d_starts.data_ptr(), len(starts), T,
                                              entry.x.data_ptr()))
        entry.replay()
        if prec == _lib.PREC_F32X3 and int(entry.flag.item()) != 0:
            return None   # (operands beyond the fp16 range: the ordinary path repeats the piece and counts the fallback)
    return entry.beat, entry.down, d_starts


def _model_hooked(model) -> bool:
    from .model import _hooked_below

    return _hooked_below(model) or bool(model._forward_hooks or model._forward_pre_hooks)


def _run_batched(model, chunks: torch.Tensor):
    """model over (B,T,128) in equal slices of at most MAX_CHUNKS_PER_LAUNCH -> beat, downbeat (B,T)."""
    B = chunks.shape[0]
    step = -(-B // max(1, -(-B // MAX_CHUNKS_PER_LAUNCH)))
    outs = [model(chunks[i: i + step]) for i in range(0, B, step)]
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
    if n == 0:  # no chunks: the reference returns empty (full_size = 0) tensors, inference.py:172-173
        empty = torch.empty((0,), dtype=torch.float32, device=spect.device)
        return {"beat": empty, "downbeat": empty.clone()}
    starts = chunk_starts(n, chunk_size, border_size)
    T = chunk_length(n, chunk_size, border_size)
    fast = _graphed_forward(model, spect, starts, T) if overlap_mode == "keep_first" else None
    if fast is not None:
        cb, cd, d_starts = fast
    else:
        chunks, d_starts = _gather_chunks(spect, starts, T)
        cb, cd = _run_batched(model, chunks)
        cb, cd = cb.float().contiguous(), cd.float().contiguous()
    if overlap_mode != "keep_first":  # "keep_last": reference-compatible torch glue
        preds = [{"beat": cb[i], "downbeat": cd[i]} for i in range(len(starts))]
        beat, down = aggregate_prediction(preds, starts, n, chunk_size, border_size, overlap_mode, spect.device)
        return {"beat": beat, "downbeat": down}
    both = torch.empty((2, n), dtype=torch.float32, device=spect.device)   # (adjacent rows: the peak picker takes them as they lie)
    beat, down = both[0], both[1]
    with torch.cuda.device(spect.device):
        _lib.check(_lib.lib().bt_aggregate(_lib.stream_ptr(spect.device), cb.data_ptr(), cd.data_ptr(),
                                           d_starts.data_ptr(), len(starts), T, border_size, n, beat.data_ptr(),
                                           down.data_ptr()))
    return {"beat": beat, "downbeat": down}


def _precision_mode(float16) -> str:
    """The ``float16`` argument of the inference classes -> "half" | "f32x3" | "exact" (module docstring)."""
    if isinstance(float16, str):
        mode = {"f32x3": "f32x3", "exact": "exact", "fp32": "exact", "f32": "exact", "half": "half", "fp16": "half",
                "float16": "half"}.get(float16.lower())
        if mode is None:
            raise ValueError(f"unknown precision float16={float16!r}: use False (fp32-class results, the default), True "
                             "(fp16 autocast) or 'exact' (exact fp32 MFMAs)")
        return mode
    return "half" if float16 else "f32x3"


class Spect2Frames:
    """Framewise beat/downbeat logits from a spectrogram (inference.py:233-257)."""

    def __init__(self, checkpoint_path="final0", device="cuda", float16=False):
        super().__init__()
        self.device = _gpu_device(device)
        mode = _precision_mode(float16)
        self.float16 = mode == "half"
        # what ``float16=False`` means here: hi + lo fp16 operands (BT_PREC_F32X3, the default of BeatThis itself) or exact fp32
        # MFMAs ("exact").  ``model.fp32_split_gemms`` is the switch; a model assigned to ``.model`` keeps the value its owner
        # gave it (a model shared by two objects is not flipped by the second one) -- except that an object created with
        # float16="exact" turns it off on the models it is given, since that is what it was asked for.
        self.fp32_mode = "exact" if mode == "exact" or _lib.lib().bt_half_is_bf16() else "f32x3"
        self.model = load_model(checkpoint_path, self.device)

    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, m):
        if isinstance(m, BeatThis) and self.fp32_mode == "exact":
            m.fp32_split_gemms = False
        self._model = m

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

    def spect2frames_batch(self, spect: torch.Tensor, frame_off, checks=None):
        """Extension: the pieces ``spect[frame_off[k]:frame_off[k+1]]`` of one concatenated (frames, 128) spectrogram
        -> (beat, downbeat) logits concatenated the same way.  One chunk-gather launch per forward slice, one aggregation
        launch for all pieces (same chunking / keep_first arithmetic as ``spect2frames``, inference.py:188-230).
        ``checks``: see ``batch_predict_aggregate``."""
        with torch.inference_mode():
            with torch.autocast(enabled=self.float16, device_type=self.device.type):
                return batch_predict_aggregate(spect, frame_off, 1500, 6, self.model, checks=checks)

    def __call__(self, spect):
        return self.spect2frames(spect)


class Audio2Frames(Spect2Frames):
    """Framewise logits from an audio signal (inference.py:260-281)."""

    def __init__(self, checkpoint_path="final0", device="cuda", float16=False):
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

    def signal2spect_many(self, signals, sr):
        """Extension: a list of waveforms (numpy (N,) / (N, C) or torch tensors on the host or the device, all at sample rate ``sr``) ->
        (spect, frame_off): ONE (sum of frames, 128) spectrogram tensor with the tracks back to back and the int64 numpy
        array of their first rows (length n + 1).  Mono mix as in ``signal2spect``; resampling and the log-mel run as one
        launch each for all tracks (bt_resample_batch / bt_logmel_batch)."""
        import ctypes as C
        from math import gcd

        dev = self.device
        for sig in signals:
            if sig.ndim not in (1, 2):
                raise ValueError(f"Expected 1D or 2D signal, got shape {tuple(sig.shape)}")
        # Host buffers: PINNED tensors go up on a copy stream of their own (the caller's stream waits for one event), so the
        # upload of batch i + 1 overlaps the kernels of batch i when batches are submitted ahead (many_async); anything else
        # is copied the ordinary, synchronous way.
        cur = torch.cuda.current_stream(dev)
        pinned = [isinstance(sig, torch.Tensor) and sig.device.type == "cpu" and sig.is_pinned() for sig in signals]
        waves = [None] * len(signals)
        if any(pinned):
            copy = _copy_stream(dev)
            with torch.cuda.stream(copy):
                for k, sig in enumerate(signals):
                    if pinned[k]:
                        waves[k] = sig.to(dev, non_blocking=True)
                        waves[k].record_stream(cur)
            ev = torch.cuda.Event()
            ev.record(copy)
            cur.wait_event(ev)
        for k, sig in enumerate(signals):
            w = waves[k]
            if w is None:
                w = sig.to(dev) if isinstance(sig, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sig)).to(dev)
            if w.dim() == 2:   # mono mix in the signal's own type (float64 for float64 / integer input), like numpy's mean(1)
                w = (w if w.dtype in (torch.float32, torch.float64) else w.to(torch.float64)).mean(1)   # in the reference (inference.py:270-271)
            waves[k] = w.to(torch.float32).contiguous()
        n = len(waves)
        if n == 0:
            return torch.empty((0, 128), dtype=torch.float32, device=dev), np.zeros(1, dtype=np.int64)
        sr = int(sr)
        g = gcd(sr, 22050)
        up, down = 22050 // g, sr // g
        n_in = np.array([w.shape[0] for w in waves], dtype=np.int64)
        n22 = n_in if up == down else -(-n_in * up // down)
        if int(n22.min()) <= 512:
            raise ValueError("signal too short: reflect padding needs more than 512 samples")
        n_fr = 1 + n22 // 441
        s_off = np.concatenate([[0], np.cumsum(n22)]).astype(np.int64)
        frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int64)
        lib, st = _lib.lib(), _lib.stream_ptr(dev)
        table = np.zeros((2, n, 4), dtype=np.int64)
        with torch.cuda.device(dev):
            if up != down:
                buf22 = torch.empty(int(s_off[-1]), dtype=torch.float32, device=dev)
                table[0, :, 0] = [w.data_ptr() for w in waves]
                table[0, :, 1], table[0, :, 2], table[0, :, 3] = n_in, s_off[:-1], n22
                table[1, :, 0] = buf22.data_ptr() + 4 * s_off[:-1]
            else:
                table[1, :, 0] = [w.data_ptr() for w in waves]
            table[1, :, 1], table[1, :, 2], table[1, :, 3] = n22, frame_off[:-1], n_fr
            d_table = _lib.upload(table, dev)
            if up != down:
                h, half = _resample_filter(up, down, dev)
                _lib.check(lib.bt_resample_batch(st, d_table[0].data_ptr(), n, int(n22.max()), up, down, h.data_ptr(), half,
                                                 buf22.data_ptr()))
            spect = torch.empty((int(frame_off[-1]), 128), dtype=torch.float32, device=dev)
            if self.spect.device != dev:
                self.spect.to(dev)
            _lib.check(lib.bt_logmel_batch(st, C.byref(self.spect._get_tables()), d_table[1].data_ptr(), n, int(n_fr.max()),
                                           spect.data_ptr()))
        return spect, frame_off

    def many(self, signals, sr):
        """Extension: [(beat_logits, downbeat_logits)] for a list of waveforms, batched per stage."""
        spect, frame_off = self.signal2spect_many(signals, sr)
        beat, down = self.spect2frames_batch(spect, frame_off)
        return [(beat[frame_off[k]: frame_off[k + 1]], down[frame_off[k]: frame_off[k + 1]]) for k in range(len(signals))]

    def __call__(self, signal, sr):
        return self.spect2frames(self.signal2spect(signal, sr))


class Audio2Beats(Audio2Frames):
    """Beat and downbeat times in seconds from an audio signal (inference.py:284-303)."""

    def __init__(self, checkpoint_path="final0", device="cuda", float16=False, dbn=False):
        super().__init__(checkpoint_path, device, float16)
        self.frames2beats = Postprocessor(type="dbn" if dbn else "minimal")

    def __call__(self, signal, sr):
        out = self._one_call(signal, sr) if USE_ONE_CALL else None
        if out is not None:
            return out
        beat_logits, downbeat_logits = super().__call__(signal, sr)
        return self.frames2beats(beat_logits, downbeat_logits)

    def _one_call(self, signal, sr, exact=False):
        """``__call__`` as ONE library call (bt_audio2beats_enqueue, include/beat_this_amd.h): the mono mix and the upload of the
        waveform happen here like in ``signal2spect`` (inference.py:269-277), then resampler, log-mel, chunk gather, the forward
        (a hipGraph the library captures itself), aggregation, peak picking and the device-to-host copy of the peak frames are
        enqueued by the library in one go -- round 5's path came back to Python five times per file, and the GPU waited each time
        (profiles/r05_latency_trace.txt).  Same kernels in the same order: bit-identical to the stage-by-stage path (tested).
        -> (beats, downbeats), or None where the call does not apply (DBN post-processing, a model that is not a plain BeatThis,
        hooks, a replaced ``self.spect``, very long tracks) and the caller takes the ordinary path."""
        import ctypes as C
        from math import gcd

        model = self.model
        if (self.frames2beats.type != "minimal" or not isinstance(model, BeatThis) or _model_hooked(model)
                or type(self.spect) is not LogMelSpect or model.device != self.device):
            return None
        signal = np.asarray(signal) if not isinstance(signal, torch.Tensor) else signal
        if signal.ndim == 2:
            signal = signal.mean(1)
        elif signal.ndim != 1:
            raise ValueError(f"Expected 1D or 2D signal, got shape {signal.shape}")
        dev = self.device
        sr = int(sr)
        g = gcd(sr, 22050)
        up, down = 22050 // g, sr // g
        eng = model.engine()
        if eng._h_prof_on() or eng._deferred is not None:
            return None
        with torch.inference_mode(), torch.autocast(enabled=self.float16, device_type=dev.type):
            prec = _lib.PREC_F32 if exact else model._precision()
        lib = _lib.lib()
        plan = _lib.A2BPlan()
        n_in = int(signal.shape[0])
        if n_in == 0:
            return None
        _lib.check(lib.bt_audio2beats_plan(eng._h, n_in, up, down, prec, C.byref(plan)))
        if plan.B > ONE_CALL_MAX_CHUNKS:
            return None
        eng.ensure_positions(plan.T)
        with torch.cuda.device(dev):
            # (from_numpy + one H2D copy: torch.tensor(array, device=...) copies the 5 MB of a 30 s file on the host first)
            wave = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32)).to(dev) if not isinstance(signal, torch.Tensor) \
                else signal.to(dev, torch.float32).contiguous()
            if self.spect.device != dev:
                self.spect.to(dev)
            h, half = _resample_filter(up, down, dev) if up != down else (None, 0)
            ws = eng._a2b_workspace(plan.ws_bytes)
            host = self.frames2beats._pinned(plan.result_words)
            _lib.check(lib.bt_audio2beats_enqueue(eng._h, _lib.stream_ptr(dev), prec, C.byref(self.spect._get_tables()),
                                                  wave.data_ptr(), n_in, up, down, _lib.ptr(h), half, ws.data_ptr(), ws.numel(),
                                                  host.data_ptr(), int(USE_GRAPHS and plan.T == Engine.GRAPH_T)))
            torch.cuda.current_stream(dev).synchronize()
        res = host.numpy()
        n = int(plan.n_frames)
        nb, nd, flag = int(res[2 * n]), int(res[2 * n + 1]), int(res[2 * n + 2])
        out = None
        if flag == 0:
            from .postprocessor import _host_post

            out = _host_post(res[:nb], res[n: n + nd], self.frames2beats.fps)
        self.frames2beats.__dict__.setdefault("_pin_pool", []).append(host)
        if flag != 0:   # BT_PREC_F32X3: an operand left the fp16 range of a hi part -- the exact fp32 path repeats the call
            eng.last_fallbacks += 1
            return self._one_call(signal, sr, exact=True)
        return out

    def many(self, signals, sr):
        """Extension: [(beats, downbeats)] (seconds, float64 arrays as ``__call__`` returns them) for a list of waveforms
        at sample rate ``sr``: every GPU stage is one launch for all tracks and the peak indices of all tracks come back in
        one device-to-host copy (``Postprocessor.ragged``)."""
        return self.many_async(signals, sr).result()

    def many_async(self, signals, sr):
        """``many`` without the final wait: all GPU work and the device-to-host copy of the peak indices are enqueued;
        ``.result()`` of the returned handle waits for the copy and runs the host step.  Submitting batch i + 1 before
        collecting batch i keeps the GPU busy during the host step (bench.py does)."""
        spect, frame_off = self.signal2spect_many(signals, sr)
        # float16="f32x3": the range flags of the forward slices are collected, not waited for; ``result()`` looks at them
        # once the batch's device-to-host copy has arrived and repeats the batch on the exact fp32 path if one fired
        if self.frames2beats.type != "minimal":
            # the DBN (madmom) runs on the host right away and needs final logits: the range flags are looked at -- and the
            # batch repeated on the exact path if one fired -- before it sees them (checks=None: the synchronous guard)
            beat, down = self.spect2frames_batch(spect, frame_off)

            class _Done:
                def __init__(s, out): s.out, s.logits = out, (beat, down, frame_off)
                def result(s): return s.out
            return _Done([self.frames2beats(beat[frame_off[k]: frame_off[k + 1]], down[frame_off[k]: frame_off[k + 1]])
                          for k in range(len(signals))])
        checks = [] if isinstance(self.model, BeatThis) and self.model.fp32_split_gemms and not self.float16 else None
        beat, down = self.spect2frames_batch(spect, frame_off, checks=checks)
        pending = self.frames2beats.ragged_async(beat, down, frame_off)
        pending.logits = (beat, down, frame_off)   # framewise logits of the batch (concatenated), for callers that want them
        if checks:
            return _GuardedPending(pending, checks, lambda: self._many_exact_async(signals, sr))
        return pending

    def _many_exact_async(self, signals, sr):
        """``many_async`` on the exact fp32 MFMA path (the repeat of a BT_PREC_F32X3 batch whose range flag fired)."""
        old = self.model.fp32_split_gemms
        self.model.fp32_split_gemms = False
        try:
            return self.many_async(signals, sr)
        finally:
            self.model.fp32_split_gemms = old


class _GuardedPending:
    """A pending ``many_async`` result of the BT_PREC_F32X3 path: ``result()`` also evaluates the forward slices' range
    flags (they arrived with / before the peak indices) and repeats the batch on the exact path when one fired."""

    def __init__(self, inner, checks, redo):
        self.inner, self.checks, self.redo = inner, checks, redo
        self.logits = inner.logits

    def result(self):
        out = self.inner.result()
        if self.checks and any(eng.range_exceeded(chk) for eng, chk in self.checks):
            self.inner = self.redo()           # (the exact path: no flags of its own)
            self.logits = self.inner.logits    # ... and its logits replace the overflowed ones
            out = self.inner.result()
        self.checks = None
        return out


class File2Beats(Audio2Beats):
    def __call__(self, audio_path):
        signal, sr = load_audio(audio_path)
        return super().__call__(signal, sr)


class File2File(File2Beats):
    def __call__(self, audio_path, output_path):
        downbeats, beats = super().__call__(audio_path)  # (sic) argument naming as in inference.py:313-315
        save_beat_tsv(downbeats, beats, output_path)


_RESAMPLE_FILTERS: dict = {}


def _resample_filter(up: int, down: int, device):
    from . import tables

    key = (up, down, torch.device(device))
    if key not in _RESAMPLE_FILTERS:
        h, half = tables.resample_filter(up, down)
        _RESAMPLE_FILTERS[key] = (torch.from_numpy(h.astype(np.float32)).to(device), half)
    return _RESAMPLE_FILTERS[key]


def batch_predict_aggregate(spect: torch.Tensor, frame_off, chunk_size: int, border_size: int, model, checks=None):
    """``split_predict_aggregate`` (keep_first) for several pieces stored back to back in ``spect`` (rows
    ``frame_off[k]:frame_off[k+1]``): -> (beat, downbeat), concatenated like the input.  Pieces longer than
    ``chunk_size - 2 border_size`` frames share one chunk table: their chunks are gathered slice by slice
    (MAX_CHUNKS_PER_LAUNCH) straight into the model and aggregated by one launch; shorter pieces (one odd-length chunk
    each) go through ``split_predict_aggregate`` one by one.
    A ``BeatThis`` with ``fp32_split_gemms`` (BT_PREC_F32X3) runs its forward slices without waiting for their range flags
    (``Engine.deferred_range_checks``): with ``checks=None`` the flags are looked at before returning and the batch is
    repeated on the exact fp32 path if one fired; a caller that passes a list gets ``(engine, flags)`` appended instead and
    evaluates them itself (``Audio2Beats.many_async``)."""
    guard = isinstance(model, BeatThis) and model._precision() == _lib.PREC_F32X3
    if guard:
        eng = model.engine()
        eng.ensure_positions(chunk_size)   # (the rotary table cannot grow while range checks are pending: grow it first)
        with eng.deferred_range_checks() as flags:
            out = batch_predict_aggregate(spect, frame_off, chunk_size, border_size, _Unguarded(model))
        if checks is not None:
            checks.append((eng, flags))
            return out
        if eng.range_exceeded(flags):
            model.fp32_split_gemms = False
            try:
                return batch_predict_aggregate(spect, frame_off, chunk_size, border_size, model)
            finally:
                model.fp32_split_gemms = True
        return out
    if isinstance(model, _Unguarded):
        model = model.model
    _lib.require_gpu(spect, "spectrogram")
    if spect.dim() != 2 or spect.shape[1] != 128:
        raise ValueError(f"expected a (frames, 128) spectrogram, got {tuple(spect.shape)}")
    spect = spect.to(torch.float32).contiguous()
    dev = spect.device
    frame_off = np.asarray(frame_off, dtype=np.int64)
    total = int(frame_off[-1])
    both = torch.empty((2, total), dtype=torch.float32, device=dev)   # (one buffer: the peak picker takes it without a concatenation)
    beat, down = both[0], both[1]
    rows, pieces = [], []
    for k in range(len(frame_off) - 1):
        lo, hi = int(frame_off[k]), int(frame_off[k + 1])
        n = hi - lo
        if n == 0:
            continue
        if n <= chunk_size - 2 * border_size:  # a single (n + 2 border)-frame chunk
            r = split_predict_aggregate(spect[lo:hi], chunk_size, border_size, "keep_first", model)
            beat[lo:hi], down[lo:hi] = r["beat"], r["downbeat"]
            continue
        c0 = len(rows)
        rows += [(lo + int(st), lo, hi, 0) for st in chunk_starts(n, chunk_size, border_size)]
        pieces.append((lo, hi, c0, len(rows)))
    if not rows:
        return beat, down
    if total + chunk_size >= 2 ** 31:
        raise ValueError("batch too long for 32-bit frame indices: split the track list")
    B = len(rows)
    tab = _lib.upload(np.concatenate([np.asarray(rows, dtype=np.int32).reshape(-1), np.asarray(pieces, dtype=np.int32).reshape(-1)]), dev)
    d_rows, d_pieces = tab[: 4 * B], tab[4 * B:]
    cb = torch.empty((B, chunk_size), dtype=torch.float32, device=dev)
    cd = torch.empty((B, chunk_size), dtype=torch.float32, device=dev)
    lib, st = _lib.lib(), _lib.stream_ptr(dev)
    # Equal slices of at most MAX_CHUNKS_PER_LAUNCH chunks; from CONCURRENT_SLICE_CHUNKS chunks on, at least two of them
    # on two streams: a launch rarely fills its last round of workgroups (the layer tail of 33 chunks is 387 workgroups
    # on 256 CUs), and the other slice's kernels take the idle CUs.  Each stream has its own chunk buffer and workspace.
    n_slices = -(-B // MAX_CHUNKS_PER_LAUNCH)
    if B >= CONCURRENT_SLICE_CHUNKS and CONCURRENT_STREAMS > 1:
        n_slices = max(n_slices, CONCURRENT_STREAMS)
    step = -(-B // n_slices)                 # (66 chunks -> 33 + 33, not 64 + 2)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev, min(n_slices, CONCURRENT_STREAMS)) if n_slices > 1 and CONCURRENT_STREAMS > 1 else [main]
        if side[0] is not main:
            ready = torch.cuda.Event()
            ready.record(main)
        for j, i in enumerate(range(0, B, step)):
            nb = min(step, B - i)
            st_ = side[j % len(side)]
            with torch.cuda.stream(st_):
                if st_ is not main:
                    st_.wait_event(ready)
                chunks = torch.empty((nb, chunk_size, 128), dtype=torch.float32, device=dev)
                _lib.check(lib.bt_split_chunks_batch(_lib.stream_ptr(dev), spect.data_ptr(), d_rows[4 * i:].data_ptr(), nb,
                                                     chunk_size, chunks.data_ptr()))
                if isinstance(model, BeatThis) and not _model_hooked(model):
                    model._run(chunks, 0, 2, out=(cb[i: i + nb], cd[i: i + nb]))   # logits straight into their rows
                else:
                    r = model(chunks)
                    cb[i: i + nb], cd[i: i + nb] = r["beat"], r["downbeat"]
        for st_ in side:
            if st_ is not main:
                main.wait_stream(st_)
        st = _lib.stream_ptr(dev)
        max_frames = max(hi - lo for lo, hi, _, _ in pieces)
        _lib.check(lib.bt_aggregate_batch(st, cb.data_ptr(), cd.data_ptr(), d_rows.data_ptr(), d_pieces.data_ptr(),
                                          len(pieces), max_frames, chunk_size, border_size, beat.data_ptr(),
                                          down.data_ptr()))
    return beat, down


class _Unguarded:
    """Marks a model whose range checks the caller has already taken over (batch_predict_aggregate's inner call)."""

    def __init__(self, model):
        self.model = model


def resample_gpu(signal: torch.Tensor, in_rate: int, out_rate: int) -> torch.Tensor:
    """1-D fp32 device waveform at ``in_rate`` -> ``out_rate`` on the GPU (csrc/frontend.hip: resample_kernel), the
    MI355X replacement of the reference's host ``soxr.resample`` (inference.py:274-275).  Polyphase
    Kaiser-windowed-sinc FIR designed to libsoxr's HQ specification (tables.resample_filter: flat to 91.3 % of the lower
    Nyquist frequency, 125 dB rejection from 100 %); like every non-libsoxr resampler it is not bit-compatible with soxr:
    parity is defined from the 22.05 kHz waveform onwards."""
    from math import gcd

    _lib.require_gpu(signal, "waveform")
    if signal.dim() != 1:
        raise ValueError(f"expected a 1-D waveform, got shape {tuple(signal.shape)}")
    in_rate, out_rate = int(in_rate), int(out_rate)
    g = gcd(in_rate, out_rate)
    up, down = out_rate // g, in_rate // g
    if up == down:
        return signal
    h, half = _resample_filter(up, down, signal.device)
    x = signal.to(torch.float32).contiguous()
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    y = torch.empty(n_out, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().bt_resample(_lib.stream_ptr(x.device), x.data_ptr(), n_in, up, down, h.data_ptr(), half,
                                          y.data_ptr(), n_out))
    return y

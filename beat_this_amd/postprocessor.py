"""``Postprocessor`` -- drop-in for beat_this.model.postprocessor.Postprocessor
(postprocessor.py:9-173).  "minimal": the peak mask (x == maxpool7(x) and x > 0) and the
ordered index compaction run in one HIP kernel per call; the handful of surviving indices
go to the host where deduplicate_peaks / nearest-beat snapping / unique run in C++
(bt_postprocess_host, float64 like numpy).  No thread pool.  Logits that live in host memory (the
reference accepts CPU tensors, postprocessor.py:58-83) get the same peak mask from the library's host entry
point (bt_peaks_host).  ``ragged`` (extension) post-processes many tracks stored back to back with one launch
and one device-to-host copy.  "dbn" defers to madmom exactly like the reference (not installed here; out of
scope)."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib

_PIN_LOCK = threading.Lock()


def _host_post(bidx: np.ndarray, didx: np.ndarray, fps: float):
    bidx = np.ascontiguousarray(bidx, dtype=np.int32)
    didx = np.ascontiguousarray(didx, dtype=np.int32)
    beats = np.empty(max(len(bidx), 1), dtype=np.float64)
    downs = np.empty(max(len(didx), 1), dtype=np.float64)
    nb, nd = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().bt_postprocess_host(bidx.ctypes.data, len(bidx), didx.ctypes.data, len(didx), float(fps),
                                              beats.ctypes.data, C.byref(nb), downs.ctypes.data, C.byref(nd)))
    return beats[: nb.value].copy(), downs[: nd.value].copy()


def deduplicate_peaks(peaks, width=1) -> np.ndarray:
    """Groups of adjacent peak frame indices that are each not more than ``width`` frames apart (from the group's running mean)
    replaced by their mean -- the reference's public helper (postprocessor.py:176-197), on the library's host entry point."""
    idx = np.ascontiguousarray(np.fromiter(map(int, peaks), dtype=np.int64), dtype=np.int32)
    out = np.empty(max(len(idx), 1), dtype=np.float64)
    m = C.c_int32(0)
    _lib.check(_lib.lib().bt_deduplicate_peaks_host(idx.ctypes.data, len(idx), float(width), out.ctypes.data, C.byref(m)))
    return out[: m.value].copy()


class PendingBeats:
    """Handle of an enqueued ``Postprocessor.ragged_async`` call; ``result()`` -> [(beats, downbeats)] per track."""

    def __init__(self, host, done, frame_off, total, fps, owner=None, keep=None):
        self._host, self._done, self._frame_off, self._total, self._fps = host, done, frame_off, total, fps
        self._owner, self._keep, self._out = owner, keep, None

    def result(self):
        if self._out is not None:
            return self._out
        n = len(self._frame_off) - 1
        if self._host is None:
            self._out = [(np.zeros(0), np.zeros(0)) for _ in range(n)]
            return self._out
        self._done.synchronize()
        host, total = self._host.numpy(), self._total
        cnt = host[2 * total:]
        out = []
        for k in range(n):
            lo = int(self._frame_off[k])
            out.append(_host_post(host[lo: lo + cnt[2 * k]], host[total + lo: total + lo + cnt[2 * k + 1]], self._fps))
        if self._owner is not None:  # hand the pinned buffer back
            self._owner.__dict__.setdefault("_pin_pool", []).append(self._host._base if self._host._base is not None else self._host)
        self._host = self._keep = None
        self._out = out
        return out


class Postprocessor:
    def __init__(self, type: str = "minimal", fps: int = 50):
        assert type in ["minimal", "dbn"]
        self.type = type
        self.fps = fps
        if type == "dbn":
            from madmom.features.downbeats import DBNDownBeatTrackingProcessor

            self.dbn = DBNDownBeatTrackingProcessor(beats_per_bar=[3, 4], min_bpm=55.0, max_bpm=215.0, fps=self.fps,
                                                    transition_lambda=100)

    def __call__(self, beat: torch.Tensor, downbeat: torch.Tensor, padding_mask: torch.Tensor | None = None):
        batched = beat.ndim != 1
        if not batched:
            beat, downbeat = beat.unsqueeze(0), downbeat.unsqueeze(0)
            padding_mask = None if padding_mask is None else padding_mask.unsqueeze(0)
        if self.type == "minimal":
            pb, pd = self.postp_minimal(beat, downbeat, padding_mask)
        else:
            pb, pd = self.postp_dbn(beat, downbeat, padding_mask)
        return (pb, pd) if batched else (pb[0], pd[0])

    def postp_minimal(self, beat, downbeat, padding_mask=None):
        B, T = beat.shape
        if (B == 1 and padding_mask is None and beat.is_cuda and beat.dtype == downbeat.dtype == torch.float32 and beat.is_contiguous()
                and downbeat.is_contiguous() and beat.untyped_storage().data_ptr() == downbeat.untyped_storage().data_ptr()
                and downbeat.data_ptr() == beat.data_ptr() + 4 * T):
            logits = torch.as_strided(beat, (1, 2, T), (2 * T, T, 1))   # (the two rows of one buffer, as split_predict_aggregate leaves them: no copy)
        else:
            logits = torch.stack([beat, downbeat], 1).to(torch.float32)  # (B, 2, T)
        if padding_mask is not None:
            logits = logits.masked_fill(~padding_mask.bool().unsqueeze(1), -1000.0)
        logits = logits.contiguous()
        if not logits.is_cuda:  # host logits: the library's host peak mask, same definition
            idx_h = np.zeros((B * 2, max(T, 1)), dtype=np.int32)
            cnt_h = np.zeros(B * 2, dtype=np.int32)
            flat = logits.view(B * 2, T).numpy()
            for a in range(B * 2):
                c = C.c_int32(0)
                _lib.check(_lib.lib().bt_peaks_host(flat[a].ctypes.data, T, idx_h[a].ctypes.data, C.byref(c)))
                cnt_h[a] = c.value
        else:
            # indices and counts travel in ONE buffer -> one device-to-host copy, one synchronisation
            buf = torch.empty((B * 2 * T + B * 2,), dtype=torch.int32, device=beat.device)
            with torch.cuda.device(beat.device):
                _lib.check(_lib.lib().bt_peaks(_lib.stream_ptr(beat.device), logits.data_ptr(), T, B * 2, buf.data_ptr(),
                                               buf[B * 2 * T:].data_ptr()))
            host = buf.cpu().numpy()
            cnt_h, idx_h = host[B * 2 * T:], host[: B * 2 * T].reshape(B * 2, T)
        out_b, out_d = [], []
        for b in range(B):
            frames_b = idx_h[2 * b, : cnt_h[2 * b]]
            frames_d = idx_h[2 * b + 1, : cnt_h[2 * b + 1]]
            if padding_mask is not None:
                # reference truncates masked frames before nonzero(); indices shift accordingly
                m = padding_mask[b].bool().cpu().numpy()
                remap = np.cumsum(m) - 1
                frames_b, frames_d = remap[frames_b], remap[frames_d]
            bt, dt = _host_post(frames_b, frames_d, self.fps)
            out_b.append(bt)
            out_d.append(dt)
        return tuple(out_b), tuple(out_d)

    def ragged(self, beat: torch.Tensor, downbeat: torch.Tensor, frame_off):
        """Extension: "minimal" post-processing of many tracks stored back to back (track k = frames
        ``frame_off[k]:frame_off[k+1]`` of the 1-D ``beat`` / ``downbeat`` device tensors) -> [(beats, downbeats)]:
        one peak-picking launch and one device-to-host copy for all tracks, then the C++ host step per track."""
        return self.ragged_async(beat, downbeat, frame_off).result()

    def ragged_async(self, beat: torch.Tensor, downbeat: torch.Tensor, frame_off) -> "PendingBeats":
        """``ragged`` split in two: everything up to the (asynchronous, pinned-memory) device-to-host copy is enqueued
        now; ``.result()`` waits for that copy and runs the host step.  A caller with several batches enqueues batch
        i + 1 before collecting batch i, so the GPU never idles during the host step."""
        assert self.type == "minimal"
        _lib.require_gpu(beat, "beat logits")
        frame_off = np.asarray(frame_off, dtype=np.int64)
        n = len(frame_off) - 1
        total = int(frame_off[-1])
        if n == 0 or total == 0:
            return PendingBeats(None, None, frame_off, 0, self.fps)
        dev = beat.device
        b1, d1 = beat.reshape(-1), downbeat.reshape(-1)
        if (b1.dtype == d1.dtype == torch.float32 and b1.is_contiguous() and d1.is_contiguous() and b1.numel() == total
                and b1.untyped_storage().data_ptr() == d1.untyped_storage().data_ptr()
                and d1.data_ptr() == b1.data_ptr() + 4 * total):
            logits = b1   # (the batched forward writes both rows into one buffer: bt_peaks_batch addresses them by offset)
        else:
            logits = torch.cat([b1.float(), d1.float()])   # [2 * total]
        lens = (frame_off[1:] - frame_off[:-1]).astype(np.int32)
        spans = np.empty((2 * n, 2), dtype=np.int32)     # array 2 k = beat of track k, 2 k + 1 = its downbeat
        spans[0::2, 0], spans[1::2, 0] = frame_off[:-1], total + frame_off[:-1]
        spans[0::2, 1] = spans[1::2, 1] = lens
        size = 2 * total + 2 * n
        buf = torch.empty((size,), dtype=torch.int32, device=dev)   # [indices | counts]
        with torch.cuda.device(dev):
            d_spans = _lib.upload(spans, dev)
            _lib.check(_lib.lib().bt_peaks_batch(_lib.stream_ptr(dev), logits.data_ptr(), d_spans.data_ptr(), 2 * n,
                                                 buf.data_ptr(), buf[2 * total:].data_ptr()))
            host = self._pinned(size)
            host[:size].copy_(buf, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        return PendingBeats(host[:size], done, frame_off, total, self.fps, owner=self, keep=(buf, logits, d_spans))

    def _pinned(self, size: int) -> torch.Tensor:
        """A pinned host buffer of at least ``size`` int32 (pool of returned buffers: page-locking costs ~a millisecond)."""
        pool = self.__dict__.setdefault("_pin_pool", [])
        with _PIN_LOCK:   # (one Postprocessor may serve several host threads)
            for i, t in enumerate(pool):
                if t.numel() >= size:
                    return pool.pop(i)
        return torch.empty((max(size, 1 << 16),), dtype=torch.int32, pin_memory=True)

    def postp_dbn(self, beat, downbeat, padding_mask=None):
        if padding_mask is None:
            padding_mask = torch.ones_like(beat, dtype=torch.bool)
        eps = 1e-5
        bp = beat.double().sigmoid() * (1 - eps) + eps / 2
        dp = downbeat.double().sigmoid() * (1 - eps) + eps / 2
        out_b, out_d = [], []
        for b in range(beat.shape[0]):
            m = padding_mask[b].bool()
            pb, pd = bp[b][m].cpu().numpy(), dp[b][m].cpu().numpy()
            act = np.vstack((np.maximum(pb - pd, eps / 2), pd)).T
            res = self.dbn(act)
            out_b.append(res[:, 0])
            out_d.append(res[res[:, 1] == 1][:, 0])
        return tuple(out_b), tuple(out_d)

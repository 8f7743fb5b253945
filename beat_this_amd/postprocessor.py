"""``Postprocessor`` -- drop-in for beat_this.model.postprocessor.Postprocessor
(postprocessor.py:9-173).  "minimal": the peak mask (x == maxpool7(x) and x > 0) and the
ordered index compaction run in one HIP kernel per call; the handful of surviving indices
go to the host where deduplicate_peaks / nearest-beat snapping / unique run in C++
(bt_postprocess_host, float64 like numpy).  No thread pool.  "dbn" defers to madmom exactly
like the reference (not installed here; out of scope)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _host_post(bidx: np.ndarray, didx: np.ndarray, fps: float):
    bidx = np.ascontiguousarray(bidx, dtype=np.int32)
    didx = np.ascontiguousarray(didx, dtype=np.int32)
    beats = np.empty(max(len(bidx), 1), dtype=np.float64)
    downs = np.empty(max(len(didx), 1), dtype=np.float64)
    nb, nd = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().bt_postprocess_host(bidx.ctypes.data, len(bidx), didx.ctypes.data, len(didx), float(fps),
                                              beats.ctypes.data, C.byref(nb), downs.ctypes.data, C.byref(nd)))
    return beats[: nb.value].copy(), downs[: nd.value].copy()


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
        _lib.require_gpu(beat, "beat logits")
        B, T = beat.shape
        logits = torch.stack([beat, downbeat], 1).to(torch.float32)  # (B, 2, T)
        if padding_mask is not None:
            logits = logits.masked_fill(~padding_mask.bool().unsqueeze(1), -1000.0)
        logits = logits.contiguous()
        idx = torch.empty((B * 2, T), dtype=torch.int32, device=beat.device)
        cnt = torch.empty((B * 2,), dtype=torch.int32, device=beat.device)
        with torch.cuda.device(beat.device):
            _lib.check(_lib.lib().bt_peaks(_lib.stream_ptr(beat.device), logits.data_ptr(), T, B * 2, idx.data_ptr(),
                                           cnt.data_ptr()))
        cnt_h = cnt.cpu().numpy()
        width = int(cnt_h.max()) if len(cnt_h) else 0
        idx_h = idx[:, : max(width, 1)].cpu().numpy()
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

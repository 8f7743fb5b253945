"""Multi-GPU sharding of the chunk batch: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no in-process multi-GPU path (README.md:53-56 runs N independent CLI
processes).  Chunks are independent forwards (inference.py:215), so the path shards with a
single exchange: weights replicated, the global chunk list block-partitioned over ranks, one
``all_gather`` of the per-chunk logits (2 x 1500 fp32 = 12 KB per chunk), then every rank
holds everything needed for aggregation + post-processing.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import inference as inf


def partition(n_items: int, world: int, rank: int):
    """Contiguous block partition with equal (padded) block size: (lo, hi, per_rank)."""
    per = (n_items + world - 1) // world if n_items else 0
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi, per


def _hip_aggregate(cb, cd, starts, T, border, n):
    from . import _lib

    dev = cb.device
    d_starts = torch.as_tensor(np.asarray(starts, dtype=np.int32), device=dev)
    beat = torch.empty((n,), dtype=torch.float32, device=dev)
    down = torch.empty((n,), dtype=torch.float32, device=dev)
    cb, cd = cb.contiguous(), cd.contiguous()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().bt_aggregate(_lib.stream_ptr(dev), cb.data_ptr(), cd.data_ptr(), d_starts.data_ptr(),
                                           len(starts), T, border, n, beat.data_ptr(), down.data_ptr()))
    return beat, down


def _hip_gather(spect, starts, T):
    return inf._gather_chunks(spect, np.asarray(starts), T)[0]


def forward_chunks_sharded(model, spects, chunk_size=1500, border=6, group=None, gather=_hip_gather,
                           aggregate=_hip_aggregate, run=None):
    """[(T_i,128)] -> [(beat_i, downbeat_i)], chunks of all full-length pieces sharded over the
    ranks of ``group`` (or run locally when torch.distributed is not initialised)."""
    run = run or inf._run_batched
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    spects = [s.to(torch.float32).contiguous() for s in spects]
    plan = []   # (piece, start) for every full-length chunk, in piece order
    meta = []
    for i, s in enumerate(spects):
        n = s.shape[0]
        starts = inf.chunk_starts(n, chunk_size, border)
        T = inf.chunk_length(n, chunk_size, border)
        meta.append((n, starts, T))
        if T == chunk_size:
            plan += [(i, int(st)) for st in starts]
    results = [None] * len(spects)
    # short pieces: a single (n+12)-frame chunk each, cheap -> computed redundantly on every rank
    for i, (n, starts, T) in enumerate(meta):
        if T != chunk_size:
            cb, cd = run(model, gather(spects[i], starts, T))
            results[i] = aggregate(cb.float(), cd.float(), starts, T, border, n)
    if plan:
        dev = spects[0].device
        lo, hi, per = partition(len(plan), world, rank)
        mine = plan[lo:hi]
        parts = []
        j = 0
        while j < len(mine):  # consecutive chunks of one piece are gathered by one kernel launch
            k = j
            while k < len(mine) and mine[k][0] == mine[j][0]:
                k += 1
            parts.append(gather(spects[mine[j][0]], [st for _, st in mine[j:k]], chunk_size))
            j = k
        local = torch.zeros((per, 2, chunk_size), dtype=torch.float32, device=dev)
        if parts:
            chunks = parts[0] if len(parts) == 1 else torch.cat(parts)
            cb, cd = run(model, chunks)
            local[: hi - lo, 0] = cb.float()
            local[: hi - lo, 1] = cd.float()
        if distributed:   # (also in a world of one: the collective path is the same code on 1 and N ranks)
            full = torch.empty((world * per, 2, chunk_size), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(full, local, group=group)
        else:
            full = local
        pos = 0
        for i, (n, starts, T) in enumerate(meta):
            if T == chunk_size:
                seg = full[pos: pos + len(starts)]
                results[i] = aggregate(seg[:, 0].contiguous(), seg[:, 1].contiguous(), starts, T, border, n)
                pos += len(starts)
    return results


def track_frames(n_samples: int, sr: int) -> int:
    """Spectrogram rows of an ``n_samples`` track at sample rate ``sr`` (resampled to 22.05 kHz, hop 441, centred)."""
    from math import gcd

    g = gcd(int(sr), 22050)
    up, down = 22050 // g, int(sr) // g
    n22 = n_samples if up == down else -(-n_samples * up // down)
    return 1 + n22 // 441


def audio2frames_sharded(signals, sr, frames_fn, group=None, device=None):
    """Track-level sharding of ``Audio2Frames.many`` (README.md:53-56 runs N independent CLI processes over a file set; this
    is the in-process form): every rank holds the same list of waveforms, computes the block ``partition`` gives it with
    ``frames_fn(sub_list) -> (beat_cat, downbeat_cat, frame_off)`` (e.g. ``lambda s: a2f.spect2frames_batch(
    *a2f.signal2spect_many(s, sr))`` plus the offsets), and ONE ``all_gather_into_tensor`` of the zero-padded framewise
    logits (2 x frames fp32 per rank: 120 KB per 5-minute track) returns every track's logits to every rank.
    -> [(beat, downbeat)] in the order of ``signals``."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    n = len(signals)
    frames = [track_frames(s.shape[0], sr) for s in signals]
    lo, hi, per = partition(n, world, rank)
    blocks = [partition(n, world, r)[:2] for r in range(world)]
    width = max([sum(frames[a:b]) for a, b in blocks] + [1])
    mine = signals[lo:hi]
    if mine:
        beat, down, off = frames_fn(mine)
        dev = beat.device
    else:
        # (an empty block -- fewer tracks than ranks -- still takes part in the collective, on the device the backend moves:
        # a CPU tensor handed to an RCCL group errors out or hangs the other ranks)
        if device is not None:
            dev = torch.device(device)
        elif distributed and dist.get_backend(group) == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            dev = torch.device("cpu")
        beat = down = torch.zeros(0, device=dev)
    local = torch.zeros((2, width), dtype=torch.float32, device=dev)
    local[0, : beat.shape[0]], local[1, : down.shape[0]] = beat.float(), down.float()
    if distributed:
        full = torch.empty((world * 2, width), dtype=torch.float32, device=dev)  # (concatenation along dim 0)
        dist.all_gather_into_tensor(full, local, group=group)
        full = full.view(world, 2, width)
    else:
        full = local[None]
    out = []
    for r, (a, b) in enumerate(blocks):
        pos = 0
        for k in range(a, b):
            out.append((full[r, 0, pos: pos + frames[k]], full[r, 1, pos: pos + frames[k]]))
            pos += frames[k]
    return out

#!/usr/bin/env python
"""Benchmark of the beat_this inference hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): final0-shaped BeatThis, bf16 MFMA operands / fp32
accumulate, a batch of 16 synthetic 30 s chunks (1500 frames x 128 mels) PER GPU through
BeatThis.forward -- the model invocation of the Spect2Frames path -- with the inputs resident
in HBM.  One step = one such batch on every rank; for N > 1 the per-chunk logits are
all-gathered (RCCL) inside the step, so every rank ends with all logits (weak scaling).
value = N * 16 * 30 audio-seconds / step time (max over ranks).

Extra objects on the JSON line:
  roofline      dominant launch category (attention: attn_frag_kernel, MFMA bound): algorithmic FLOP of
                its launches in one forward / their summed duration, HIP events on the launch stream
                (a separate profiled pass of the same workload after the timed region).
  cpu_baseline  the CPU oracle (torch fp32 restatement of the reference, kind "port") timed on
                this host on a bounded sample (a few single-chunk forwards), rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CHUNK_FRAMES = 1500
CHUNK_SECONDS = 30.0
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}  # MI355X_MICROARCH.md: dense MFMA peaks


def flops_per_chunk(D: int, T: int = CHUNK_FRAMES):
    """Algorithmic MACs*2 per chunk, by launch category of csrc/engine.hip (recomputed;
    the sum is the 134.71 GFLOP / chunk of SURVEY.md section 8d for final0).  bf16 path:
      attn_freq_fused = attnff_fused_kernel   frequency direction: QKV+gates, attention, out-proj AND its FF
      ff_fused        = outff_fused_kernel    time direction: out-proj and its FF
      qkv_gemm        = gemm3 QKV (main layers) + qkv_front_kernel (time-direction QKV of the frontend)
      attn_flash      = attn_frag_kernel      time-direction + main attention
      out_gemm / ff1_gemm / ff2_gemm = gemm3  main layers only"""
    cat = dict(stem=2 * T * 32 * 32 * 12, qkv_gemm=0, attn_freq=0, attn_flash=0, out_gemm=0, ff1_gemm=0, ff2_gemm=0,
               conv_gemm=0, linear_gemm=2 * T * 1024 * D, head=2 * T * D * 2, ff_fused=0, attn_freq_fused=0)
    for blk in range(3):
        Cc, F = 32 << blk, 32 >> blk
        h = Cc // 32
        tokens = T * F
        ff = 2 * 2 * tokens * Cc * 4 * Cc
        qkv = 2 * tokens * Cc * (3 * Cc + h)
        out = 2 * tokens * Cc * Cc
        cat["attn_freq_fused"] += qkv + out + 2 * 2 * T * h * F * F * 32 + ff
        cat["qkv_gemm"] += qkv
        cat["attn_flash"] += 2 * 2 * F * h * T * T * 32
        cat["ff_fused"] += out + ff
        cat["conv_gemm"] += 2 * T * (F // 2) * (6 * Cc) * (2 * Cc)
    H = D // 32
    for _ in range(6):
        cat["qkv_gemm"] += 2 * T * D * (3 * D + H)
        cat["out_gemm"] += 2 * T * D * D
        cat["ff1_gemm"] += 2 * T * D * 4 * D
        cat["ff2_gemm"] += 2 * T * D * 4 * D
        cat["attn_flash"] += 2 * 2 * H * T * T * 32
    return cat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="final0")
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8 = BT_PREC_FP8: bf16 path with the main layers' feed-forward GEMMs on e4m3 (BASELINE config 5)")
    ap.add_argument("--chunks", type=int, default=16, help="chunks per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from beat_this_amd import _lib
    from beat_this_amd import weights as W
    from beat_this_amd.model import BeatThis

    hp = W.resolve_hparams(args.model)
    sd = W.random_state_dict(hp, seed=0, style="init")
    model = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim",
                                           "stem_dim")})
    model.load_state_dict(sd)
    model = model.to(dev)
    B = args.chunks
    x = torch.from_numpy(np.stack([W.synthetic_spect(CHUNK_FRAMES, seed=1000 * rank + i) for i in range(B)])).to(dev)
    half = args.prec != "f32"
    model.fp8_weights = args.prec == "fp8"
    gathered = torch.empty((world * B, 2, CHUNK_FRAMES), dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        with torch.inference_mode(), torch.autocast("cuda", enabled=half):
            r = model(x)
        if world > 1:
            local = torch.stack((r["beat"], r["downbeat"]), 1)
            dist.all_gather_into_tensor(gathered, local)
        return r

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Untimed settling before the W warm-up steps: a fresh process on a fresh box needs a few hundred ms of GPU work
    # before clocks, page tables and RCCL channels are in their steady state (a 15 ms timed region right after start-up
    # once measured 7x slow); part of the set-up, not of the W / K steps of the contract.
    for _ in range(max(5, -(-1600 // B))):  # a fixed count (every rank issues the same collectives): ~0.35 s at 16 chunks
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * CHUNK_SECONDS / (elapsed / args.steps)

    # ---- roofline leg: per-kernel HIP-event timing of the same workload (rank 0) ---------------
    roofline = None
    breakdown = None
    if rank == 0:
        lib = _lib.lib()
        n_prof = 3
        lib.bt_profile_begin()
        for _ in range(n_prof):
            with torch.inference_mode(), torch.autocast("cuda", enabled=half):
                model(x)
        ncat = len(_lib.PROFILE_CATEGORIES)
        ms = (C.c_double * ncat)()
        cnt = (C.c_int32 * ncat)()
        _lib.check(lib.bt_profile_end(ms, cnt, ncat))
        fl = flops_per_chunk(hp["transformer_dim"])
        breakdown = {}
        for i, name in enumerate(_lib.PROFILE_CATEGORIES):
            if cnt[i]:
                t_fwd = ms[i] / n_prof  # ms per forward spent in this category
                breakdown[name] = {"ms_per_step": round(t_fwd, 4), "launches_per_step": cnt[i] // n_prof,
                                   "tflops": round(fl[name] * B / (t_fwd * 1e-3) / 1e12, 2)}
        dom = max(breakdown, key=lambda k: breakdown[k]["ms_per_step"])
        d = breakdown[dom]
        # (fp8 mode: only the feed-forward GEMMs run on e4m3 operands, the other launch categories are the bf16 kernels)
        peak = PEAK_TFLOPS["fp8" if args.prec == "fp8" and dom in ("ff1_gemm", "ff2_gemm") else
                           ("f32" if args.prec == "f32" else "bf16")]
        # HBM bytes per launch of the dominant category: PMC counters collected in separate rocprofv3 passes
        # (tools/pmc_traffic.sh) for exactly this workload, committed under profiles/; null for any other workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if tj["workload"] == {"model": args.model, "prec": args.prec, "chunks": B} and dom in tj:
                traffic = tj[dom]["bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": round(d["tflops"] / peak, 4), "traffic": traffic,
                    "avg_launch_ms": round(d["ms_per_step"] / d["launches_per_step"], 4),
                    "flop_per_launch": fl[dom] * B / d["launches_per_step"],
                    "whole_forward_tflops": round(sum(fl.values()) * B / (ms_per_step * 1e-3) / 1e12, 2)}

    # ---- CPU baseline: the oracle on this host, bounded sample, rank 0 at N = 1 only ------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import beat_this_oracle as O

        xc = x[:1].cpu()
        with torch.inference_mode():
            # torch's default (one thread per logical core) oversubscribes big hosts badly
            # (128 threads: 12-23 s per chunk vs 1.2 s with 16 on the 2 x EPYC 9575F box): probe.
            best = None
            for nt in (8, 16, 32):
                if nt > (os.cpu_count() or 1):
                    continue
                torch.set_num_threads(nt)
                O.model_forward(sd, xc)  # warm-up at this thread count
                tp = time.perf_counter()
                O.model_forward(sd, xc)
                tp = time.perf_counter() - tp
                if best is None or tp < best[1]:
                    best = (nt, tp)
            torch.set_num_threads(best[0])
            reps = 0
            t1 = time.perf_counter()
            while True:
                O.model_forward(sd, xc)
                reps += 1
                if time.perf_counter() - t1 > 12.0 or reps >= 20:
                    break
            tc = (time.perf_counter() - t1) / reps
        cpu = {"value": round(CHUNK_SECONDS / tc, 2), "unit": "audio-seconds/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"{reps} x BeatThis.forward of one 1500-frame chunk (30 s), fp32, "
                                          f"oracle/beat_this_oracle.py, {tc * 1e3:.0f} ms each"}

    if rank == 0:
        out = {
            "metric": "audio-seconds processed/sec", "value": round(value, 1), "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f32": "f32", "fp8": "bf16+fp8(e4m3 feed-forward GEMMs)"}[args.prec], "data": "synthetic",
            "config": {"workload": f"{args.model} BeatThis.forward (Spect2Frames path), {B} x 30 s chunks "
                                   f"(1500 frames x 128 mels) per GPU, random-init weights, logits all-gathered",
                       "chunks_per_gpu": B, "global_chunks": world * B, "parallelism": f"chunk-sharded x{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "breakdown": breakdown,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

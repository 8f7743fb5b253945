#!/usr/bin/env python
"""Benchmark of the beat_this inference hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W        (N > 1 re-launches itself under torch.distributed.run)

Headline workload = BASELINE.json's metric: final0-shaped BeatThis, 5-minute (300 s) synthetic 44.1 kHz mono tracks
through ``Audio2Beats`` -- GPU resampler -> log-mel -> chunk gather -> BeatThis.forward -> keep_first aggregation -> peak
picking -> device-to-host copy of the peak indices -> C++ host post-processing -- audio in, beat / downbeat times out, in the
precision of the API's DEFAULT (``float16=False``): BT_PREC_F32X3, fp32 activations with every product on three fp16 MFMAs
over hi + lo operand halves -- the path that carries north_star's gate (logits within 1e-3 of the CPU reference, identical
beats), which the in-run `parity` object shows on every run.  One step = one batch of TRACKS_PER_GPU = 6 tracks (66 chunks
of 1500 frames; BASELINE config 4's per-GPU share is 64) on every rank, waveforms resident in HBM when the timed region
starts; step i + 1 is enqueued before the host part of step i is collected (Audio2Beats.many_async).  For N > 1 every
rank processes its own tracks and the framewise logits are all-gathered (RCCL) inside the step -- weak scaling.
value = N * 6 * 300 audio-seconds / step time (max over ranks).

Extra objects on the JSON line:
  parity        GPU logits / beats of track 0 against the CPU oracle run on the same waveform (the run cpu_baseline times):
                max |logit difference|, beat / downbeat frame flips.
  roofline      dominant launch category of the forward (attention, MFMA bound): algorithmic FLOP per launch / average
                launch duration from HIP events on the launch stream (bt_profile_*, a profiled pass of the same workload
                right after the timed region); peak = the dense fp16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s), frac =
                achieved / peak; pipe_utilisation = frac x the fp16 MFMAs this arithmetic issues per product (3; 2.5 - 3 in the
                attention); whole_step_frac = SURVEY 8d's 134.71 GFLOP x chunks / ms_per_step / peak;
                traffic = HBM bytes per launch from the committed PMC passes (labelled).
  half_path     the same workload with float16=True (fp16 MFMA operands, what BASELINE configs 2 / 4 call bf16): NOT under the
                gate -- its parity object says by how much, next to the reference's own fp16-autocast error -- with its own
                roofline (dense fp16 peak) and breakdown.
  fp32_exact_path  the same workload on exact fp32 MFMAs (float16="exact"), the fallback of the default path.
  latency       single-file latency (BASELINE config 1: File2Beats on one 30 s track, and one 300 s track), host waveform in
                -> beat times out, one call at a time, in the three precisions, next to the CPU oracle on the same inputs.
  stress_weights  the default path on trained-like "outlier" weights (residual outlier channels of ~10^3, heavy-tailed
                matrices): throughput, parity, and how many batches fell back to the exact path (range guard).
  frontend      the HBM-bound stages (resample, log-mel, chunk gather, aggregation, peak picking): ms per step and GB/s of
                ALGORITHMIC bytes (3.41 MB per chunk for the log-mel).
  forward_only  BASELINE config 2 (16 chunks through BeatThis.forward, spectrograms resident) in the headline precision and
                in half precision.
  configs       the other BASELINE.json configurations as short legs: cfg3 (small0 exact fp32, 128 chunks: fp32-FLOP
                fraction AND counter-measured HBM GB/s), cfg4_share (final0, 64 chunks = one GPU's share of the 512-chunk
                job, both precisions), cfg5 (the 64-chunk batch with the cross terms of the hi + lo products on the block-scaled
                fp8 MFMA, opt-in levels 1 and 2, next to the default path).
  strong_scaling_cfg4  BASELINE config 4 itself: 512 chunks sharded over the N ranks (512 / N each), logits all-gathered.
  timed_region  the K steps are repeated until the timed region holds >= 2 s of GPU work (clocks and temperatures settle);
                ms_per_step = region / (K x repeats).  rccl_ranks = ranks counted by an RCCL all-reduce (the process group
                is brought up at N = 1 as well, so every run exercises the N > 1 code path).
  host_inclusive  the headline job with the waveforms starting in pinned HOST memory (PCIe-inclusive rate; never `value`).
  cpu_baseline  the CPU oracle's Audio2Beats (torch fp32, SDPA attention like the reference) on ONE 300 s track on this
                host, three repeats (value = the fastest, median beside it), thread count probed and stated (rank 0, N = 1).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK_FRAMES = 1500
FRESH_SECONDS_PER_CHUNK = 29.76   # 1488 fresh frames at 50 fps
TRACK_SECONDS = 300.0
TRACK_SR = 44100
TRACKS_PER_GPU = 6
# MI355X_MICROARCH.md: dense MFMA peaks.  roofline.frac is ALGORITHMIC flops / the guide's dense peak of the MFMA the path
# issues (fp16 for half and f32x3, fp32 for the exact path).  The hi + lo path spends three fp16 MFMAs per product (its
# attention 2.5 / 2.75 / 3 by BT_OPT_X3_ATTN_P16): that is a cost of the chosen arithmetic, not a lower ceiling -- the pipe's
# own occupancy is reported beside it as `pipe_utilisation` = frac x MFMAs per product.
PEAK_TFLOPS = {"half": 2500.0, "f32": 157.3, "f32x3": 2500.0}
MFMAS_PER_PRODUCT = {"half": 1.0, "f32": 1.0, "f32x3": 3.0}
PEAK_HBM_GBS = 8000.0


def flops_per_chunk(D: int, T: int = CHUNK_FRAMES, ff_mult: int = 4):
    """Algorithmic MACs*2 per chunk, by launch category of csrc/engine.hip (recomputed;
    the sum is the 134.71 GFLOP / chunk of SURVEY.md section 8d for final0).  Half path:
      attn_freq_fused = attnff_fused_kernel   frequency direction: QKV+gates, attention, out-proj AND its FF
      ff_fused        = outff_fused_kernel    time direction: out-proj and its FF
      qkv_gemm        = gemm3 QKV (main layers) + qkv_front_kernel (time-direction QKV of the frontend)
      attn_flash      = attn_frag_kernel      time-direction + main attention
      out_gemm / ff1_gemm / ff2_gemm = gemm3  main layers only"""
    cat = dict(stem=2 * T * 32 * 32 * 12, qkv_gemm=0, attn_flash=0, out_gemm=0, ff1_gemm=0, ff2_gemm=0,
               conv_gemm=0, linear_gemm=2 * T * 1024 * D, head=2 * T * D * 2, ff_fused=0, attn_freq_fused=0, layer_tail=0)
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
        cat["ff1_gemm"] += 2 * T * D * ff_mult * D
        cat["ff2_gemm"] += 2 * T * D * ff_mult * D
        cat["attn_flash"] += 2 * 2 * H * T * T * 32
    # layer_tail_kernel = out-projection + FF1 + FF2 of a main layer in one launch (when it runs, those three are absent)
    cat["layer_tail"] = cat["out_gemm"] + cat["ff1_gemm"] + cat["ff2_gemm"]
    return cat


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (the JSON line on stdout stays alone): a stuck phase is visible in the captured log"""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="final0")
    ap.add_argument("--prec", default="f32x3", choices=["half", "f32", "f32x3"],
                    help="f32x3 (default, = float16=False of the Python API) = fp32-class results on hi + lo fp16 operands, three "
                         "MFMAs per product (BT_PREC_F32X3): the path under the 1e-3 / identical-beats gate; half = fp16 MFMA "
                         "operands (float16=True); f32 = exact fp32 MFMA (float16='exact')")
    ap.add_argument("--tracks", type=int, default=TRACKS_PER_GPU, help="5-minute tracks per GPU per step")
    ap.add_argument("--workload", default="tracks", choices=["tracks", "forward"],
                    help="tracks = Audio2Beats on 300 s 44.1 kHz tracks (BASELINE metric); forward = BeatThis.forward on "
                         "--chunks resident spectrogram chunks (BASELINE configs 2 / 3 / 5)")
    ap.add_argument("--chunks", type=int, default=16, help="--workload forward: chunks per GPU per step")
    ap.add_argument("--slice", type=int, default=0, help="chunks per forward launch (0 = the library default)")
    ap.add_argument("--streams", type=int, default=0, help="streams the forward slices of a step run on (0 = the library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side legs (other precisions, latency, configs, frontend)")
    ap.add_argument("--no-dist", action="store_true", help="N = 1 without bringing up the RCCL process group")
    ap.add_argument("--x3-p16", type=int, default=1, choices=[0, 1, 2],
                    help="BT_OPT_X3_ATTN_P16 of the default precision: probabilities enter P.V as fp16 hi parts in the main layers only "
                         "(1, the library's default: chosen by the flip-soak rule, DESIGN.md section 3), in the frontend as well (2, "
                         "round 5's default), nowhere = three-term P.V of rounds 3 - 4 (0)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the K timed steps are repeated until the timed region is at least this long (0: exactly K steps)")
    ap.add_argument("--watchdog", type=int, default=900, help="seconds after which a stuck run dumps its stacks and exits")
    ap.add_argument("--share-gpu", action="store_true",
                    help="development / tests: all N ranks run on cuda:0 and the collectives go over gloo -- the N > 1 code path of "
                         "this file (sharding, gathers, timing reduction, failure handling) on a box with ONE GPU; not a measurement")
    ap.add_argument("--fail-rank", type=int, default=-1, help="development / tests: this rank raises inside timed step --fail-step")
    ap.add_argument("--fail-step", type=int, default=0)
    args = ap.parse_args()
    import faulthandler

    faulthandler.dump_traceback_later(args.watchdog, exit=True)  # a hang becomes a traceback on stderr, not a silent timeout

    if args.gpus > 1 and "RANK" not in os.environ:
        # fail fast with a clear message when the box does not have the GPUs asked for (instead of N ranks fighting over
        # cuda:0 until a collective's watchdog fires)
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus and not (args.share_gpu and have >= 1):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (torch.cuda.device_count()); "
                             "run with --gpus <= that, one process per GPU")
        # self-launch: one process per GPU under torch.distributed.run (RCCL rendezvous on 127.0.0.1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)

    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1 behind Python's back (RCCL
    # prints a version banner there when the process group comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but this node shows {torch.cuda.device_count()} GPU(s)")
    if world > 1:
        # N launch loops on one host: give every rank its own slice of the cores (affinity by LOCAL_RANK) and size torch's
        # intra-op pool to it, so that eight Python launch threads + their helpers do not migrate over / fight for the same cores
        try:
            cores = sorted(os.sched_getaffinity(0))
            per_rank = max(1, len(cores) // int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            mine = cores[local_rank * per_rank: (local_rank + 1) * per_rank] or cores
            os.sched_setaffinity(0, mine)
            torch.set_num_threads(max(1, min(len(mine), 16)))
        except (AttributeError, OSError) as e:  # noqa: PERF203  (no affinity interface: leave the scheduler alone)
            log(f"rank {rank}: host threads not pinned ({type(e).__name__})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # N = 1 brings the RCCL process group up as well (a world of one): the logits all-gather, the barriers and the
    # all-reduced time of the N > 1 path then run on every driver invocation; --no-dist (or a group that fails to come up)
    # gives the plain single-process run, rccl_ranks = null
    use_dist = world > 1 or not args.no_dist
    dist_note = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        try:
            from datetime import timedelta

            # (a collective whose peer died is aborted after three minutes instead of RCCL's default ten; a rank that RAISES
            # takes the whole job down at once: see the wrapper around main())
            if args.share_gpu:
                dist.init_process_group("gloo", timeout=timedelta(seconds=180))
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=180))
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise
            use_dist, dist_note = False, f"RCCL group of one did not come up ({type(e).__name__}): single-process run"
            log(dist_note)
    def all_gather(out_t, in_t):
        """dist.all_gather_into_tensor; --share-gpu (gloo): through host copies"""
        if args.share_gpu:
            tmp = torch.empty(out_t.shape, dtype=out_t.dtype)
            dist.all_gather_into_tensor(tmp, in_t.cpu().contiguous())
            out_t.copy_(tmp)
        else:
            dist.all_gather_into_tensor(out_t, in_t)

    def all_reduce(t_, op=None):
        if args.share_gpu:
            tmp = t_.cpu()
            dist.all_reduce(tmp, **({} if op is None else {"op": op}))
            t_.copy_(tmp)
        else:
            dist.all_reduce(t_, **({} if op is None else {"op": op}))

    rccl_ranks = None
    if use_dist:   # proof of N ranks: an RCCL all-reduce counts them
        one = torch.ones(1, dtype=torch.int32, device=dev)
        all_reduce(one)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size()

    from beat_this_amd import _lib
    from beat_this_amd import inference as _inf
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Beats

    if args.slice > 0:
        _inf.MAX_CHUNKS_PER_LAUNCH = args.slice
    if args.streams > 0:
        _inf.CONCURRENT_STREAMS = args.streams
    from beat_this_amd.model import BeatThis

    log(f"rank {rank}/{world} on {dev}, library half type {_lib.half_dtype_name()}")
    hp = W.resolve_hparams(args.model)
    # "lively" seeded weights (O(1) logits, sharp softmaxes, beats to pick): same FLOPs as any other weights of this
    # architecture, but the post-processor has real work and the parity figures mean something
    sd = W.random_state_dict(hp, seed=1, style=os.environ.get("BT_BENCH_STYLE", "lively"))   # (BT_BENCH_STYLE=outlier: development A/B of the stress weights)
    model = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    model.load_state_dict(sd)
    a2b = Audio2Beats(checkpoint_path=None, device=dev, float16={"half": True, "f32": "exact", "f32x3": False}[args.prec], dbn=False)
    a2b.model = model.to(dev)
    a2b.model.engine().set_options({"x3_attn_p16": args.x3_p16})
    half_name = _lib.half_dtype_name()
    # BASELINE configs 2 / 4 say "bf16": the reference's float16=True on a GPU IS fp16 autocast (inference.py:245-246, cli.py:82),
    # which is what this path runs; bfloat16 operands are a build switch of the same kernels (tools/build_variant.py -DBT_HALF_BF16,
    # report of the round: profiles/r06_bf16_variant.txt), 8x coarser at the same MFMA rate
    half_desc = "fp16 operands, f32 accumulate (= the reference's GPU autocast); bf16 build of the same kernels: -DBT_HALF_BF16" \
        if half_name == "f16" else "bf16 operands, f32 accumulate (-DBT_HALF_BF16 build)"
    DTYPE = {"half": half_desc, "f32": "f32 (v_mfma_f32_32x32x2_f32)",
             "f32x3": "f32 activations; every product: 3 x v_mfma_f32_32x32x16_f16 on hi + lo operands (BT_PREC_F32X3)"}

    def set_prec(prec):
        """precision of everything that follows: the API's float16=True / False (default, f32x3) / "exact" """
        a2b.float16 = prec == "half"
        a2b.model.fp32_split_gemms = prec == "f32x3"
    set_prec(args.prec)
    half = args.prec == "half"

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- workload -------------------------------------------------------------------------------------------------
    if args.workload == "tracks":
        n_tr = args.tracks
        tracks = [torch.from_numpy(W.synthetic_audio(TRACK_SECONDS, seed=100 * rank + i, sr=TRACK_SR)).to(dev)
                  for i in range(n_tr)]
        from beat_this_amd.parallel import track_frames

        frames_per_track = track_frames(tracks[0].shape[0], TRACK_SR)
        gathered = torch.empty((world * 2, n_tr * frames_per_track), dtype=torch.float32, device=dev) if use_dist else None
        pending = []

        def step():
            h = a2b.many_async(tracks, TRACK_SR)
            if use_dist:
                beat, down, _ = h.logits
                all_gather(gathered, torch.stack((beat, down)))
            pending.append(h)
            if len(pending) > 1:
                return pending.pop(0).result()
            return None

        def drain():
            out = None
            while pending:
                out = pending.pop(0).result()
            return out

        units_per_step = world * n_tr * TRACK_SECONDS
        chunks_per_step = n_tr * 11
        n_settle = 4
        workload = (f"{args.model} Audio2Beats on {n_tr} x {TRACK_SECONDS:.0f} s synthetic {TRACK_SR / 1000:.1f} kHz mono tracks per GPU "
                    f"(resample, log-mel, {chunks_per_step} chunks through BeatThis.forward, aggregation, peak picking, host "
                    f"post-processing), seeded random weights")
    else:
        B = args.chunks
        x = torch.from_numpy(np.stack([W.synthetic_spect(CHUNK_FRAMES, seed=1000 * rank + i) for i in range(B)])).to(dev)
        gathered = torch.empty((world * B, 2, CHUNK_FRAMES), dtype=torch.float32, device=dev) if use_dist else None

        def step():
            with torch.inference_mode(), torch.autocast("cuda", enabled=half):
                r = a2b.model(x)
            if use_dist:
                all_gather(gathered, torch.stack((r["beat"], r["downbeat"]), 1))
            return r

        def drain():
            return None

        units_per_step = world * B * FRESH_SECONDS_PER_CHUNK
        chunks_per_step = B
        n_settle = max(5, -(-1600 // B))
        workload = (f"{args.model} BeatThis.forward (Spect2Frames path), {B} chunks of 1500 frames x 128 mels per GPU resident in "
                    f"HBM (29.76 s of fresh audio each), seeded random weights")

    # Untimed settling before the W warm-up steps: a fresh process on a fresh box needs a few hundred ms of GPU work
    # before clocks, page tables and RCCL channels are in their steady state; part of the set-up, not of the W / K steps.
    log("workload resident, settling")
    for _ in range(n_settle):
        step()
    drain()
    fence()
    log("warm-up")
    tw = time.perf_counter()
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    tw = (time.perf_counter() - tw) / max(1, args.warmup)
    # The K steps are timed `repeats` times back to back so that the region holds >= --min-seconds of GPU work (0.3 s of
    # a 20-step default run is over before clocks and temperatures have settled); every rank uses the same count.
    repeats = 1
    if args.min_seconds > 0 and args.warmup > 0:
        est = torch.tensor([tw * args.steps], dtype=torch.float64, device=dev)
        if use_dist:
            all_reduce(est, dist.ReduceOp.MIN)
        repeats = max(1, int(-(-args.min_seconds // max(float(est.item()), 1e-6))))
    log(f"timed region: {args.steps} steps x {repeats}")
    # package energy over the timed region from the SMU's accumulator (tools/smi.py: rsmi_dev_energy_count_get), read right
    # before the first and right after the last step's fence -- joules are what this path is bound by
    from tools.smi import EnergyMeter

    meter = EnergyMeter(dev)
    meter.start()
    t0 = time.perf_counter()
    for i_step in range(args.steps * repeats):
        if rank == args.fail_rank and i_step == args.fail_step:
            raise RuntimeError(f"--fail-rank {rank}: raising inside timed step {i_step} (test of the N > 1 failure rule)")
        step()
    last = drain()
    fence()
    elapsed = time.perf_counter() - t0
    joules = meter.stop()
    n_timed = args.steps * repeats
    energy = {"available": False, "reason": meter.why}
    if joules is not None:
        jt = torch.tensor([joules[0]], dtype=torch.float64, device=dev)
        if use_dist:
            all_reduce(jt)   # whole job: the sum over the ranks' packages
        energy = {"available": True, "joules_per_step": round(float(jt.item()) / n_timed, 3),
                  "audio_seconds_per_joule": round(units_per_step * n_timed / max(float(jt.item()), 1e-9), 2),
                  "avg_package_power_W": round(joules[0] / max(joules[1], 1e-9), 1), "packages": world,
                  "counter_resolution_uJ": round(meter.smi.resolution_uj, 3), "power_cap_W": meter.smi.cap_w(),
                  "region_seconds_by_counter": round(joules[1], 3),
                  "source": "rsmi_dev_energy_count_get before / after the timed region (idle power of the package included)"}
        log(f"energy: {energy['joules_per_step']} J / step, {energy['avg_package_power_W']} W average")
    log(f"timed region done: {1e3 * elapsed / n_timed:.3f} ms / step over {elapsed:.2f} s")
    per_rank_ms = [round(1e3 * elapsed / n_timed, 3)]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        all_gather(allt, t)
        per_rank_ms = [round(1e3 * float(v) / n_timed, 3) for v in allt.tolist()]   # (which rank is the slow one, if any)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / n_timed
    value = units_per_step / (elapsed / n_timed)

    # ---- BASELINE config 4 as written: 512 chunks sharded over the ranks, logits all-gathered (strong scaling) ------------
    strong = None
    if args.workload == "tracks" and not args.no_extras:
        log("strong-scaling leg (512 chunks / N)")
        per = 512 // world
        xs = torch.from_numpy(np.stack([W.synthetic_spect(CHUNK_FRAMES, seed=50000 + 512 * rank + i) for i in range(min(per, 64))])).to(dev)
        xs = xs.repeat(-(-per // xs.shape[0]), 1, 1)[:per]   # (per-rank share; distinct seeds for the first 64, repeated beyond)
        g512 = torch.empty((world * per, 2, CHUNK_FRAMES), dtype=torch.float32, device=dev) if use_dist else None

        ev_s = []   # (start, forward done, gather done) stream events per timed step

        def sstep(record=False):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
            if e:
                e[0].record()
            with torch.inference_mode(), torch.autocast("cuda", enabled=half):
                outs = [a2b.model(xs[i: i + 64]) for i in range(0, per, 64)]
                r = torch.stack((torch.cat([o["beat"] for o in outs]), torch.cat([o["downbeat"] for o in outs])), 1)
            if e:
                e[1].record()
            if use_dist:
                all_gather(g512, r)
            if e:
                e[2].record()
                ev_s.append(e)
            return r
        for _ in range(2):
            sstep()
        fence()
        ts = time.perf_counter()
        n_s = 3 if world == 1 else 6
        for _ in range(n_s):
            sstep(True)
        fence()
        ts = time.perf_counter() - ts
        # this rank's own forward time and what the collective added behind it (stream events: the gather's interval includes
        # the wait for the slowest rank, which is what a strong-scaling step pays)
        fwd_ms = sum(a.elapsed_time(b) for a, b, _ in ev_s) / n_s
        gat_ms = sum(b.elapsed_time(c) for _, b, c in ev_s) / n_s
        s_per_rank, s_gather = [round(fwd_ms, 3)], [round(gat_ms, 3)]
        if use_dist:
            t = torch.tensor([ts], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            ts = float(t.item())
            mine = torch.tensor([[fwd_ms, gat_ms]], dtype=torch.float64, device=dev)   # (1, 2) -> (world, 2): concatenation along dim 0
            allm = torch.empty((world, 2), dtype=torch.float64, device=dev)
            all_gather(allm, mine)
            s_per_rank = [round(float(v), 3) for v in allm[:, 0].tolist()]
            s_gather = [round(float(v), 3) for v in allm[:, 1].tolist()]
        ts /= n_s
        strong = {"workload": f"BASELINE config 4: 512 x 1500-frame chunks, {per} per GPU in slices of 64, {args.prec} forward, "
                              "logits all-gathered (RCCL)" + ("" if use_dist else " -- one GPU: no collective"),
                  "global_chunks": world * per, "chunks_per_gpu": per, "ms_per_step": round(ts * 1e3, 3),
                  "audio_seconds_per_s": round(world * per * FRESH_SECONDS_PER_CHUNK / ts, 1), "scaling": "strong",
                  "per_rank_ms": s_per_rank, "gather_ms": s_gather,
                  "per_rank_note": "stream events on every rank: per_rank_ms = its own forward slices, gather_ms = from the end of "
                                   "its forward to the end of the all-gather (includes waiting for the slowest rank)"}
        del xs, g512

    out = None
    if rank == 0:
        lib = _lib.lib()
        fl = flops_per_chunk(hp["transformer_dim"], ff_mult=hp["ff_mult"])
        FLOP_PER_CHUNK = sum(v for k, v in fl.items() if k != "layer_tail")  # 134.71 GFLOP for final0
        eng = a2b.model.engine()

        def profile_forward(run, n_prof, chunks):
            # (per-launch durations are taken with the forward slices on ONE stream: concurrent slices share the chip and
            # would stretch each other's event intervals)
            saved, _inf.CONCURRENT_STREAMS = _inf.CONCURRENT_STREAMS, 1
            run()
            eng.profiling = True   # (no graph captures while events are recorded)
            lib.bt_profile_begin(eng._h)
            for _ in range(n_prof):
                run()
            _inf.CONCURRENT_STREAMS = saved
            ncat = len(_lib.PROFILE_CATEGORIES)
            ms = (C.c_double * ncat)()
            cnt = (C.c_int32 * ncat)()
            _lib.check(lib.bt_profile_end(eng._h, ms, cnt, ncat))
            eng.profiling = False
            bd = {}
            for i, name in enumerate(_lib.PROFILE_CATEGORIES):
                if cnt[i]:
                    t_step = ms[i] / n_prof
                    bd[name] = {"ms_per_step": round(t_step, 4), "launches_per_step": cnt[i] // n_prof,
                                "tflops": round(fl[name] * chunks / (t_step * 1e-3) / 1e12, 2)}
            return bd

        # ---- roofline leg: per-kernel HIP-event timing of the same workload ------------------------------------------
        log("roofline leg")
        if args.workload == "tracks":
            breakdown = profile_forward(lambda: a2b.many(tracks, TRACK_SR), 2, chunks_per_step)
        else:
            breakdown = profile_forward(step, 3, chunks_per_step)
        def soak_of(scheme):
            """what the committed flip soak (tools/flip_soak.py: many 300 s tracks x 3 weight styles against the fp32 CPU oracle)
            says about `scheme`: flips per 1000 beat / downbeat decisions, next to the exact fp32 MFMA path's -- the rule the
            default arithmetic is chosen by -- with the sha256 of the table it is read from"""
            import glob
            import hashlib

            try:
                path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flip_frontier.json")))[-1]
                sj = json.load(open(path))
                txt = os.path.splitext(path)[0] + ".txt"
                sha = hashlib.sha256(open(txt, "rb").read()).hexdigest()[:16]

                def rate(name):
                    rows = [v for k, v in sj[name].items() if k != "rule"]
                    return round(1000.0 * sum(v["flips"] for v in rows) / max(1, sum(v["decisions"] for v in rows)), 4)
                return {"flips_per_1000": rate(scheme), "flips_per_1000_exact_fp32_path": rate("exact"),
                        "flips_per_1000_by_style": {k: v["flips_per_1000"] for k, v in sj[scheme].items() if k != "rule"},
                        "soak_rule": sj[scheme].get("rule"), "soak_scheme": scheme,
                        "soak_tracks_per_style": min(v["tracks"] for k, v in sj[scheme].items() if k != "rule"),
                        "soak_file": os.path.relpath(txt, ROOT), "soak_sha256_16": sha}
            except (OSError, IndexError, KeyError, ValueError) as e:
                return {"flips_per_1000": None, "soak_note": f"no committed soak summary for {scheme} ({type(e).__name__})"}

        def roofline_of(bd, prec, chunks, step_ms=None):
            """roofline object of a profiled forward: its dominant launch category against the guide's dense matrix peak of
            the MFMA `prec` issues (no self-derated peak)"""
            dom = max(bd, key=lambda k: bd[k]["ms_per_step"])
            d = bd[dom]
            peak = PEAK_TFLOPS[prec]
            per_product = MFMAS_PER_PRODUCT[prec]
            if prec == "f32x3" and dom == "attn_flash":
                # scores on three MFMAs per product; P.V on two where the probabilities enter as fp16 hi parts (BT_OPT_X3_ATTN_P16:
                # 2 = everywhere, 1 = main layers only = half of the attention flops, 0 = nowhere)
                per_product = {0: 3.0, 1: 2.75, 2: 2.5}[args.x3_p16]
            traffic, traffic_src = None, None
            try:  # HBM bytes per launch of the dominant category: PMC counters of separate rocprofv3 passes, committed
                name = "pmc_traffic.json" if prec == "half" else f"pmc_traffic_{prec}.json"
                tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                if tj["workload"]["model"] == args.model and tj["workload"]["prec"] == prec and dom in tj:
                    traffic = tj[dom]["bytes_per_launch"] * chunks / tj["workload"]["chunks"] / (
                        d["launches_per_step"] / tj[dom]["launches_per_forward"])
                    traffic_src = f"profiles/{name} ({tj['workload']['chunks']}-chunk forward, scaled per launch)"
            except (OSError, ValueError, KeyError, ZeroDivisionError):
                pass
            tot = sum(v["ms_per_step"] for v in bd.values())
            r = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": round(peak, 1), "unit": "TFLOP/s",
                 "frac": round(d["tflops"] / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                 "avg_launch_ms": round(d["ms_per_step"] / d["launches_per_step"], 4),
                 "flop_per_launch": fl[dom] * chunks / d["launches_per_step"],
                 "mfmas_per_product": per_product,
                 "pipe_utilisation": round(d["tflops"] * per_product / peak, 4),
                 "forward_ms_per_step": round(tot, 3),
                 "whole_forward_tflops": round(FLOP_PER_CHUNK * chunks / (tot * 1e-3) / 1e12, 2),
                 "whole_forward_frac": round(FLOP_PER_CHUNK * chunks / (tot * 1e-3) / 1e12 / peak, 4),
                 "peak_note": {"half": "dense fp16 MFMA peak (MI355X_MICROARCH.md)", "f32": "fp32 MFMA peak (MI355X_MICROARCH.md)",
                               "f32x3": "dense fp16 MFMA peak (MI355X_MICROARCH.md); the path issues mfmas_per_product fp16 MFMAs "
                                        "per algorithmic product, pipe_utilisation = frac x that"}[prec]}
            if step_ms is not None:   # the timed region itself (two streams, steps pipelined): SURVEY 8d flops / ms_per_step / peak
                r["whole_step_tflops"] = round(FLOP_PER_CHUNK * chunks / (step_ms * 1e-3) / 1e12, 2)
                r["whole_step_frac"] = round(FLOP_PER_CHUNK * chunks / (step_ms * 1e-3) / 1e12 / peak, 4)
            return r

        roofline = roofline_of(breakdown, args.prec, chunks_per_step, ms_per_step)

        frontend = forward_only = host_inclusive = configs = None
        if args.workload == "tracks" and not args.no_extras:
            # ---- HBM-bound stages: events around each stage on torch's current stream (the launch stream) -------------
            def timed(fn, reps=5):
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    r = fn()
                b.record()
                b.synchronize()
                return a.elapsed_time(b) / reps, r

            log("frontend / forward-only legs")
            n_samp = sum(int(t.shape[0]) for t in tracks)
            t_front, (spect, foff) = timed(lambda: a2b.signal2spect_many(tracks, TRACK_SR))
            tr22 = [torch.from_numpy(W.synthetic_audio(TRACK_SECONDS, seed=i)).to(dev) for i in range(n_tr)]
            t_mel, _ = timed(lambda: a2b.signal2spect_many(tr22, 22050))
            n22 = sum(int(t.shape[0]) for t in tr22)
            beat, down = a2b.spect2frames_batch(spect, foff)
            t_post, _ = timed(lambda: a2b.frames2beats.ragged(beat, down, foff))
            mel_bytes = 4 * n22 + 4 * 128 * int(foff[-1])
            res_bytes = 4 * n_samp + 4 * n22
            frontend = {
                "resample+logmel_ms": round(t_front, 4), "logmel_ms": round(t_mel, 4),
                "logmel_GBps_algorithmic": round(mel_bytes / (t_mel * 1e-3) / 1e9, 1),
                "logmel_frac_of_hbm_peak": round(mel_bytes / (t_mel * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "resample_ms": round(max(t_front - t_mel, 0.0), 4),
                "resample_GBps_algorithmic": round(res_bytes / (max(t_front - t_mel, 1e-6) * 1e-3) / 1e9, 1),
                "peaks+d2h+host_post_ms": round(t_post, 4),
                "note": "python launch overhead included (stage wall time between stream events); bytes = SURVEY 8d algorithmic"}
            # ---- BASELINE configs 2 / 3 / 4 (share): forward only, chunks resident ---------------------------------------
            def time_forward(model_, xin, prec_, n_warm, n_time):
                keep = model_.fp32_split_gemms
                model_.fp32_split_gemms = prec_ == "f32x3"

                def f():
                    with torch.inference_mode(), torch.autocast("cuda", enabled=prec_ == "half"):
                        return model_(xin)
                for _ in range(n_warm):
                    f()
                torch.cuda.synchronize(dev)
                t_ = time.perf_counter()
                for _ in range(n_time):
                    f()
                torch.cuda.synchronize(dev)
                model_.fp32_split_gemms = keep
                return (time.perf_counter() - t_) / n_time

            def fwd_obj(t_, chunks):
                return {"ms_per_step": round(t_ * 1e3, 3), "audio_seconds_per_s": round(chunks * FRESH_SECONDS_PER_CHUNK / t_, 1),
                        "whole_forward_tflops": round(FLOP_PER_CHUNK * chunks / t_ / 1e12, 1)}

            other = "half" if args.prec != "half" else "f32x3"
            x16 = torch.from_numpy(np.stack([W.synthetic_spect(CHUNK_FRAMES, seed=1000 + i) for i in range(16)])).to(dev)
            forward_only = {"workload": "BASELINE config 2: 16 chunks x 1500 frames, BeatThis.forward, spectrograms resident",
                            **fwd_obj(time_forward(a2b.model, x16, args.prec, 20, 30), 16), "dtype": args.prec,
                            other: fwd_obj(time_forward(a2b.model, x16, other, 20, 30), 16)}

            log("config legs (cfg3, cfg4 share)")
            x64 = torch.from_numpy(np.stack([W.synthetic_spect(CHUNK_FRAMES, seed=2000 + i) for i in range(64)])).to(dev)
            t64 = time_forward(a2b.model, x64, args.prec, 3, 5)
            t64o = time_forward(a2b.model, x64, other, 3, 5)
            hp_s = W.resolve_hparams("small0")
            m_s = BeatThis(**{k: hp_s[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
            m_s.load_state_dict(W.random_state_dict(hp_s, seed=1, style="lively"))
            m_s = m_s.to(dev)
            x128 = x64.repeat(2, 1, 1)
            t128 = time_forward(m_s, x128, "f32", 2, 3)
            t128x = time_forward(m_s, x128, "f32x3", 2, 3)
            fl_s = flops_per_chunk(hp_s["transformer_dim"], ff_mult=hp_s["ff_mult"])
            flop_s = sum(v for k, v in fl_s.items() if k != "layer_tail")   # 59.57 GFLOP / chunk
            cfg3 = {"workload": "BASELINE config 3: small0, exact fp32 MFMA, 128 chunks x 1500 frames resident",
                    "ms_per_step": round(t128 * 1e3, 2), "audio_seconds_per_s": round(128 * FRESH_SECONDS_PER_CHUNK / t128, 1),
                    "tflops_fp32": round(flop_s * 128 / t128 / 1e12, 1),
                    "frac_of_fp32_matrix_peak": round(flop_s * 128 / t128 / 1e12 / PEAK_TFLOPS["f32"], 4),
                    "hbm_GBps_counters": None, "frac_of_hbm_peak": None,
                    "f32x3": {"ms_per_step": round(t128x * 1e3, 2), "audio_seconds_per_s": round(128 * FRESH_SECONDS_PER_CHUNK / t128x, 1),
                              "note": "the same batch in the API's default precision (fp32-class results on hi + lo fp16 operands)"}}
            try:  # HBM bytes of one such forward from the committed counter passes (FETCH_SIZE x 2 + WRITE_SIZE, separate runs)
                tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_cfg3.json")))
                gb = tj["bytes_per_forward"] / 1e9
                cfg3.update(hbm_GB_per_forward=round(gb, 2), hbm_GBps_counters=round(gb / t128, 1),
                            frac_of_hbm_peak=round(gb / t128 / PEAK_HBM_GBS, 4), hbm_source="profiles/pmc_traffic_cfg3.json "
                            "(rocprofv3 PMC passes of this configuration; bytes per forward / this run's time)")
            except (OSError, ValueError, KeyError):
                pass
            # ---- BASELINE config 5: the 64-chunk batch with the cross terms of the hi + lo products on the block-scaled fp8 MFMA
            # (opt-in BT_OPT_X3_GEMM_FP8; the default path for comparison is t64 above when --prec f32x3)
            cfg5 = None
            if args.prec == "f32x3":
                eng = a2b.model.engine()
                lv = {}
                with torch.inference_mode():
                    ref64 = a2b.model(x64)
                    ref64 = torch.stack([ref64["beat"].float(), ref64["downbeat"].float()])
                for level in (1, 2):
                    eng.set_options({"x3_gemm_fp8": level})
                    fb0 = eng.last_fallbacks
                    t5 = time_forward(a2b.model, x64, "f32x3", 3, 5)
                    with torch.inference_mode():
                        o5 = a2b.model(x64)
                    d5 = float((torch.stack([o5["beat"].float(), o5["downbeat"].float()]) - ref64).abs().max())
                    lv[f"level{level}"] = {**fwd_obj(t5, 64), "max_abs_logit_vs_default_path": d5, "range_fallbacks": int(eng.last_fallbacks - fb0)}
                eng.set_options({"x3_gemm_fp8": 0})
                cfg5 = {"workload": "BASELINE config 5: final0, 64 chunks x 1500 frames resident; hi . hi on fp16 MFMAs, both cross terms "
                                    "(hi . lo + lo . hi) of every product on ONE v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 bytes of the value "
                                    "and of its lo part, the power of two in the E8M0 scale) -- level 1: feed-forward GEMMs, level 2: QKV and "
                                    "out-projection as well",
                        "default_path": fwd_obj(t64, 64), **lv, "default": "off (opt-in Engine.set_options / bt_engine_set_option)",
                        "parity": "tests/test_gpu_scale.py::test_cfg5_final0_fp8_cross_terms_vs_oracle (logits within 3e-4 of the fp32 "
                                  "oracle, same beats); flip rates over 48 tracks x 3 weight styles: profiles/r05_flip_frontier.txt "
                                  "(x3p16f8ff / x3p16f8)"}
            # ... and config 5 AS WRITTEN (MX e4m3 operands), report-only: the GEMM itself measured here at the 64-chunk batch's shape next
            # to the fp16 GEMM (tools/mx8_probe.py), the arithmetic's price from the committed simulation on the oracle
            if cfg5 is not None:
                try:
                    from tools.mx8_probe import measure as mx8_measure

                    mx = mx8_measure(chunks=64, seconds=0.15, dev=dev, energy=False, with_hl32=False)
                    t16_, t8_ = sum(r["fp16_us"] for r in mx), sum(r["mx8_us"] for r in mx)
                    cfg5["as_written_mx_e4m3_operands"] = {
                        "status": "report-only, no forward built on it: closed with numbers (profiles/r06_cfg5_mx8.txt)",
                        "gemm_us_one_main_layer_fp16": round(t16_, 1), "gemm_us_one_main_layer_mx_e4m3": round(t8_, 1),
                        "gemm_speedup_upper_bound": round(t16_ / t8_, 3),
                        "projected_forward_speedup_over_fp16_path": None,   # (filled in below from the fp16 path's own breakdown)
                        "keep_bar": 1.25,
                        "simulated_max_abs_logit_vs_fp32_oracle": {"lively": 0.465, "outlier": 0.412, "init": 0.093},
                        "simulated_flips_per_1000": {"lively": 341.3, "outlier": 227.9, "init": 1310.4},
                        "fp16_path_same_simulation": {"max_abs_logit": {"lively": 0.006, "outlier": 0.0088, "init": 0.0011},
                                                      "flips_per_1000": {"lively": 5.9, "outlier": 5.8, "init": 30.6}},
                        "note": "GEMM speedup = bt_gemm_mx8 (fp32 results, no quantising epilogue) against bt_gemm3 fp16 with a residual-type "
                                "epilogue, the four GEMM shapes of a main layer at 96000 rows, measured in this run; projection = only those "
                                "GEMMs get faster; error / flips: tools/flip_soak.py sim, 8 tracks x 3 styles (committed table)"}
                except Exception as e:  # noqa: BLE001
                    cfg5["as_written_mx_e4m3_operands"] = {"status": f"not measured in this run ({type(e).__name__}: {e})"}
            configs = {
                "cfg2": "= forward_only",
                "cfg3": cfg3,
                "cfg4_share": {"workload": "BASELINE config 4, one GPU's share: final0, 64 chunks x 1500 frames resident",
                               **fwd_obj(t64, 64), "dtype": args.prec, other: fwd_obj(t64o, 64)},
                "cfg5": cfg5 or "measured with --prec f32x3 only"}
            del m_s, x64, x128

            # ---- the same job from HOST buffers (PCIe-inclusive; never `value`): waveforms in pinned host memory, uploaded
            # on a copy stream by Audio2Beats.many_async while the previous step computes -----------------------------------
            log("host-inclusive leg")
            htracks = [t.cpu().pin_memory() for t in tracks]
            hpend = []

            def hstep():
                hpend.append(a2b.many_async(htracks, TRACK_SR))
                if len(hpend) > 1:
                    hpend.pop(0).result()
            for _ in range(3):
                hstep()
            while hpend:
                hpend.pop(0).result()
            torch.cuda.synchronize(dev)
            th = time.perf_counter()
            for _ in range(10):
                hstep()
            while hpend:
                hpend.pop(0).result()
            torch.cuda.synchronize(dev)
            th = (time.perf_counter() - th) / 10
            host_inclusive = {"ms_per_step": round(th * 1e3, 3), "audio_seconds_per_s": round(n_tr * TRACK_SECONDS / th, 1),
                              "upload_MB_per_step": round(4 * n_samp / 1e6, 1),
                              "note": "waveforms start in pinned host memory; H2D on a copy stream overlaps the previous step"}
            del htracks

        # ---- CPU baseline + in-run parity: the oracle's Audio2Beats on track 0, bounded sample ------------------------
        cpu = parity = half_path = fp32_exact_path = f32x3_path = latency = stress = None
        p16_legs = {}
        if world == 1 and not args.no_cpu_baseline and args.workload == "tracks":
            from oracle import beat_this_oracle as O

            log("cpu baseline: thread probe")
            sig = tracks[0].cpu().numpy()
            with torch.inference_mode():
                # torch's default (one thread per logical core) oversubscribes big hosts badly: probe on one chunk
                xc = torch.from_numpy(W.synthetic_spect(CHUNK_FRAMES, seed=5))[None]
                best, probe = None, {}
                for nt in (8, 16, 32, 64):  # (one thread per logical core -- torch's default -- is 10x slower on a 2 x 64-core host)
                    if nt > (os.cpu_count() or 1):
                        continue
                    torch.set_num_threads(nt)
                    O.model_forward(sd, xc)
                    tp = time.perf_counter()
                    O.model_forward(sd, xc)
                    tp = time.perf_counter() - tp
                    probe[nt] = round(tp * 1e3)
                    if best is None or tp < best[1]:
                        best = (nt, tp)
                torch.set_num_threads(best[0])
                # bounded sample: the whole 300 s track if its 11 chunks fit ~12 s of CPU time (three repeats follow), else a
                # shorter excerpt
                sample_s = TRACK_SECONDS if 11 * best[1] < 12.0 else max(30.0, 29.76 * int(12.0 / best[1]))
                sig = sig[: int(sample_s * TRACK_SR)]
                log(f"cpu baseline: {best[0]} threads, {best[1] * 1e3:.0f} ms per chunk; 3 x Audio2Beats of {sample_s:.0f} s of track 0")
                tcs = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    ob, od = O.audio2frames(sd, sig, TRACK_SR)
                    obeats, odown = O.postp_minimal(ob, od)
                    tcs.append(time.perf_counter() - t1)
            tc = min(tcs)
            cpu = {"value": round(sample_s / tc, 2), "unit": "audio-seconds/s", "cores": best[0],
                   "median": round(sample_s / sorted(tcs)[1], 2), "repeats_s": [round(t, 2) for t in tcs],
                   "host_logical_cores": os.cpu_count(), "kind": "port",
                   "threads_probe_ms_per_chunk": probe,
                   "sample": f"3 x Audio2Beats of {sample_s:.0f} s of one {TRACK_SR} Hz track (resample, log-mel, {len(ob) // 1488 + 1} chunks "
                             f"batch-1 like the reference, SDPA attention, fp32, post-processing), oracle/beat_this_oracle.py, value = "
                             f"the fastest repeat; thread count = fastest of 8/16/32/64 on one chunk (more threads are slower on "
                             f"this host: threads_probe_ms_per_chunk), the reference's own default would be one per logical core"}
            try:   # how the port compares with the code it stands in for (measured in the build container, where the reference
                # exists: tools/port_speed.py -- alternating runs of one chunk, fastest of each)
                pv = json.load(open(os.path.join(ROOT, "profiles", "r05_port_vs_reference.json")))
                cpu["port_vs_reference"] = {k: pv[k] for k in ("port_vs_reference", "port_ms_per_chunk", "reference_ms_per_chunk",
                                                               "threads", "host_cores", "model")}
                cpu["port_vs_reference"]["note"] = "oracle.model_forward / the unmodified reference's BeatThis.forward, build container (tools/port_speed.py)"
            except (OSError, KeyError, ValueError):
                pass
            ptrack = [torch.from_numpy(sig).to(dev)]

            def flips(a, b):
                return len(set(np.round(np.asarray(a) * 100).astype(np.int64)) ^ set(np.round(np.asarray(b) * 100).astype(np.int64)))

            def parity_of(a2b_):
                res = a2b_.many_async(ptrack, TRACK_SR)
                beats, downbeats = res.result()[0]
                gb, gd, _ = res.logits
                return {"max_abs_logit": round(max(float((gb.cpu() - ob).abs().max()), float((gd.cpu() - od).abs().max())), 6),
                        "flips_beat": flips(beats, obeats), "flips_downbeat": flips(downbeats, odown),
                        "n_beats": len(obeats), "n_downbeats": len(odown), "logit_spread": round(float(ob.std()), 3),
                        "against": f"CPU oracle (fp32) on the same {sample_s:.0f} s waveform, {len(ob)} frames"}
            log(f"cpu baseline done ({tc:.1f} s fastest); parity")
            parity = parity_of(a2b)
            parity.update(soak_of({0: "x3", 1: "x3p16m", 2: "x3p16"}[args.x3_p16] if args.prec == "f32x3" else
                                  {"half": "half", "f32": "exact"}[args.prec]))

            def path_leg(prec, seconds):
                """the headline workload in another precision: ~`seconds` of pipelined steps, parity, per-launch profile"""
                set_prec(prec)
                for _ in range(3):
                    a2b.many(tracks, TRACK_SR)
                torch.cuda.synchronize(dev)
                est = ms_per_step * {"half": 0.5, "f32": 2.7, "f32x3": 1.0}[prec] / {"half": 0.5, "f32": 2.7, "f32x3": 1.0}[args.prec]
                n = max(3, int(seconds / max(est * 1e-3, 1e-3)))
                fb0 = eng.last_fallbacks
                t_ = time.perf_counter()
                for _ in range(n):
                    step()
                drain()
                torch.cuda.synchronize(dev)
                t_ = (time.perf_counter() - t_) / n
                bd = profile_forward(lambda: a2b.many(tracks, TRACK_SR), 2, chunks_per_step)
                leg = {"ms_per_step": round(t_ * 1e3, 2), "audio_seconds_per_s": round(units_per_step / t_, 1), "steps": n,
                       "dtype": DTYPE[prec], "parity": parity_of(a2b), "roofline": roofline_of(bd, prec, chunks_per_step),
                       "breakdown": bd}
                if prec == "f32x3":
                    leg["range_fallbacks"] = eng.last_fallbacks - fb0
                set_prec(args.prec)
                return leg

            if not args.no_extras:
                if args.prec != "half":
                    log("half path leg")
                    half_path = path_leg("half", 1.0)
                    try:  # the reference's OWN float16-autocast error against its fp32 forward (generated from the unmodified
                        # reference on the final0 golden case, oracle/make_golden.py): the yardstick for this path
                        rep = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_autocast_report.json")))["final0_lively_T1500_f16"]
                        half_path["reference_fp16_autocast"] = {
                            "max_abs_logit": round(max(rep["max_abs_beat"], rep["max_abs_downbeat"]), 6),
                            "flips_beat": rep["flips_beat"], "flips_downbeat": rep["flips_downbeat"], "n_beats": rep["n_beats_fp32"],
                            "note": "unmodified reference, torch.autocast(float16) vs its own fp32 forward, one 1500-frame chunk of "
                                    "the same weights (tests/golden/reference_autocast_report.json)"}
                    except (OSError, KeyError, ValueError):
                        pass
                    half_path["note"] = "float16=True: NOT under the 1e-3 / identical-beats gate (see parity); never `value`"
                    try:   # config 5 as written: only the main layers' GEMMs of the fp16 path get faster, by the measured GEMM ratio
                        aw = configs["cfg5"]["as_written_mx_e4m3_operands"]
                        hb = half_path["breakdown"]
                        tot = sum(v["ms_per_step"] for v in hb.values())
                        # (main layers' GEMM time of the fp16 step: the fused layer tail + the six gemm3 launches of the nine QKV launches,
                        # 72 % of that category's time by profiles/r06_kernel_trace_fwd.txt)
                        gemm = hb.get("layer_tail", {"ms_per_step": 0})["ms_per_step"] + 0.72 * hb["qkv_gemm"]["ms_per_step"] + sum(
                            hb.get(k, {"ms_per_step": 0})["ms_per_step"] for k in ("out_gemm", "ff1_gemm", "ff2_gemm"))
                        aw["main_layer_gemm_share_of_fp16_step"] = round(gemm / tot, 3)
                        aw["projected_forward_speedup_over_fp16_path"] = round(tot / (tot - gemm * (1 - 1 / aw["gemm_speedup_upper_bound"])), 3)
                    except (KeyError, TypeError, ZeroDivisionError):
                        pass
                if args.prec != "f32":
                    log("exact fp32 path leg")
                    fp32_exact_path = path_leg("f32", 0.4)
                if args.prec != "f32x3" and not _lib.lib().bt_half_is_bf16():
                    log("f32x3 path leg")
                    f32x3_path = path_leg("f32x3", 1.0)

                # ---- the other arithmetic levels of the default precision (BT_OPT_X3_ATTN_P16), same workload, ~0.7 s each -----
                if args.prec == "f32x3":
                    log("P16 level legs")
                    for level in (0, 1, 2):
                        if level == args.x3_p16:
                            continue
                        eng.set_options({"x3_attn_p16": level})
                        for _ in range(3):
                            step()
                        drain()
                        torch.cuda.synchronize(dev)
                        n = max(3, int(0.7 / max(ms_per_step * 1e-3, 1e-3)))
                        t_ = time.perf_counter()
                        for _ in range(n):
                            step()
                        drain()
                        torch.cuda.synchronize(dev)
                        t_ = (time.perf_counter() - t_) / n
                        p16_legs[f"value_p16_{level}"] = {"value": round(units_per_step / t_, 1), "ms_per_step": round(t_ * 1e3, 3), "steps": n,
                                                          "parity": {**parity_of(a2b), **soak_of({0: "x3", 1: "x3p16m", 2: "x3p16"}[level])}}
                    eng.set_options({"x3_attn_p16": args.x3_p16})

                # ---- single-file latency (BASELINE config 1): one call at a time, host waveform in, beat times out ---------
                log("latency leg")
                sig30 = W.synthetic_audio(30.0, seed=7, sr=TRACK_SR)
                sig300 = tracks[0].cpu().numpy()
                latency = {"workload": "Audio2Beats.__call__ (= File2Beats minus the audio decoder) on ONE track in host memory: mono "
                                       "float32 44.1 kHz -> H2D, resample, log-mel, 2 / 11 chunks through the model, aggregation, peak "
                                       "picking, D2H, host post-processing -> beat times; one call at a time, median of 10"}
                for prec in ("f32x3", "half", "f32"):
                    set_prec(prec)
                    row = {}
                    for tag, sg in (("30s", sig30), ("300s", sig300)):
                        for _ in range(3):
                            a2b(sg, TRACK_SR)
                        ts = []
                        for _ in range(10):
                            t_ = time.perf_counter()
                            a2b(sg, TRACK_SR)
                            ts.append(time.perf_counter() - t_)
                        ts.sort()
                        row[tag] = {"ms_median": round(ts[5] * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3),
                                    "audio_seconds_per_s": round(len(sg) / TRACK_SR / ts[5], 1)}
                    latency[prec] = row
                set_prec(args.prec)
                with torch.inference_mode():
                    t_ = time.perf_counter()
                    b30, d30 = O.audio2frames(sd, sig30, TRACK_SR)
                    O.postp_minimal(b30, d30)
                    t30 = time.perf_counter() - t_
                latency["cpu_oracle"] = {"30s": {"ms": round(t30 * 1e3, 1), "audio_seconds_per_s": round(30.0 / t30, 1)},
                                         "300s": {"ms": round(tc * 1e3 * TRACK_SECONDS / sample_s, 1), "audio_seconds_per_s": cpu["value"]},
                                         "threads": best[0]}

                # ---- trained-like stress weights: does the default path stay on its fast route? ---------------------------
                if not _lib.lib().bt_half_is_bf16():
                    log("stress-weights leg")
                    sd_o = W.random_state_dict(hp, seed=1, style="outlier")
                    m_o = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
                    m_o.load_state_dict(sd_o)
                    a2b_o = Audio2Beats(checkpoint_path=None, device=dev, float16=False, dbn=False)
                    a2b_o.model = m_o.to(dev)
                    eng_o = a2b_o.model.engine()
                    for _ in range(2):
                        a2b_o.many(tracks, TRACK_SR)
                    torch.cuda.synchronize(dev)
                    fb0, pend_o = eng_o.last_fallbacks, []
                    n_o = 8
                    t_ = time.perf_counter()
                    for _ in range(n_o):
                        pend_o.append(a2b_o.many_async(tracks, TRACK_SR))
                        if len(pend_o) > 1:
                            pend_o.pop(0).result()
                    while pend_o:
                        pend_o.pop(0).result()
                    torch.cuda.synchronize(dev)
                    t_ = (time.perf_counter() - t_) / n_o
                    fallbacks = eng_o.last_fallbacks - fb0
                    sig_o = sig[: int(60.0 * TRACK_SR)]
                    with torch.inference_mode():
                        ob_o, od_o = O.audio2frames(sd_o, sig_o, TRACK_SR)
                    obt, odt = O.postp_minimal(ob_o, od_o)
                    res = a2b_o.many_async([torch.from_numpy(sig_o).to(dev)], TRACK_SR)
                    bt_o, dt_o = res.result()[0]
                    stress = {"weights": "style='outlier' (beat_this_amd/weights.py): residual outlier channels of ~10^3, heavy-tailed "
                                         "(Student-t) matrices, sharper attention, frontend activations of ~10^2",
                              "ms_per_step": round(t_ * 1e3, 2), "audio_seconds_per_s": round(units_per_step / t_, 1), "steps": n_o,
                              "range_fallbacks": fallbacks, "batches": n_o,
                              "parity": {"max_abs_logit": round(max(float((res.logits[0].cpu() - ob_o).abs().max()),
                                                                    float((res.logits[1].cpu() - od_o).abs().max())), 6),
                                         "flips_beat": flips(bt_o, obt), "flips_downbeat": flips(dt_o, odt), "n_beats": len(obt),
                                         "n_downbeats": len(odt), "logit_spread": round(float(ob_o.std()), 3),
                                         "against": "CPU oracle (fp32) on the first 60 s of track 0"}}
                    del a2b_o, m_o

        out = {
            "metric": "audio-seconds processed/sec", "value": round(value, 1), "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"half": half_desc, "f32": "f32", "f32x3": "f32 activations, f16x3 products (3 x f16 MFMA on hi+lo operand halves, f32 accumulate)" + {0: "", 1: "; main-layer attention P.V: 2 x, probabilities as fp16 hi parts", 2: "; attention P.V: 2 x, probabilities as fp16 hi parts (opt-in level 2)"}[args.x3_p16]}[args.prec],
            "data": "synthetic",
            "config": {"workload": workload, "tracks_per_gpu": args.tracks if args.workload == "tracks" else None,
                       "chunks_per_gpu": chunks_per_step, "global_chunks": world * chunks_per_step,
                       "parallelism": f"track-sharded x{world}, framewise logits all-gathered" if args.workload == "tracks"
                       else f"chunk-sharded x{world}, logits all-gathered"},
            "parity": parity, "roofline": roofline, "cpu_baseline": cpu, "energy": energy, "frontend": frontend, "forward_only": forward_only,
            "half_path": half_path, "fp32_exact_path": fp32_exact_path, "latency": latency,
            "stress_weights": stress, "host_inclusive": host_inclusive, "configs": configs,
            "strong_scaling_cfg4": strong, "rccl_ranks": rccl_ranks,
            "rccl_note": "--share-gpu: all ranks on cuda:0, gloo collectives (development run of the N > 1 path, not a measurement)" if args.share_gpu else dist_note,
            "timed_region": {"steps": args.steps, "repeats": repeats, "steps_timed": n_timed, "seconds": round(elapsed, 3),
                             "per_rank_ms_per_step": per_rank_ms},
            "breakdown": breakdown, **p16_legs,
        }
        if f32x3_path is not None:   # (only when the headline is another precision: --prec half / f32)
            out["f32x3_path"] = f32x3_path
        if last is not None and args.workload == "tracks":
            out["config"]["beats_in_last_track"] = int(len(last[-1][0]))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _guarded_main():
    """main() with the N > 1 failure rule: a rank that raises must not leave the others inside a collective until a watchdog
    fires.  The traceback goes to stderr, then the process leaves with os._exit (no interpreter teardown: the RCCL process
    group's destructor would wait for the collectives the dead step never issued) -- torch.distributed.run sees a failed
    worker and terminates the rest of the group at once."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001
        import traceback

        traceback.print_exc()
        sys.stderr.write(f"[bench] rank {os.environ.get('RANK', '0')} failed: aborting the job\n")
        sys.stderr.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os._exit(1)
        raise


if __name__ == "__main__":
    _guarded_main()

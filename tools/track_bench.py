"""Track-level end-to-end timing on the GPU box: `Audio2Beats` (upload, resample 44.1k -> 22.05k, log-mel, chunking,
forward, aggregation, peak picking, host post-processing) on synthetic 44.1 kHz tracks of a stated length --
the use case BASELINE.json's metric is named after.  Prints one JSON object; not the bench contract (bench.py
measures the HBM-resident model path), this is the PCIe- and host-inclusive figure quoted in DESIGN.md.

    python tools/track_bench.py [--seconds 300] [--tracks 8] [--prec bf16|f32] [--model final0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from beat_this_amd import weights as W  # noqa: E402
from beat_this_amd.inference import Audio2Beats  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--tracks", type=int, default=8)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="final0")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    hp = W.HPARAMS[args.model]
    ckpt = {"state_dict": {"model." + k: v for k, v in W.random_state_dict(hp, seed=0, style="init").items()},
            "hyper_parameters": dict(hp)}
    a2b = Audio2Beats(ckpt, dev, float16=args.prec == "bf16")
    tracks = [W.synthetic_audio(args.seconds, sr=44100, seed=i) for i in range(args.tracks)]

    def stage_times(sig):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t0 = time.perf_counter()
        ev[0].record()
        spect = a2b.signal2spect(sig, 44100)
        ev[1].record()
        bl, dl = a2b.spect2frames(spect)
        ev[2].record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        beats, downs = a2b.frames2beats(bl, dl)
        t2 = time.perf_counter()
        return {"gpu_frontend_ms": ev[0].elapsed_time(ev[1]), "gpu_model_ms": ev[1].elapsed_time(ev[2]),
                "host_to_logits_ms": 1e3 * (t1 - t0), "postprocess_ms": 1e3 * (t2 - t1), "frames": int(spect.shape[0]),
                "beats": int(len(beats)), "downbeats": int(len(downs))}

    for s in tracks[:2]:
        a2b(s, 44100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in tracks:
        a2b(s, 44100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = stage_times(tracks[0])
    out = {"workload": "%d x %.0f s mono 44.1 kHz float32 numpy tracks through Audio2Beats(%s, %s), one after another"
                       % (args.tracks, args.seconds, args.model, args.prec),
           "audio_seconds_per_s": round(args.tracks * args.seconds / dt, 1), "ms_per_track": round(1e3 * dt / args.tracks, 3),
           "stages_one_track": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

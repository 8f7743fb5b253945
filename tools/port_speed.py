"""How fast is the CPU oracle (the "port" bench.py times as cpu_baseline) relative to the code it stands in for -- the
unmodified reference's BeatThis.forward, imported with the three third-party stand-ins of oracle/shims?  (VERDICT r4 item 8:
494 vs 357 ms per final0 chunk then.)  Build container only (/root/reference); writes profiles/r05_port_vs_reference.json,
which bench.py quotes on its line.  Development tool: imports oracle/ and the reference.

    python tools/port_speed.py [final0|small0] [threads] [repeats]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def measure(model="final0", threads=8, repeats=5):
    import torch

    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), REFERENCE]
    try:
        from beat_this.model.beat_tracker import BeatThis
    finally:
        del sys.path[:2]
    from beat_this_amd import weights as W
    from oracle import beat_this_oracle as O

    torch.set_num_threads(threads)
    hp = W.resolve_hparams(model)
    sd = W.random_state_dict(hp, seed=1, style="lively")
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")}).eval()
    m.load_state_dict(sd)
    x = torch.from_numpy(W.synthetic_spect(1500, seed=5))[None]
    t_ref, t_port = [], []
    with torch.inference_mode():
        r = m(x)
        b, d = O.model_forward(sd, x)
        err = float(max((r["beat"] - b).abs().max(), (r["downbeat"] - d).abs().max()))
        for _ in range(repeats):   # alternating, fastest of each: robust against other load on the host
            t = time.perf_counter()
            m(x)
            t_ref.append(time.perf_counter() - t)
            t = time.perf_counter()
            O.model_forward(sd, x)
            t_port.append(time.perf_counter() - t)
    return {"model": model, "threads": threads, "repeats": repeats, "reference_ms_per_chunk": round(min(t_ref) * 1e3, 1),
            "port_ms_per_chunk": round(min(t_port) * 1e3, 1), "port_vs_reference": round(min(t_port) / min(t_ref), 3),
            "max_abs_logit_difference": err, "host_cores": os.cpu_count(),
            "what": "oracle.model_forward vs the unmodified reference's BeatThis.forward (oracle/shims for its three absent "
                    "third-party leaves) on one 1500-frame chunk, same weights, same input, alternating, fastest of each"}


if __name__ == "__main__":
    out = measure(sys.argv[1] if len(sys.argv) > 1 else "final0", int(sys.argv[2]) if len(sys.argv) > 2 else 8,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    print(json.dumps(out))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_port_vs_reference.json"), "w"), indent=1)

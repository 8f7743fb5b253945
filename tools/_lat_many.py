import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from beat_this_amd import weights as W
from beat_this_amd.inference import Audio2Beats
from beat_this_amd.model import BeatThis
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
a2b = Audio2Beats(checkpoint_path=None, device=dev, float16=False)
m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
m.load_state_dict(W.random_state_dict(hp, seed=1, style="lively"))
a2b.model = m.to(dev)
for secs in (30.0, 300.0):
    sig = W.synthetic_audio(secs, seed=7, sr=44100)
    for name, fn in (("__call__", lambda: a2b(sig, 44100)), ("many([x])", lambda: a2b.many([sig], 44100)[0])):
        for _ in range(5): out = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{secs:.0f} s {name:10s}: median {ts[15] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms, {len(out[0])} beats")
    a, b = a2b(sig, 44100), a2b.many([sig], 44100)[0]
    print("  identical:", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))

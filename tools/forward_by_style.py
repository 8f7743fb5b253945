"""Development: wall time of a 33-chunk forward of the default path on one weight style (lively | outlier | init)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beat_this_amd import weights as W
from beat_this_amd.model import BeatThis
style = sys.argv[1] if len(sys.argv) > 1 else "outlier"
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
m.load_state_dict(W.random_state_dict(hp, seed=1, style=style))
m = m.to(dev)
x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=1000 + i) for i in range(33)])).to(dev)
with torch.inference_mode():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize()
print(f"{style}: 33-chunk forward {(time.perf_counter() - t) / 10 * 1e3:.3f} ms, fallbacks {m.engine().last_fallbacks}")

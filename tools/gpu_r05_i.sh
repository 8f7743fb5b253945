#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu --tb=short -k "cfg5 or cfg2_final0_16_chunks_f32x3_vs" 2>&1 | tail -12
grep "cfg5\|cfg2_f32x3\"" gpurun_out/test_report.jsonl | tail -4
python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from beat_this_amd import weights as W
from beat_this_amd.model import BeatThis
dev = torch.device("cuda:0")
hp = W.resolve_hparams("final0")
m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
m.load_state_dict(W.random_state_dict(hp, seed=1, style="lively"))
m = m.to(dev)
x = torch.from_numpy(np.stack([W.synthetic_spect(1500, seed=1000 + i) for i in range(33)])).to(dev)
for rep in range(2):
    for lvl in (0, 1):
        m.engine().set_options({"x3_gemm_fp8": lvl})
        with torch.inference_mode():
            for _ in range(3): m(x)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): m(x)
            torch.cuda.synchronize()
        print(f"33-chunk forward, x3_gemm_fp8 = {lvl}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
PY

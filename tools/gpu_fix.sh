#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/fix
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short -k "attention" -x 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu --tb=short 2>&1 | tail -12
for st in lively outlier; do BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_dev.so timeout 300 python tools/safe_rate_probe.py 16 $st 2>&1 | grep -v amdgpu.ids; done | tee $O/rate.txt
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], (d.get("energy") or {}).get("joules_per_step"))
print("attn", d["breakdown"]["attn_flash"])
PY
timeout 900 python bench.py 2>/dev/null > $O/bench_full.json
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], json.dumps(d["parity"]))
print("stress", json.dumps(d["stress_weights"])[:600])
PY
bash tools/gpu_fix2.sh

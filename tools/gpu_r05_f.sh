#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu --tb=short -k "attention or model or scale" 2>&1 | tail -25 > $O/tests.txt
cat $O/tests.txt
echo "== re-run rate (BT_DEV build)"
for st in lively outlier; do BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_dev.so timeout 300 python tools/safe_rate_probe.py 16 $st 2>&1 | grep -v amdgpu.ids; done | tee $O/safe_rate.txt
echo "== bench (default) + stress leg"
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "energy", json.dumps(d["energy"]))
print("stress", json.dumps({k: d["stress_weights"][k] for k in ("audio_seconds_per_s","range_fallbacks","parity")}))
print("half", d["half_path"]["audio_seconds_per_s"], "exact", d["fp32_exact_path"]["audio_seconds_per_s"])
print("latency", json.dumps(d["latency"]["f32x3"]), json.dumps(d["latency"]["half"]))
print("roofline", json.dumps(d["roofline"]))
PY
echo "== energy probe"
timeout 600 python tools/energy_probe.py 33 1.0 2>&1 | grep -v amdgpu.ids | tee $O/energy_probe.txt

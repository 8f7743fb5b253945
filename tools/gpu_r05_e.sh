#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu --tb=short -k "attention or model or scale" 2>&1 | tail -30 > $O/tests.txt
cat $O/tests.txt
echo "== flip soak"
timeout 900 python tools/flip_soak.py gpu --tag e 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
echo "== bench p16 = 2 / 1 / 0, alternating"
for i in 1 2; do
  for p in 2 1 0; do
    timeout 600 python bench.py --no-cpu-baseline --no-extras --x3-p16 $p 2>/dev/null > $O/bench_p${p}_$i.json
    python - <<PY
import json
d=json.loads(open("$O/bench_p${p}_$i.json").read().strip().splitlines()[-1])
e=d.get("energy") or {}
print("p16=$p run $i", d["value"], d["ms_per_step"], "J/step", e.get("joules_per_step"), "W", e.get("avg_package_power_W"), "attn ms", d["breakdown"]["attn_flash"]["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"]["peak"], d["roofline"]["frac"])
PY
  done
done

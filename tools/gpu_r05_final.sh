#!/bin/bash
# end-of-round validation: the whole GPU test suite, the bit-for-bit soak of the pipelined path, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05final
mkdir -p $O
rm -f gpurun_out/test_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 600 python tools/soak.py 400 pinned f32x3 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
timeout 600 python tools/soak.py 200 device half 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 1200 python bench.py 2>$O/bench.err > $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], json.dumps(d["parity"]), json.dumps(d["energy"]))
print("latency", json.dumps(d["latency"]["f32x3"]))
print("cfg5", json.dumps(d["configs"]["cfg5"]))
PY

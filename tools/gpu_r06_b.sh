#!/bin/bash
# round 6, second GPU call: whole GPU suite (no -x), latency of the one-call path (walls + kernel trace of one 30 s call), bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06b
mkdir -p $O
rm -f gpurun_out/test_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
cp gpurun_out/test_report.jsonl $O/parity_report.jsonl 2>/dev/null
bash tools/latency_profile.sh 2>&1 | grep -v amdgpu.ids | tee $O/latency_trace.txt
timeout 900 python bench.py 2>$O/bench.err > $O/bench.json
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"])
print("latency", json.dumps(d["latency"]))
PY

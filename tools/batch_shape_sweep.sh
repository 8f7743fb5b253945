#!/bin/bash
# batch-shape sweep of the headline + full default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05g
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --no-dist"
for cfg in "" "--streams 1" "--streams 3 --slice 22" "--tracks 8" "--tracks 9 --streams 3" "--tracks 12 --slice 66" ""; do
  timeout 300 $B $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$cfg'.ljust(28), d['value'], d['ms_per_step'], (d.get('energy') or {}).get('joules_per_step'))"
done | tee $O/sweep.txt
echo "== full default line"
timeout 1200 python bench.py 2>$O/bench.err > $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","parity","energy","roofline","cpu_baseline"):
    print(k, json.dumps(d[k])[:600])
print("stress", json.dumps(d["stress_weights"])[:700])
print("half", d["half_path"]["audio_seconds_per_s"], json.dumps(d["half_path"]["parity"]))
print("exact", d["fp32_exact_path"]["audio_seconds_per_s"], json.dumps(d["fp32_exact_path"]["parity"]))
print("latency", json.dumps(d["latency"])[:900])
print("configs", json.dumps(d["configs"])[:900])
print("forward_only", json.dumps(d["forward_only"])[:500])
print("host_inclusive", json.dumps(d["host_inclusive"])[:300])
PY

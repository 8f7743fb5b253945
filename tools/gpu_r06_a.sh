#!/bin/bash
# round 6, first GPU call: the whole GPU test suite on the new default arithmetic, the default bench line, and the A/B that bounds
# what a fused x3 layer tail could save (a build whose FF hidden activation never leaves L2 / MALL: -DBT_ABL_HID_WRAP=2048)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06a
mkdir -p $O
rm -f gpurun_out/test_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
cp gpurun_out/test_report.jsonl $O/parity_report.jsonl 2>/dev/null
timeout 900 python bench.py 2>$O/bench.err > $O/bench.json
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], json.dumps(d["parity"]), json.dumps(d["energy"]))
print("roofline", json.dumps(d["roofline"]))
print("p16 legs", {k: (v["value"], v["ms_per_step"]) for k, v in d.items() if k.startswith("value_p16")})
print("latency", json.dumps(d["latency"]["f32x3"]))
print("breakdown", {k: v["ms_per_step"] for k, v in d["breakdown"].items()})
PY
export BT_DEV=1
for i in 1 2 3; do
  for l in tools/variants/lib_hidwrap.so ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/$l; else unset BT_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['breakdown']
print('${l:-in-tree}'.ljust(32), d['ms_per_step'], d['energy'].get('joules_per_step'), d['energy'].get('avg_package_power_W'), ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))" | tee -a $O/ab_hidwrap.txt
  done
done

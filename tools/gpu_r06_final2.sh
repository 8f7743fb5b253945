#!/bin/bash
# round 6, end-of-round validation part 2: the flip soak on the real kernels (every cached track x 3 weight styles x 6 arithmetics)
# and -- with profiles/r06_flip_frontier.json in place (run part 2a, report in the build container, then part 2b) -- the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06final
mkdir -p $O
if [ "$1" = "soak" ]; then
  rm -f gpurun_out/flip_soak/*.json
  timeout 2400 python tools/flip_soak.py gpu --tracks 96 --tag r06 --schemes exact,x3,x3p16m,x3p16f,x3p16,half 2>&1 | grep -v amdgpu.ids | tee $O/flip_soak_gpu.txt
else
  timeout 1200 python bench.py 2>$O/bench.err > $O/bench.json
  tail -3 $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], json.dumps(d["parity"]), json.dumps(d["energy"]))
print("roofline", json.dumps(d["roofline"]))
print("legs", {k: (v["value"], v["ms_per_step"]) for k, v in d.items() if k.startswith("value_p16")})
print("latency", json.dumps(d["latency"]["f32x3"]), "half", d["half_path"]["audio_seconds_per_s"])
print("cfg5", json.dumps(d["configs"]["cfg5"]["as_written_mx_e4m3_operands"])[:400])
PY
fi

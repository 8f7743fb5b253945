#!/bin/bash
# Round 5, GPU call A: the P16 attention kernels (unit tests), model-level tests, the flip-rate soak on the real kernels,
# headline A/B of the two P.V arithmetics with the energy accumulator, and the power-cap sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
echo "== unit tests: x3 attention" > $O/log.txt
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu -k "attention" -x 2>&1 | tail -15 >> $O/log.txt
echo "== model / scale tests" >> $O/log.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu 2>&1 | tail -25 >> $O/log.txt
echo "== flip soak (real kernels)" >> $O/log.txt
timeout 900 python tools/flip_soak.py gpu --tag a 2>&1 | tail -20 >> $O/log.txt
echo "== bench A/B" >> $O/log.txt
for i in 1 2; do
  for p in 1 0; do
    timeout 600 python bench.py --no-cpu-baseline --no-extras --x3-p16 $p 2>$O/bench_p${p}_$i.err > $O/bench_p${p}_$i.json
    python - <<PY >> $O/log.txt
import json
d=json.loads(open("$O/bench_p${p}_$i.json").read().strip().splitlines()[-1])
print("p16=$p run $i", d["value"], d["ms_per_step"], json.dumps(d.get("energy")), json.dumps({k: d["roofline"][k] for k in ("achieved","peak","frac","avg_launch_ms")}))
print("   breakdown", json.dumps({k: v["ms_per_step"] for k, v in d["breakdown"].items()}))
PY
  done
done
echo "== cap sweep" >> $O/log.txt
timeout 900 python tools/cap_sweep.py 1200 1000 2>&1 | tail -12 >> $O/log.txt
cat $O/log.txt

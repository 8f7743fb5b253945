#!/bin/bash
# round 6, third GPU call: profile collection (kernel traces, SQ counters, HBM traffic incl. the hidden-activation ablation), the
# A/B of non-temporal stores, the bf16 variant through the scale tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06c
mkdir -p $O
bash tools/profile_r06.sh 2>&1 | tail -40
cd $R
export BT_DEV=1
for i in 1 2 3; do
  for l in tools/variants/lib_ntstore.so ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/$l; else unset BT_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['breakdown']
print('${l:-in-tree}'.ljust(32), d['ms_per_step'], d['energy'].get('joules_per_step'), d['energy'].get('avg_package_power_W'), ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))" | tee -a $O/ab_ntstore.txt
  done
done
export BT_LIB_PATH=$R/tools/variants/lib_bf16.so
rm -f gpurun_out/test_report.jsonl
( echo "# -DBT_HALF_BF16 build of the library (tools/build_variant.py bf16 -DBT_HALF_BF16) through tests/test_gpu_scale.py and the half-path tests of tests/test_gpu_model.py";
  timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_model.py -q -m gpu -k "half or bf16 or autocast" 2>&1 | tail -12;
  grep -h "scale_parity\|forward_half" gpurun_out/test_report.jsonl | head -20 ) > $O/bf16_variant.txt 2>&1
cat $O/bf16_variant.txt | tail -25
unset BT_LIB_PATH BT_DEV

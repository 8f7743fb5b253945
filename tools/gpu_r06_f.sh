#!/bin/bash
# round 6, sixth GPU call: the fp16 attention on two query blocks per wave (attn_frag_hq2_kernel): parity + bit identity tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frag.py -q -m gpu -x -k "attention_frag" 2>&1 | tail -15 | tee $O/pytest_frag.txt
export BT_DEV=1
for i in 1 2 3; do
  for l in tools/variants/lib_hq2off.so ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/$l; else unset BT_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --steps 30 --prec half 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['breakdown']
print('${l:-in-tree}'.ljust(32), d['ms_per_step'], d['value'], d['energy'].get('joules_per_step'), ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))" | tee -a $O/ab_hq2.txt
  done
done
unset BT_DEV BT_LIB_PATH

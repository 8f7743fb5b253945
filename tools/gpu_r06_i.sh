#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06i
mkdir -p $O
export BT_DEV=1
for i in 1 2 3; do
  for l in tools/variants/lib_f2n2.so tools/variants/lib_f2n3.so ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/$l; else unset BT_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['breakdown']
print('${l:-in-tree}'.ljust(32), d['ms_per_step'], d['energy'].get('joules_per_step'), ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))" | tee -a $O/ab.txt
  done
done

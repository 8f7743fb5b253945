#!/bin/bash
# In-situ A/B of two builds of the library on ONE box (boxes differ by several %): alternating bench runs.
#   tools/ab.sh tools/bin/lib_a.so [tools/bin/lib_b.so ...]     (the in-tree build is always the last candidate)     [extra bench args via AB_ARGS]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for i in 1 2 3; do
  for l in "$@" ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/$l; else unset BT_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-extras --steps 30 $AB_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['breakdown']
print('${l:-in-tree}'.ljust(28), d['ms_per_step'], ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))"
  done
done

#!/bin/bash
# In-situ A/B of builds of the library on ONE box (boxes differ by several %): three rounds of alternating bench runs.
#   tools/ab.sh tools/variants/lib_a.so [tools/variants/lib_b.so ...]     (the in-tree build is always the last candidate)
#   AB_ARGS="--prec half" tools/ab.sh ...                                   (extra bench.py arguments)
# Variant builds: python tools/build_variant.py NAME -DSWITCH=...  ->  tools/variants/lib_NAME.so
# columns: build, ms per step, joules per step, average package W, per-category ms per step (roofline leg, one stream)
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
print('${l:-in-tree}'.ljust(32), d['ms_per_step'], d['energy'].get('joules_per_step'), d['energy'].get('avg_package_power_W'), ' '.join('%s=%.3f' % (k[:8], v['ms_per_step']) for k, v in b.items()))"
  done
done

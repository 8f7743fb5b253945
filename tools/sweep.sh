#!/bin/bash
# batch-shape sweep of the headline workload on one box: tracks per step x chunks per forward launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cfg in "6 0" "6 33" "6 22" "4 0" "4 22" "8 0" "8 44" "8 22" "12 0" "12 44" "16 0" "16 44"; do
  set -- $cfg
  python bench.py --tracks $1 --slice $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --watchdog 120 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('tracks $1 slice $2:', d['ms_per_step'], 'ms', d['value'], 'audio-s/s', 'fwd', d['roofline']['forward_ms_per_step'])"
done

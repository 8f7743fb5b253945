#!/bin/bash
# Development: package power / shader clock while bench.py's timed region runs (default f32x3 path, 6 x 300 s tracks per step).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/power_probe_bench.txt
: > $O
python bench.py --no-cpu-baseline --no-extras --no-dist --steps 300 --warmup 3 > /tmp/ppb_$$.json 2>/tmp/ppb_$$.err &
P=$!
# the timed region starts ~7 s after launch (model set-up, warm-up) and lasts ~9 s
sleep 9
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>&1 | grep -iE "Package Power|sclk" | tr '\n' ' ' >> $O; echo >> $O; sleep 0.4; done
wait $P
grep "timed region" /tmp/ppb_$$.err >> $O
python -c "
import json,sys
d=json.loads(open('/tmp/ppb_$$.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])" >> $O
cat $O

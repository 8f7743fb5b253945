#!/bin/bash
# Development: package power / shader clock while ONE x3 GEMM of the main layers loops (tools/x3_probe.py gloop mode).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/power_probe_gemm.txt
: > $O
export BT_DEV=1
for w in ff1 ff2 qkv out; do
  python tools/x3_probe.py 33 gloop:$w 5 > /tmp/ppg_$$.log 2>&1 &
  P=$!
  sleep 4.5
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -iE "Package Power|sclk" | sed "s/^/[$w] /" >> $O; sleep 0.5; done
  wait $P
  grep gemm3 /tmp/ppg_$$.log | tail -2 | sed "s/^/[$w] /" >> $O
done
cat $O

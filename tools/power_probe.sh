#!/bin/bash
# Development: package power and shader clock (rocm-smi, every 0.5 s) while ONE kernel loops -- the x3 attention kernels
# (tools/x3_probe.py loop mode: 2 = compiler-scheduled 64-key tiles, 5 = hand-scheduled two query blocks) and, with
# tools/variants/lib_abl7.so present, the same loop with nothing but its MFMAs.  -> gpurun_out/power_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/power_probe.txt
: > $O
export BT_DEV=1
sample() {  # $1 = label, rest = command
  label=$1; shift
  "$@" > /tmp/pp_$$.log 2>&1 &
  P=$!
  sleep 2.0
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>&1 | grep -iE "Package Power|sclk" | sed "s/^/[$label] /" >> $O
    sleep 0.5
  done
  wait $P
  grep -v amdgpu.ids /tmp/pp_$$.log | sed "s/^/[$label] /" >> $O
}
rocm-smi --showpower --showclocks 2>&1 | grep -iE "Package Power|sclk" | sed "s/^/[idle] /" >> $O
unset BT_LIB_PATH
sample "x3 attention, 64-key tiles (x3 = 2)" python tools/x3_probe.py 16 loop:2 5
sample "x3 attention, hand-scheduled (x3 = 5)" python tools/x3_probe.py 16 loop:5 5
for l in abl1 abl7; do
  if [ -f tools/variants/lib_$l.so ]; then
    BT_LIB_PATH=$R/tools/variants/lib_$l.so sample "hand-scheduled loop, ablation $l" python tools/x3_probe.py 16 loop:5 5
  fi
done
sample "x3 forward, 16 chunks" python bench.py --workload forward --chunks 16 --prec f32x3 --steps 600 --warmup 5 --no-cpu-baseline --no-extras --no-dist --min-seconds 0
sample "half forward, 16 chunks" python bench.py --workload forward --chunks 16 --prec half --steps 1200 --warmup 5 --no-cpu-baseline --no-extras --no-dist --min-seconds 0
rocm-smi --showmaxpower 2>&1 | grep -i power >> $O
cat $O

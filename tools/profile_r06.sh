#!/bin/bash
# Round-6 profile collection on one GPU box (outputs under gpurun_out/prof_r06, summarised by tools/profile_summary.py):
#   kernel traces of the bench command (headline, single-stream headline, 16-chunk forward) for the half AND the f32x3
#   path, SQ / GRBM counters (MFMA-busy cycles, wave cycles, GUI active) of both forwards, HBM traffic counters in SEPARATE
#   passes (MI355X_MICROARCH.md, HBM section) for the half forward, the f32x3 forward, BASELINE config 3 (small0 fp32, 128
#   chunks) and the headline, and rocm-smi power / clock samples while the forwards loop.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --no-dist --watchdog 150 --min-seconds 0"
HEAD="$B --steps 5 --warmup 1 --prec half"   # (bench.py's default precision is f32x3 since round 4: the half legs say so)
HEADX="$B --steps 5 --warmup 1 --prec f32x3"
FWD="$B --workload forward --chunks 16 --prec half --steps 10 --warmup 2"
FWDX="$B --workload forward --chunks 16 --prec f32x3 --steps 8 --warmup 2"
CFG3="$B --workload forward --model small0 --prec f32 --chunks 128 --steps 2 --warmup 1"
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { name=$1; shift; timeout 240 rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name $?"; }
run trace_head     --kernel-trace --stats -d $O/trace_head -o t --output-format csv -- $HEAD
run trace_head1s   --kernel-trace --stats -d $O/trace_head1s -o t --output-format csv -- $HEAD --streams 1
run trace_fwd      --kernel-trace --stats -d $O/trace_fwd -o t --output-format csv -- $FWD
run trace_fwd_x3   --kernel-trace --stats -d $O/trace_fwd_x3 -o t --output-format csv -- $FWDX
run trace_head_x3  --kernel-trace --stats -d $O/trace_head_x3 -o t --output-format csv -- $HEADX --streams 1
run trace_head_x3_2s --kernel-trace --stats -d $O/trace_head_x3_2s -o t --output-format csv -- $HEADX
run pmc_sq         --kernel-trace --pmc $SQ1 -d $O/pmc_sq -o p --output-format csv -- $FWD
run pmc_sq2        --kernel-trace --pmc $SQ2 -d $O/pmc_sq2 -o p --output-format csv -- $FWD
run pmc_sq_x3      --kernel-trace --pmc $SQ1 -d $O/pmc_sq_x3 -o p --output-format csv -- $FWDX
run pmc_sq2_x3     --kernel-trace --pmc $SQ2 -d $O/pmc_sq2_x3 -o p --output-format csv -- $FWDX
for w in "" _x3 _cfg3 _head; do
  case "$w" in "") C="$FWD";; _x3) C="$FWDX";; _cfg3) C="$CFG3";; _head) C="$HEADX";; esac
  run pmc_fetch$w --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch$w -o p --output-format csv -- $C
  run pmc_write$w --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write$w -o p --output-format csv -- $C
done
# the ablation that bounds the fused x3 layer tail (-DBT_ABL_HID_WRAP=2048: the FF hidden activation never leaves L2 / MALL; results
# garbage by construction): its traffic and kernel times next to the real forward's
if [ -f $R/tools/variants/lib_hidwrap.so ]; then
  export BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_hidwrap.so
  run trace_fwd_x3_hidwrap --kernel-trace --stats -d $O/trace_fwd_x3_hidwrap -o t --output-format csv -- $FWDX
  run pmc_fetch_x3_hidwrap --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_x3_hidwrap -o p --output-format csv -- $FWDX
  run pmc_write_x3_hidwrap --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_x3_hidwrap -o p --output-format csv -- $FWDX
  unset BT_DEV BT_LIB_PATH
fi
# power / clocks: idle sample, then samples while the forward loops, every 0.5 s
cd $R
( rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -v "^$" ) > $O/smi_idle.txt
for tag in fwd fwd_x3; do
  P=half; [ $tag = fwd_x3 ] && P=f32x3
  python bench.py --workload forward --chunks 16 --prec $P --steps 1500 --warmup 5 --no-cpu-baseline --no-extras --no-dist --watchdog 150 --min-seconds 0 > $O/loop_$tag.json 2>/dev/null &
  LP=$!
  sleep 4
  for i in 1 2 3 4 5 6; do ( date +%s.%N; rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk|fclk" ) >> $O/smi_$tag.txt; sleep 0.5; done
  wait $LP
done
rocm-smi --showmaxpower 2>&1 | grep -iE "power" >> $O/smi_idle.txt
find $O -name "*.csv" | wc -l
du -sh $O
# (the csv files travel back with gpurun_out/; tools/profile_summary.py turns them into profiles/r05_* in the build container)
find $O -name "*.db" -delete 2>/dev/null

#!/bin/bash
# Round-2 profile collection on one GPU box (outputs under gpurun_out/prof_r02, summarised by tools/profile_summary.py):
#   kernel trace + stats of the bench command (headline workload and BASELINE config 2), SQ / GRBM counters (MFMA busy
#   cycles, wave cycles, GUI active), HBM traffic counters in SEPARATE passes (MI355X_MICROARCH.md, HBM section), and
#   rocm-smi power / clock samples while the forward loops.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
HEAD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --watchdog 150"
FWD="python $R/bench.py --workload forward --chunks 16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --watchdog 150"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_head -o t --output-format csv -- $HEAD > $O/trace_head.log 2>&1; echo "trace_head $?"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_head1s -o t --output-format csv -- $HEAD --streams 1 > $O/trace_head1s.log 2>&1; echo "trace_head1s $?"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_fwd -o t --output-format csv -- $FWD > $O/trace_fwd.log 2>&1; echo "trace_fwd $?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- $FWD > $O/pmc_sq.log 2>&1; echo "pmc_sq $?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq2 -o p --output-format csv -- $FWD > $O/pmc_sq2.log 2>&1; echo "pmc_sq2 $?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $FWD > $O/pmc_fetch.log 2>&1; echo "pmc_fetch $?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- $FWD > $O/pmc_write.log 2>&1; echo "pmc_write $?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_head -o p --output-format csv -- $HEAD > $O/pmc_fetch_head.log 2>&1; echo "pmc_fetch_head $?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_head -o p --output-format csv -- $HEAD > $O/pmc_write_head.log 2>&1; echo "pmc_write_head $?"
# power / clocks: idle sample, then samples while the forward loops (~12 s), every 0.5 s
cd $R
( rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -v "^$" ) > $O/smi_idle.txt
python bench.py --workload forward --chunks 16 --steps 3000 --warmup 5 --no-cpu-baseline --no-extras --watchdog 150 > $O/loop_fwd.json 2>/dev/null &
LP=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do ( date +%s.%N; rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk|fclk" ) >> $O/smi_fwd.txt; sleep 0.5; done
wait $LP
python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extras --watchdog 150 > $O/loop_head.json 2>/dev/null &
LP=$!
sleep 4
for i in 1 2 3 4 5 6; do ( date +%s.%N; rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk" ) >> $O/smi_head.txt; sleep 0.5; done
wait $LP
rocm-smi --showmaxpower 2>&1 | grep -iE "power" >> $O/smi_idle.txt
find $O -name "*.csv" | head -40
du -sh $O

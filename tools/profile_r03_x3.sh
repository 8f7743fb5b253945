#!/bin/bash
# Refresh of the f32x3 passes of tools/profile_r03.sh (same output directory: gpurun_out/prof_r03) after a change that
# touches only that path.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --watchdog 150 --min-seconds 0"
HEAD="$B --steps 5 --warmup 1"
FWDX="$B --workload forward --chunks 16 --prec f32x3 --steps 8 --warmup 2"
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { name=$1; shift; rm -rf $O/$name; timeout 240 rocprofv3 "$@" > $O/$name.log 2>&1; echo "$name $?"; }
run trace_fwd_x3   --kernel-trace --stats -d $O/trace_fwd_x3 -o t --output-format csv -- $FWDX
run trace_head_x3  --kernel-trace --stats -d $O/trace_head_x3 -o t --output-format csv -- $HEAD --prec f32x3 --streams 1
run pmc_sq_x3      --kernel-trace --pmc $SQ1 -d $O/pmc_sq_x3 -o p --output-format csv -- $FWDX
run pmc_sq2_x3     --kernel-trace --pmc $SQ2 -d $O/pmc_sq2_x3 -o p --output-format csv -- $FWDX
run pmc_fetch_x3   --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_x3 -o p --output-format csv -- $FWDX
run pmc_write_x3   --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_x3 -o p --output-format csv -- $FWDX
find $O -name "*.db" -delete 2>/dev/null

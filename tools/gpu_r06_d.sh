#!/bin/bash
# round 6, fourth GPU call: the MX e4m3 GEMM (test + rate next to the fp16 / hl32 GEMMs), the hidden-activation ablation's traffic
# counters, the full suite again on the nt-store build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mx8.py -q -m gpu 2>&1 | tail -12 | tee $O/pytest_mx8.txt
timeout 600 python tools/mx8_probe.py 33 2>&1 | grep -v amdgpu.ids | tee $O/mx8_probe.txt
timeout 600 python tools/mx8_probe.py 16 2>&1 | grep -v amdgpu.ids | tee -a $O/mx8_probe.txt
P=$R/gpurun_out/prof_r06
mkdir -p $P
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --no-dist --watchdog 150 --min-seconds 0"
FWDX="$B --workload forward --chunks 16 --prec f32x3 --steps 8 --warmup 2"
run() { name=$1; shift; timeout 240 rocprofv3 "$@" > $P/$name.log 2>&1; echo "$name $?"; }
export BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_hidwrap.so
run trace_fwd_x3_hidwrap --kernel-trace --stats -d $P/trace_fwd_x3_hidwrap -o t --output-format csv -- $FWDX
run pmc_fetch_x3_hidwrap --kernel-trace --pmc FETCH_SIZE -d $P/pmc_fetch_x3_hidwrap -o p --output-format csv -- $FWDX
run pmc_write_x3_hidwrap --kernel-trace --pmc WRITE_SIZE -d $P/pmc_write_x3_hidwrap -o p --output-format csv -- $FWDX
unset BT_DEV BT_LIB_PATH
# the default build again (nt stores): trace + traffic of the x3 forward
run trace_fwd_x3 --kernel-trace --stats -d $P/trace_fwd_x3 -o t --output-format csv -- $FWDX
run pmc_fetch_x3 --kernel-trace --pmc FETCH_SIZE -d $P/pmc_fetch_x3 -o p --output-format csv -- $FWDX
run pmc_write_x3 --kernel-trace --pmc WRITE_SIZE -d $P/pmc_write_x3 -o p --output-format csv -- $FWDX
find $P -name "*.db" -delete 2>/dev/null
cd $R
rm -f gpurun_out/test_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
cp gpurun_out/test_report.jsonl $O/parity_report.jsonl 2>/dev/null

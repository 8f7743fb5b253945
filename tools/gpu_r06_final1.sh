#!/bin/bash
# round 6, end-of-round validation part 1: whole GPU suite on the final build, bit-for-bit soaks, smoke(), profile collection
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06final
mkdir -p $O
rm -f gpurun_out/test_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
cp gpurun_out/test_report.jsonl $O/parity_report.jsonl 2>/dev/null
timeout 600 python tools/soak.py 400 pinned f32x3 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
timeout 600 python tools/soak.py 200 device half 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
bash tools/latency_profile.sh 2>&1 | grep -v amdgpu.ids > $O/latency_trace.txt
tail -5 $O/latency_trace.txt
bash tools/profile_r06.sh 2>&1 | tail -30

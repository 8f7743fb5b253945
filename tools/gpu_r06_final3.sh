#!/bin/bash
# round 6, last validation of the final build: whole GPU suite, bit-for-bit soaks, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06final
mkdir -p $O
rm -f gpurun_out/test_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
grep -E "passed|failed|error" $O/pytest_gpu.txt
cp gpurun_out/test_report.jsonl $O/parity_report.jsonl 2>/dev/null
timeout 600 python tools/soak.py 400 pinned f32x3 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt | tail -2
timeout 600 python tools/soak.py 200 device half 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt | tail -2
bash tools/gpu_r06_final2.sh

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short -k "gemm3" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu --tb=short 2>&1 | tail -8
for p in f32x3 half; do python tools/latency_probe.py $p 30; python tools/latency_probe.py $p 300; done 2>&1 | grep -v amdgpu.ids | tee $O/latency.txt

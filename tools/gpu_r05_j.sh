#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short -k "hl8 or f8 or qkv" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu --tb=short -k "cfg5 or outlier" 2>&1 | tail -12
grep "cfg5\|outlier_weights" gpurun_out/test_report.jsonl | tail -12
timeout 1200 python tools/flip_soak.py gpu --tag f8 --schemes x3p16f8ff,x3p16f8 2>&1 | grep -v amdgpu.ids

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short -k "attention" -x 2>&1 | tail -3
bash tools/gpu_fix3.sh
bash tools/gpu_fix2.sh 2>&1 | grep "forward\|attn_f"
bash tools/gpu_ab.sh

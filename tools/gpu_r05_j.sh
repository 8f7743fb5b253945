#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short -x -k "hl8_rows" 2>&1 | tail -12

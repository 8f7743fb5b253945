#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests/test_gpu_bench_dist.py -q -m gpu -x 2>&1 | tail -30

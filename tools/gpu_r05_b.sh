#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu -k "attention" --tb=short 2>&1 | tail -40 > $O/x3.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/model.txt
cat $O/x3.txt $O/model.txt

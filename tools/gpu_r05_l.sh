#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for st in lively outlier; do BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_dev.so timeout 300 python tools/safe_rate_probe.py 16 $st 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/rerun_rate.txt
bash tools/gpu_r05_final.sh

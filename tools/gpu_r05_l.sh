#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python tools/flip_soak.py gpu --tag f8 --schemes x3p16f8ff,x3p16f8 2>&1 | grep -v amdgpu.ids
bash tools/gpu_r05_final.sh

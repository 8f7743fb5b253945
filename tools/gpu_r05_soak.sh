#!/bin/bash
# the flip-rate soak on the real kernels (48 tracks x 3 weight styles x 7 arithmetics) and the operand-rounding simulations of
# the schemes that are not built (8 tracks x 3 styles), torch on the GPU for the simulation's fp32 matmuls
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -f gpurun_out/flip_soak/*.json
timeout 1200 python tools/flip_soak.py gpu --tracks 48 --tag r05 --schemes exact,x3,x3p16m,x3p16,x3p16f8ff,x3p16f8,half 2>&1 | grep -v amdgpu.ids
timeout 1500 python tools/flip_soak.py sim --device cuda --tracks 8 --schemes vhi,e4m3,e2m3 --tag r05 2>&1 | grep -v amdgpu.ids | tail -12

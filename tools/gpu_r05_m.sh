#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python tools/energy_probe.py 33 0.6 tiles 2>&1 | grep -v amdgpu.ids | tee gpurun_out/energy_probe.txt

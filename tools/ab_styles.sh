#!/bin/bash
# tools/ab.sh on the benchmark's weights and on the outlier stress weights: tools/variants/lib_old.so against the in-tree build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== lively (the benchmark's weights)"
bash tools/ab.sh tools/variants/lib_old.so 2>&1 | grep -v amdgpu.ids | cut -c1-60
echo "== outlier"
BT_BENCH_STYLE=outlier bash tools/ab.sh tools/variants/lib_old.so 2>&1 | grep -v amdgpu.ids | cut -c1-60

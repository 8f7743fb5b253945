#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05d
mkdir -p $O
mkdir -p /tmp/ub && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/dot2_denorm.hip -o /tmp/ub/dot2_denorm 2>/dev/null && /tmp/ub/dot2_denorm > $O/dot2_denorm.txt 2>&1
cat $O/dot2_denorm.txt
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu -k "attention" --tb=short 2>&1 | tail -15 > $O/x3.txt
echo "== dot2 variant: P16 unit tests" >> $O/x3.txt
BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_dot2.so timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu -k "p16 or per_query or at_scale" --tb=short 2>&1 | tail -15 >> $O/x3.txt
cat $O/x3.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "ablation" --tb=line 2>&1 | tail -8
echo "== A/B in-tree (mfma row sums) vs dot2"
bash tools/ab.sh tools/variants/lib_dot2.so 2>&1 | tee $O/ab_dot2.txt

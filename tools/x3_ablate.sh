#!/bin/bash
# Development (GPU box): where does the x3 GEMM's time go?  tools/x3_probe.py on a -DBT_DEV build (tools/build_variant.py dev
# with BT_DEV_BUILD=1) with BT_G3_ABL = 0 (whole kernel) / 1 (no LDS-DMA after the prologue) / 4 (no MFMAs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1 BT_LIB_PATH=$R/tools/variants/lib_dev.so
for abl in 0 1 4; do
  echo "== BT_G3_ABL=$abl"
  BT_G3_ABL=$abl python tools/x3_probe.py 16 2>&1 | grep gemm3
done

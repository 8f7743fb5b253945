#!/bin/bash
# Development: A/B of attention variants on one box -- tools/attn_probe.py (half kernel) for each
# tools/variants/lib_NAME.so given, then for the in-tree build, twice; AB_X3=1 adds tools/x3_probe.py's attention lines,
# AB_TEST=1 the attention unit tests of each variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for i in 1 2; do
  for l in "$@" ""; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/tools/variants/lib_$l.so; else unset BT_LIB_PATH; fi
    echo "== ${l:-in-tree}"
    python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
    [ -n "$AB_X3" ] && python tools/x3_probe.py 16 attn 2>&1 | grep "attention x3 variant 2"
    [ -n "$AB_TEST" ] && [ $i = 1 ] && python -m pytest tests/test_gpu_frag.py -q -k "attn or attention" 2>&1 | tail -2
  done
done

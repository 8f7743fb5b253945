#!/bin/bash
# Development: what the hand-scheduled x3 attention loop spends where -- ablated builds of attn_frag_x3q2_kernel
# (tools/gen/attn_x3_loop.py: BT_X3Q2_ABL bits 1 no softmax VALU, 2 no fragment reads, 4 no refill / barrier, 8 no P.V MFMAs,
# 16 no score MFMAs; results of ablated builds are garbage), each timed by tools/x3_probe.py on the two launch shapes.
#   for a in 1 2 3 4 7 8 16 9 17; do python tools/build_variant.py abl$a -DBT_X3Q2_ABL=$a; done; tools/attn_x3q2_ablate.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for i in 1 2; do
  for l in "" abl1 abl2 abl3 abl4 abl7 abl8 abl16 abl9 abl17; do
    if [ -n "$l" ]; then export BT_LIB_PATH=$R/tools/variants/lib_$l.so; [ -f $BT_LIB_PATH ] || continue; else unset BT_LIB_PATH; fi
    echo "== ${l:-product}"
    python tools/x3_probe.py 16 attn 2>&1 | grep "variant 5" | grep -v "block 2"
  done
done

#!/bin/bash
# How far to shift the byte sections of an hl8 activation (csrc/common.h BT_HL8_ACT_EXP; profiles/r05_hl8_shift.txt): the level-2 flip
# soak on variant builds of the library, one box.  Before the gpurun call, in the build container:
#   python tools/build_variant.py e2 -DBT_HL8_ACT_EXP=2; python tools/build_variant.py e4 -DBT_HL8_ACT_EXP=4   (the in-tree build is 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for v in e2 e4 ""; do
  if [ -n "$v" ]; then export BT_LIB_PATH=$R/tools/variants/lib_$v.so; else unset BT_LIB_PATH; fi
  echo "== hl8 activation shift variant: ${v:-e3 (in-tree)}"
  timeout 1200 python tools/flip_soak.py gpu --tag f8_${v:-e3} --schemes x3p16f8 2>&1 | grep -v amdgpu.ids
done

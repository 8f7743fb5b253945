#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for v in e2 e3 ""; do
  if [ -n "$v" ]; then export BT_LIB_PATH=$R/tools/variants/lib_$v.so; else unset BT_LIB_PATH; fi
  echo "== hl8 activation shift variant: ${v:-e4 (in-tree)}"
  timeout 1200 python tools/flip_soak.py gpu --tag f8_${v:-e4} --schemes x3p16f8 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# Development: the half attention kernel's key loop with parts removed (-DBT_ATTN_EXPT=bits variants built by
# tools/build_variant.py eN), timed on the two launch shapes of the final0 forward by tools/attn_probe.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export BT_DEV=1
for l in "$@"; do
  export BT_LIB_PATH=$R/tools/variants/lib_$l.so
  echo "== $l"; python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
done

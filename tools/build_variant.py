#!/usr/bin/env python
"""Build a variant of the library for A/B runs on one GPU box (tools/ab.sh):
    python tools/build_variant.py NAME -DBT_ATTN_ROWSUM=1 ...   ->   tools/variants/lib_NAME.so
(development tool; the product build is beat_this_amd._lib.build() with no defines)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beat_this_amd import _lib  # noqa: E402

name, defines = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "variants")   # (travels to the GPU box; git-ignored)
os.makedirs(out, exist_ok=True)
print(_lib.build(force=True, lib_path=os.path.join(out, f"lib_{name}.so"), obj_dir=os.path.join(out, f"obj_{name}"), defines=defines))

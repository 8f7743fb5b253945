#!/usr/bin/env python
"""Development: copies per benchmark step from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --no-cpu-baseline --no-extras --no-dist \
        --steps 2 --warmup 1 --min-seconds 0 --prec f32x3
    python tools/trace_copies.py DIR/t_kernel_trace.csv
A step = one logmel_kernel launch (the front-end of one batch of tracks); ROCclr's copyBuffer / fillBuffer kernels between two
of them are what the host path adds to a step (table uploads, the range-flag and beat-index read-backs).  Everything before
the first stem_kernel is model set-up (weight upload and packing)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = next(i for i, n in enumerate(names) if "stem_kernel" in n)
setup = collections.Counter("copy" if "copyBuffer" in n else "fill" if "fillBuffer" in n else "kernel" for n in names[:first])
print(f"set-up (before the first forward): {setup['copy']} copyBuffer, {setup['fill']} fillBuffer, {setup['kernel']} kernels")
marks = [i for i, n in enumerate(names) if "logmel_kernel" in n]
print("step  kernels  copyBuffer  fillBuffer  ms")
for k in range(len(marks) - 1):
    seg = names[marks[k]: marks[k + 1]]
    ms = (int(rows[marks[k + 1]]["Start_Timestamp"]) - int(rows[marks[k]]["Start_Timestamp"])) / 1e6
    print(f"{k:4d}  {len(seg):7d}  {sum('copyBuffer' in n for n in seg):10d}  {sum('fillBuffer' in n for n in seg):10d}  {ms:6.2f}")

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/attn_fix
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python tools/attn_fix_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr -o p --output-format csv -- python $R/tools/attn_fix_probe.py > $O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/attn_fix/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "attn_fix" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# 23 launches per case
for i in range(0, len(d), 23):
    c = d[i + 3:i + 23]
    print(f"case {i // 23}: fix-up kernel {sum(c) / max(len(c), 1):8.1f} us avg (min {min(c):.1f}, max {max(c):.1f})")
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/fix2
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python tools/_fwd_style.py lively 2>&1 | grep -v amdgpu.ids
python tools/_fwd_style.py outlier 2>&1 | grep -v amdgpu.ids
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o p --output-format csv -- python $R/tools/_fwd_style.py outlier > $O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/fix2/tr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']} %")
for r in rows:
    if "attn_fix" in r["Name"]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']} %")
PY

#!/bin/bash
# kernel trace of the single-file path (tools/latency_probe.py): per-kernel time of one 30 s call
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_lat
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for p in f32x3 half; do python $R/tools/latency_probe.py $p 30; python $R/tools/latency_probe.py $p 300; done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/trace -o p --output-format csv -- python $R/tools/latency_probe.py f32x3 30 > $O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections, re
src = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_lat", "trace")
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the last ~N kernels: find the last stem_kernel and take from the preceding resample / logmel kernel
idx = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
call = rows[i0:i1]
t0 = int(call[0]["Start_Timestamp"]); busy = 0
agg = collections.OrderedDict()
for r in call:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
    n = re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]))[:50]
    v = agg.setdefault(n, [0, 0]); v[0] += 1; v[1] += d
span = int(call[-1]["End_Timestamp"]) - t0
print(f"one call (stem to stem): {len(call)} kernels, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:50s} {c:3d} {d / 1e3:8.1f} us")
gaps = sorted(((int(call[i + 1]["Start_Timestamp"]) - int(call[i]["End_Timestamp"])) / 1e3, i) for i in range(len(call) - 1))
short = lambda r: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]))[:44]  # noqa: E731
print("largest gaps (us: after kernel -> before kernel):")
for g, i in gaps[::-1][:12]:
    print(f"  {g:7.1f}  {short(call[i]):44s} -> {short(call[i + 1])}")
print(f"median gap {gaps[len(gaps) // 2][0]:.1f} us; sum of the 12 largest {sum(g for g, _ in gaps[::-1][:12]):.1f} us")
cp = []
for f in glob.glob(os.path.join(src, "**", "*memory_copy_trace.csv"), recursive=True):
    cp += list(csv.DictReader(open(f)))
print("memory copies in the whole run:", len(cp), "per call ~", len(cp) / 25)
PY

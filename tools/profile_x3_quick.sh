#!/bin/bash
# Quick counter pass of the BT_PREC_F32X3 forward (16 chunks): MFMA-busy, wave cycles, effective clock per kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_x3q
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
FWD="python $R/bench.py --workload forward --chunks 16 --prec f32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --watchdog 150"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- $FWD > $O/pmc_sq.log 2>&1; echo "pmc_sq $?"
cd $R
python - <<'PY'
import collections, csv, glob, os, re
src = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_x3q")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:64]
k = collections.OrderedDict(); dur = {}
for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        v = k.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0, 0.0]); v[0] += 1; v[1] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["kernel                                                            n   avg us  clk GHz  mfma %  wait_any % wait_inst %"]
for name, c in sorted(k.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    n = c["GRBM_GUI_ACTIVE"][0]; gui = c["GRBM_GUI_ACTIVE"][1] / n; d = sum(dur.get(name, [0])) / max(1, len(dur.get(name, [1])))
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n; wc = c["SQ_WAVE_CYCLES"][1] / n
    lines.append(f"{name:64s} {n:4d} {d / 1e3:8.1f} {gui / 8 / max(d, 1):8.3f} {100 * mf / max(gui / 8 * 1024, 1):7.2f} "
                 f"{100 * c['SQ_WAIT_ANY'][1] / n / max(wc, 1):10.1f} {100 * c['SQ_WAIT_INST_ANY'][1] / n / max(wc, 1):10.1f}")
open(os.path.join(src, "summary.txt"), "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY

#!/bin/bash
# Development: counters of the x3 attention kernels in isolation (tools/x3_probe.py launch shapes): MFMA-busy share, wave
# cycles, effective shader clock, instruction counts -- for the product build and any tools/variants/lib_NAME.so given.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_attn
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp BT_DEV=1
for l in "" "$@"; do
  if [ -n "$l" ]; then export BT_LIB_PATH=$R/tools/variants/lib_$l.so; else unset BT_LIB_PATH; fi
  tag=${l:-product}
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/$tag -o p --output-format csv -- python $R/tools/x3_probe.py 16 attn > $O/$tag.log 2>&1
  cd $R
  python - $O/$tag $tag <<'PY'
import collections, csv, glob, os, re, sys
src, tag = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:48]
k = collections.OrderedDict(); dur = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), r["Grid_Size"])
        v = k.setdefault(key, {}).setdefault(r["Counter_Name"], [0, 0.0]); v[0] += 1; v[1] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur.setdefault((short(r["Kernel_Name"]), r["Grid_Size"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"[{tag}] kernel / grid                                                   n   avg us  clk GHz  mfma %  wait_any % wait_inst %  valu/launch")
for key, c in k.items():
    if "attn" not in key[0]: continue
    n = c["GRBM_GUI_ACTIVE"][0]; gui = c["GRBM_GUI_ACTIVE"][1] / n; d = sum(dur.get(key, [0])) / max(1, len(dur.get(key, [1])))
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n; wc = c["SQ_WAVE_CYCLES"][1] / n
    print(f"{key[0]:48s} {key[1]:>9s} {n:4d} {d / 1e3:8.1f} {gui / 8 / max(d, 1):8.3f} {100 * mf / max(gui / 8 * 1024, 1):7.2f} "
          f"{100 * c['SQ_WAIT_ANY'][1] / n / max(wc, 1):10.1f} {100 * c['SQ_WAIT_INST_ANY'][1] / n / max(wc, 1):10.1f} {c['SQ_INSTS_VALU'][1] / n:12.0f}")
PY
done

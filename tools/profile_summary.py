#!/usr/bin/env python
"""Summarise a profile collection (tools/profile_r06.sh -> gpurun_out/prof_r06) into profiles/rNN_*: per-kernel trace
statistics, MFMA-busy / wave-cycle counters with the derived utilisation and effective shader clock, HBM traffic per
kernel (FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes, WRITE_SIZE as reported) and the rocm-smi power /
clock samples; plus the small json files bench.py reads for `roofline.traffic` and config 3's counter GB/s.
    python tools/profile_summary.py [gpurun_out/prof_r06] [round tag, default r06]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r06")
RND = sys.argv[2] if len(sys.argv) > 2 else "r06"
out_dir = os.path.join(ROOT, "profiles")
B = "python bench.py --no-cpu-baseline --no-extras --no-dist --min-seconds 0"
CMD = {"head": f"{B} --steps 5 --warmup 1 --prec half", "head1s": f"{B} --steps 5 --warmup 1 --prec half --streams 1   (the launch "
       "configuration of bench.py's roofline legs: whole 66-chunk launches on one stream)",
       "fwd": f"{B} --workload forward --chunks 16 --prec half --steps 10 --warmup 2",
       "head_x3_2s": f"{B} --steps 5 --warmup 1 --prec f32x3   (the headline's own launch configuration: 33-chunk slices on two streams)",
       "fwd_x3": f"{B} --workload forward --chunks 16 --prec f32x3 --steps 8 --warmup 2",
       "head_x3": f"{B} --steps 5 --warmup 1 --prec f32x3 --streams 1   (the f32x3 path's roofline launch shape)",
       "cfg3": f"{B} --workload forward --model small0 --prec f32 --chunks 128 --steps 2 --warmup 1"}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def read_trace(d):
    """kernel -> [durations ns]"""
    k = collections.OrderedDict()
    for f in glob.glob(os.path.join(src, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return k


def read_pmc(d):
    """(kernel) -> counter -> [launches, sum]"""
    k = collections.OrderedDict()
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            v = k.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0, 0.0])
            v[0] += 1
            v[1] += float(r["Counter_Value"])
    return k


def write(name, lines):
    open(os.path.join(out_dir, name), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))
    print(f"-> profiles/{name}\n")


CMD["fwd_x3_hidwrap"] = CMD["fwd_x3"] + "   [ABLATION build -DBT_ABL_HID_WRAP=2048: FF hidden activation kept in L2 / MALL, results garbage]"
for tag in ("head", "head1s", "fwd", "fwd_x3", "head_x3", "head_x3_2s", "fwd_x3_hidwrap"):
    tr = read_trace("trace_" + tag)
    if not tr:
        continue
    tot = sum(sum(v) for v in tr.values())
    lines = [f"# rocprofv3 --kernel-trace --stats -- {CMD[tag]}   (MI355X, round {RND[1:]}; durations in us)",
             f"{'kernel':72s} {'calls':>7s} {'total us':>12s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'%':>6s}"]
    for k, v in sorted(tr.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k:72s} {len(v):7d} {sum(v) / 1e3:12.1f} {sum(v) / len(v) / 1e3:10.2f} {min(v) / 1e3:9.2f} {max(v) / 1e3:9.2f} "
                     f"{100.0 * sum(v) / tot:6.2f}")
    write(f"{RND}_kernel_trace_{tag.replace('fwd_x3', 'x3')}.txt", lines)

for suffix, wl in (("", "BeatThis.forward, final0, 16 chunks, half operands"), ("_x3", "BeatThis.forward, final0, 16 chunks, BT_PREC_F32X3")):
    sq, sq2 = read_pmc("pmc_sq" + suffix), read_pmc("pmc_sq2" + suffix)
    tr = read_trace("pmc_sq" + suffix)   # (durations of the SAME run as the counters)
    if not sq:
        continue
    lines = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY",
             "#   SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE  (one pass)  +  SQ_INSTS_VALU SQ_INSTS_MFMA ... (second pass)",
             f"# workload: {wl} (bench.py --workload forward), values = averages per launch.",
             "# mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs); GRBM_GUI_ACTIVE counts shader-clock cycles while",
             "# the dispatch is active (summed over the 8 XCDs -> / 8): eff. clock = GUI_ACTIVE / 8 / duration.  SQ_*_CYCLES other than",
             "# MFMA_BUSY are quad-cycles (MI355X_MICROARCH.md).",
             f"{'kernel':60s} {'n':>4s} {'avg us':>8s} {'GUI_ACTIVE':>12s} {'clk GHz':>8s} {'MFMA_BUSY':>13s} {'mfma %':>7s} {'WAVE_CYC':>13s} "
             f"{'wait_any %':>10s} {'wait_inst %':>11s} {'INSTS_MFMA':>11s} {'INSTS_VALU':>11s}"]
    for k, c in sorted(sq.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0])[1]):
        n = c["GRBM_GUI_ACTIVE"][0]
        g = c["GRBM_GUI_ACTIVE"][1] / n
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [1, 0])[1] / n
        wc = c.get("SQ_WAVE_CYCLES", [1, 0])[1] / n
        wa = c.get("SQ_WAIT_ANY", [1, 0])[1] / n
        wi = c.get("SQ_WAIT_INST_ANY", [1, 0])[1] / n
        dur = sum(tr[k]) / len(tr[k]) / 1e3 if k in tr else float("nan")
        c2 = sq2.get(k, {})
        im = c2.get("SQ_INSTS_MFMA", [1, 0])
        iv = c2.get("SQ_INSTS_VALU", [1, 0])
        lines.append(f"{k[:60]:60s} {n:4d} {dur:8.2f} {g:12.0f} {g / 8 / (dur * 1e3) if dur == dur else 0:8.3f} {mf:13.0f} "
                     f"{100 * mf / (g / 8 * 1024) if g else 0:7.2f} {wc:13.0f} {100 * wa / wc if wc else 0:10.1f} {100 * wi / wc if wc else 0:11.1f} "
                     f"{im[1] / max(im[0], 1):11.0f} {iv[1] / max(iv[0], 1):11.0f}")
    write(f"{RND}_pmc_mfma{suffix}.txt", lines)

CATS = {"attn_flash": ("attn_frag_kernel", "attn_frag_x3_kernel", "attn_frag_x3q2_kernel", "attn_flash"), "layer_tail": ("layer_tail_kernel",)}
for tag, wl, jname, meta in (("", "BeatThis.forward, final0, 16 chunks, half operands", "pmc_traffic.json", {"model": "final0", "prec": "half", "chunks": 16}),
                             ("_x3", "BeatThis.forward, final0, 16 chunks, BT_PREC_F32X3", "pmc_traffic_f32x3.json", {"model": "final0", "prec": "f32x3", "chunks": 16}),
                             ("_cfg3", "BASELINE config 3: BeatThis.forward, small0, exact fp32, 128 chunks", "pmc_traffic_cfg3.json", {"model": "small0", "prec": "f32", "chunks": 128}),
                             ("_x3_hidwrap", "ABLATION build -DBT_ABL_HID_WRAP=2048 (FF hidden activation kept in L2 / MALL; results garbage): BeatThis.forward, final0, 16 chunks, BT_PREC_F32X3", None, None),
                             ("_head", "headline (6 x 300 s tracks through Audio2Beats, default precision)", None, None)):
    fetch, wr = read_pmc("pmc_fetch" + tag), read_pmc("pmc_write" + tag)
    if not fetch:
        continue
    lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (they do not fit one pass).",
             "# Counter unit = KB.  On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM",
             "# section): 'fetch MB' below is 2 x FETCH_SIZE; WRITE_SIZE is uncalibrated and shown as reported.  Averages per launch.",
             f"# workload: {wl}",
             f"{'kernel':72s} {'launches':>8s} {'fetch MB (x2)':>14s} {'write MB':>10s} {'total MB/launch':>16s}"]
    rows = []
    for k, c in fetch.items():
        n, fs = c["FETCH_SIZE"]
        w = wr.get(k, {}).get("WRITE_SIZE", [1, 0.0])
        rows.append((k, n, 2 * fs / n / 1e3, w[1] / max(w[0], 1) / 1e3))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    for k, n, f, w in rows:
        lines.append(f"{k:72s} {n:8d} {f:14.2f} {w:10.2f} {f + w:16.2f}")
    total = sum((f + w) * n for _, n, f, w in rows)
    # forwards in the run: launches of the head kernel (one per forward)
    n_fwd = max([n for k, n, _, _ in rows if k.startswith("head_kernel")] + [1])
    lines.append(f"# sum over all launches of the run: {total / 1e3:.2f} GB = {total / n_fwd / 1e3:.3f} GB per forward ({n_fwd} forwards)")
    write(f"{RND}_pmc_hbm_traffic{tag}.txt", lines)
    if jname:
        js = {"workload": meta, "bytes_per_forward": int(1e6 * total / n_fwd),
              "source": f"profiles/{RND}_pmc_hbm_traffic{tag}.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; fetch x2 on gfx950)"}
        for cat, prefixes in CATS.items():
            sel = [(n, f + w) for k, n, f, w in rows if k.startswith(prefixes)]
            if sel:
                launches = sum(n for n, _ in sel)
                js[cat] = {"bytes_per_launch": int(1e6 * sum(n * b for n, b in sel) / launches), "launches_per_forward": launches // n_fwd}
        json.dump(js, open(os.path.join(out_dir, jname), "w"), indent=1)

smi = []
for name in ("smi_idle.txt", "smi_fwd.txt", "smi_fwd_x3.txt", "smi_head.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        smi += [f"==== {name} (rocm-smi --showpower --showclocks, sampled every 0.5 s while the workload loops)"] + \
               [l.rstrip() for l in open(f) if l.strip()]
if smi:
    write(f"{RND}_power_clocks.txt", smi)

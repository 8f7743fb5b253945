#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md's HBM section prescribes) into profiles/r01_pmc_hbm_traffic.txt and profiles/r01_pmc_traffic.json
(the `traffic` figure of bench.py's roofline object: HBM bytes per launch of the attention category).
    python tools/pmc_summary.py [gpurun_out]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)IDF16bLi(\d+)E", name)
    return f"{m.group(1)}<bf16,{m.group(2)}>" if m else name[:60]


def collect(d, counter):
    """(kernel, grid) -> [launches, sum of counter]"""
    out = collections.OrderedDict()
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0)
            v = out.setdefault(key, [0, 0.0])
            v[0] += 1
            v[1] += float(r["Counter_Value"])
    return out


fetch, write = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
rows = []
for key, (n, fs) in fetch.items():
    wn, ws = write.get(key, [0, 0.0])
    rows.append((key[0], key[1], n, fs / n, ws / wn if wn else float("nan")))
rows.sort(key=lambda r: -r[3])
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes), python bench.py --steps 3 --warmup 1",
         "# MI355X, final0 bf16, 16 chunks.  Counter unit = KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports",
         "# half of the bytes of wide coalesced reads: the 'fetch MB (x2)' column is the corrected figure; WRITE_SIZE is uncalibrated.",
         f"{'kernel':60s} {'grid':>10s} {'launches':>8s} {'FETCH_SIZE KB':>14s} {'fetch MB (x2)':>14s} {'WRITE_SIZE KB':>14s}"]
for k, g, n, f, w in rows:
    lines.append(f"{k:60s} {g:10d} {n:8d} {f:14.0f} {2 * f / 1e3:14.1f} {w:14.0f}")
open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
attn = [(n, 2e3 * f + 1e3 * w) for k, g, n, f, w in rows if k.startswith("attn_frag_kernel")]
launches = sum(n for n, _ in attn)
per_launch = sum(n * b for n, b in attn) / launches
js = {"workload": {"model": "final0", "prec": "bf16", "chunks": 16},
      "source": "profiles/r01_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; fetch x2 on gfx950)",
      "attn_flash": {"bytes_per_launch": int(round(per_launch)), "launches": 9,
                     # K + V + Q fragment streams read once, bf16 output written once: 3 x 512 x 48 x 2 KB + 512 x 1500 x 64 B (front),
                     # half of that for the main-layer shape; weighted 3 : 6
                     "algorithmic_bytes_per_launch": 137625600}}
json.dump(js, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines[:12]))
print(js["attn_flash"])

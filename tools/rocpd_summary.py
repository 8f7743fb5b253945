#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace) into per-kernel stats and,
optionally, the ordered launch list of the last forward.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db [--launches N]"""
import re
import sqlite3
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
    stats = OrderedDict()
    for n, s, e, *_ in rows:
        d = stats.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(v[1] for v in stats.values())
    print(f"{'kernel':110s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {v[0]:6d} {v[1]:10.1f} {v[1] / v[0]:9.1f} {v[2]:9.1f} {v[3]:9.1f} {100 * v[1] / tot:6.2f}")
    if "--launches" in sys.argv:
        n = int(sys.argv[sys.argv.index("--launches") + 1])
        print("\nlast", n, "launches (us, grid):")
        for nme, s, e, gx, gy, gz in rows[-n:]:
            print(f"{(e - s) / 1e3:9.1f}  {gx}x{gy}x{gz}  {short(nme)}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite outputs (<src>/<pass>/**/*_results.db) into the small text summaries that are
committed under profiles/.  Usage: python profiles/summarize_rocpd.py <src dir> <tag> [--out <dir>]  (default: profiles/)"""
import glob
import json
import os
import sqlite3
import sys


def main(src: str, tag: str, out: str = None):
    here = out or os.path.dirname(os.path.abspath(__file__))
    lines = []
    pmc = {}
    for d in sorted(os.listdir(src)):
        dbs = glob.glob(os.path.join(src, d, "**", "*_results.db"), recursive=True)
        if not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels "
                         "group by name order by sum(duration) desc").fetchall()
        lines.append(f"== pass {d}: kernel durations (ns) [rocprofv3 --kernel-trace]")
        lines.append(f"{'calls':>6} {'avg_ns':>12} {'min_ns':>12} {'max_ns':>12} {'total_ns':>14}  kernel")
        for name, n, avg, mn, mx, tot in rows:
            lines.append(f"{n:6d} {avg:12.0f} {mn:12.0f} {mx:12.0f} {tot:14.0f}  {name}")
        try:
            prow = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                             "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except sqlite3.OperationalError:
            prow = []
        if prow:
            lines.append(f"-- pass {d}: PMC counters, mean per dispatch")
            for kn, cn, v, n in prow:
                lines.append(f"{v:18.1f}  {cn:28s} n={n:<4d} {kn}")
                pmc.setdefault(kn, {})[cn] = v
        lines.append("")
    with open(os.path.join(here, f"{tag}_rocprof_summary.txt"), "w") as f:
        f.write("\n".join(lines))
    with open(os.path.join(here, f"{tag}_pmc.json"), "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)

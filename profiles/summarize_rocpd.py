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
    per_pass = {}   # pass name -> (dominant kernel, {counter: mean per dispatch of that kernel})
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
        if rows:
            dom = rows[0][0]
            per_pass[d] = (dom, {cn: v for kn, cn, v, n in prow if kn == dom}, rows[0][2])
        lines.append("")
    with open(os.path.join(here, f"{tag}_rocprof_summary.txt"), "w") as f:
        f.write("\n".join(lines))
    with open(os.path.join(here, f"{tag}_pmc.json"), "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    print("\n".join(lines))
    # HBM traffic of the dominant kernel per workload (what bench.py's roofline.traffic quotes), with the kernel it was measured on:
    # bench.py prints the figure only while that is still the kernel it launches.  gfx950: FETCH_SIZE counts 64 B per 128-B
    # request on wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); both counters are in KiB.
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flashattention.c_amd", "libflashattn_amd.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    traffic = {"_tag": tag,
               "lib_sha256": sha,   # the binary these passes ran on: bench.py prints `traffic` only while it loads this very library
               "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py [--workload c3] --steps 20 --warmup 3 "
                          "--no-cpu-baseline --no-extras` (profiles/collect.sh)",
               "_method": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch of the dominant kernel"}
    bad = []
    for wl, fpass, wpass, line_file in (("c4", "pmc_fetch", "pmc_write", "bench_line.json"), ("c3", "pmc_c3_fetch", "pmc_c3_write", "bench_line_c3.json"),
                                        ("c4acc", "pmc_acc_fetch", "pmc_acc_write", "bench_line_accurate.json")):
        if fpass in per_pass and wpass in per_pass:
            kf, cf, _ = per_pass[fpass]
            kw, cw, _ = per_pass[wpass]
            base = kf.split("<")[0].split("::")[-1].replace("void ", "").strip()
            traffic[f"{wl}_kernel"] = base
            traffic[f"{wl}_kernel_full"] = kf
            traffic[f"{wl}_fetch_size_kb"] = cf.get("FETCH_SIZE")
            traffic[f"{wl}_write_size_kb"] = cw.get("WRITE_SIZE")
            if cf.get("FETCH_SIZE") is not None and cw.get("WRITE_SIZE") is not None and kf == kw:
                traffic[f"{wl}_hbm_bytes_per_launch"] = int((2.0 * cf["FETCH_SIZE"] + cw["WRITE_SIZE"]) * 1024)
            # the kernel the profile saw must be the kernel the bench line names (fa_kernel_name_for)
            lf = os.path.join(src, line_file)
            if os.path.exists(lf):
                try:
                    named = json.loads(open(lf).read().strip().splitlines()[-1])["roofline"]["kernel"]
                    if named != base:
                        bad.append(f"{wl}: profile's dominant kernel {base!r} != bench line's roofline.kernel {named!r}")
                except Exception as e:   # pragma: no cover
                    bad.append(f"{wl}: could not read {lf}: {e!r}")
    with open(os.path.join(here, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    if bad:
        print("KERNEL NAME MISMATCH:\n  " + "\n  ".join(bad), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)

#!/usr/bin/env python3
"""profiles/dispatch_sweep_report.py FILE [TOL] -- read the lines of profiles/dispatch_sweep.sh and list the shapes where the dispatch
(variant 0, or kernel "auto") is more than TOL (default 4 %) behind the best explicit tiling of the same shape."""
import collections
import sys


def main(path, tol=0.04):
    rows = collections.OrderedDict()
    for line in open(path):
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        try:
            if "v" in p[2:]:                       # dtype d D bh BH n N c C v V ms
                key = (p[0], int(p[2]), int(p[4]), int(p[6]), int(p[8]))
                name, ms = ("auto" if p[10] == "0" else "v" + p[10]), float(p[11])
            else:                                   # dtype kernel d D bh BH n N c C ms
                key = (p[0], int(p[3]), int(p[5]), int(p[7]), int(p[9]))
                name, ms = p[1], float(p[10])
        except (ValueError, IndexError):
            continue
        rows.setdefault(key, {})[name] = ms
    misses = 0
    for key, v in rows.items():
        others = [(ms, name) for name, ms in v.items() if name != "auto"]
        if "auto" not in v or not others:
            continue
        best = min(others)
        if v["auto"] > best[0] * (1.0 + tol):
            misses += 1
            print(f"{key[0]} d={key[1]} bh={key[2]} n={key[3]} causal={key[4]}  " + "  ".join(f"{a}={b:.4f}" for a, b in v.items()) +
                  f"   <-- dispatch {v['auto'] / best[0]:.2f}x of {best[1]}")
    print(f"{len(rows)} shapes, {misses} where the dispatch is more than {100 * tol:.0f} % behind the best tiling")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.04)

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name (+ grid size) the
call count, average / total duration.  Usage: prof_summary.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
    agg = OrderedDict()
    for r in rows:
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""))
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print("%-72s %9s %6s %10s %10s %6s" % ("kernel", "grid", "calls", "avg_us", "total_ms", "%"))
    for (n, gx, gy), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %9s %6d %10.1f %10.3f %6.1f" % (n, "%sx%s" % (gx, gy), c, t / c / 1e3, t / 1e6, 100.0 * t / tot))
    print("total %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))


if __name__ == "__main__":
    main()

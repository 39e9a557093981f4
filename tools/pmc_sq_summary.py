#!/usr/bin/env python3
"""Mean SQ counters per kernel (+grid) from a rocprofv3 --pmc counter_collection CSV joined with the
kernel trace of the same run (durations), plus the MFMA pipe utilisation they imply:
    util = SQ_VALU_MFMA_BUSY_CYCLES / (duration * 2.4 GHz * 1024 SIMDs)
Usage: pmc_sq_summary.py <counter_collection.csv> <kernel_trace.csv>"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:44]


def main():
    dur = {}
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = OrderedDict()
    names = []
    for r in csv.DictReader(open(sys.argv[1])):
        k = (short(r["Kernel_Name"]), r["Grid_Size"])
        a = agg.setdefault(k, {"n": defaultdict(int), "v": defaultdict(float), "d": 0.0, "nd": 0, "seen": set()})
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        a["n"][c] += 1
        a["v"][c] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in a["seen"]:
            a["seen"].add(r["Dispatch_Id"])
            a["d"] += dur.get(r["Dispatch_Id"], 0)
            a["nd"] += 1
    print("%-44s %8s %9s " % ("kernel", "grid", "avg_us") + " ".join("%12s" % n.replace("SQ_", "")[:12] for n in names)
          + "   MFMA_util")
    for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1]["d"]):
        us = a["d"] / max(a["nd"], 1) / 1e3
        vals = [a["v"][n] / max(a["n"][n], 1) for n in names]
        mf = a["v"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(a["n"].get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1)
        util = mf / (us * 2400.0 * 1024.0) if us > 0 else 0.0
        print("%-44s %8s %9.1f " % (k, g, us) + " ".join("%12.3g" % v for v in vals) + "   %8.1f%%" % (100 * util))


if __name__ == "__main__":
    main()

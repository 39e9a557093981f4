#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

MI355X_MICROARCH.md "HBM": counters are in KiB-like units of 1024 B
(hbm_bytes = counter * 1024); on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of a wide coalesced streaming read -> doubled here; WRITE_SIZE is exact
for 16-B-per-lane stores.  Other access widths are uncalibrated (ratios only).

    pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = (short(r["Kernel_Name"]), r["Grid_Size"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    print("%-60s %10s %6s %14s %14s" % ("kernel", "grid", "calls", "read MB/launch", "write MB/launch"))
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [1, 0])[1] + w.get(k, [1, 0])[1])):
        nf, vf = f.get(k, [1, 0.0])
        nw, vw = w.get(k, [1, 0.0])
        rd = 2.0 * vf * 1024 / max(nf, 1)  # gfx950 FETCH_SIZE = 1/2 of a wide coalesced stream
        wr = vw * 1024 / max(nw, 1)
        out["%s|%s" % k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
        print("%-60s %10s %6d %14.2f %14.2f" % (k[0][:60], k[1], max(nf, nw), rd / 1e6, wr / 1e6))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

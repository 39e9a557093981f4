#!/usr/bin/env python3
"""One replayed training step out of a rocprofv3 --kernel-trace CSV of `bench.py ... --train-steps N`: kernels per step,
span, busy time, and the kernels grouped by name.   python tools/train_step_trace.py <kernel_trace.csv> [top]"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step = from one forward launch of the fused head + loss kernel to the next
    idx = [i for i, r in enumerate(rows) if re.search(r"head_ce_kernel<4, (0|false)>", r["Kernel_Name"])]
    if len(idx) < 3:
        raise SystemExit("fewer than three training steps in the trace")
    a, b = idx[-3], idx[-2]
    step = rows[a:b]
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for r in step:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)[:64]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        x = agg.setdefault(n, [0, 0])
        x[0] += 1
        x[1] += d
    busy = sum(v[1] for v in agg.values()) / 1e3
    print("one replayed training step: %d kernels, span %.1f us, busy %.1f us" % (len(step), span, busy))
    print("%-66s %5s %9s %9s" % ("kernel", "calls", "avg_us", "total_us"))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-66s %5d %9.1f %9.1f" % (n, c, t / c / 1e3, t / 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-kernel table from a rocprofv3 --pmc <COUNTER> --kernel-trace counter_collection CSV:
    python tools/pmc_table.py counter_collection.csv COUNTER out.csv
-> rows  kernel, dispatches, summed counter value, summed duration (ms)."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):               # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    f, c, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[1] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a[0] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    with open(out, "w") as o:
        for k, (n, v, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            o.write('"%s",%d,%.6g,%.4f\n' % (k, n, v, ms))


if __name__ == "__main__":
    main()

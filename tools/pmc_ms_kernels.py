#!/usr/bin/env python
"""Print counters + duration of every ms_iterate* dispatch in a rocprofv3 counter_collection CSV."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "ms_iterate" not in k:
        continue
    kind = "bounds" if "bounds" in k else ("sparse" if "Lb1" in k or "<true>" in k else "dense")
    key = (int(r["Dispatch_Id"]), kind)
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for key in sorted(agg):
    print(key, "ms %.1f" % dur[key], {c: "%.3g" % v for c, v in agg[key].items()})

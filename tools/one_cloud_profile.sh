#!/bin/bash
# kernel time vs wall time at one cloud per call: rocprofv3 --kernel-trace --stats of tools/one_cloud_latency.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/one_cloud_latency.py 8 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oc -- python $R/tools/one_cloud_latency.py 8 > /tmp/oc.out 2>&1
K=$(find /tmp/oc -name "*kernel_stats.csv" | head -1)
python - "$K" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
n = sum(int(r["Calls"]) for r in rows)
print(f"kernel time {tot:.1f} ms over {n} launches in 32 pipeline calls (16 network-embedding + 16 planted): {tot / 32:.2f} ms, {n / 32:.0f} launches per call")
for r in rows[:22]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']) / 32:7.1f} per call {float(r['TotalDurationNs']) / 32e3:9.1f} us per call")
PY

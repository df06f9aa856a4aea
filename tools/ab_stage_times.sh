#!/bin/bash
# A/B of two builds of libsedhip.so on one box: per-kernel averages of tools/stage_times.py under rocprofv3 --kernel-trace --stats.
# usage (gpurun): bash tools/ab_stage_times.sh [script.py args...]   (old build = sed-net_amd/sednet_hip/libsedhip_old.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-tools/stage_times.py}; shift
for L in old new; do
  if [ $L = old ]; then export SEDHIP_LIB=$R/sed-net_amd/sednet_hip/libsedhip_old.so; else unset SEDHIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$L -- python $R/$S "$@" > /tmp/ab_$L.out 2>&1
  K=$(find /tmp/ab_$L -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  python - "$K" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 16:
        print(f"{r['Name'][:86]:86s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:10.1f} us")
PY
done

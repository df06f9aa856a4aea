"""kernel_stats.csv of rocprofv3 -> table per pass: python tools/kernel_stats_table.py <csv> <passes> [top]"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_table import short
rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print(f"kernel time per pass {tot / passes:.2f} ms")
for r in rows[:top]:
    t = float(r["TotalDurationNs"]) / 1e6
    print(f"{short(r['Name'])[:84]:84s} {int(r['Calls']) / passes:7g} {float(r['AverageNs']) / 1e6:9.3f} {t / passes:8.2f} {100 * t / tot:6.2f}")

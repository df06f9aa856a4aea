#!/usr/bin/env python
"""Per-kernel PMC table (markdown) from one rocprofv3 --pmc ... --kernel-trace counter_collection CSV of a bench step:
    python tools/pmc_kernels.py counter_collection.csv out.md
Columns: time, MFMA-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)), non-MFMA VALU instructions per
MFMA, LDS bank-conflict cycles per LDS instruction, effective clock."""
import collections, csv, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from pmc_table import short

f, out = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k))
        agg[k]["_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        agg[k]["_n"] += 1
rows = ["| kernel | dispatches | total ms | MFMA-pipe busy | VALU per MFMA | LDS conflict cycles per LDS instr | clock GHz |",
        "|---|---:|---:|---:|---:|---:|---:|"]
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["_ms"]):
    if c["_ms"] < 0.3:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8) if gui else float("nan")
    mf = c.get("SQ_INSTS_MFMA", 0.0)
    vpm = (c.get("SQ_INSTS_VALU", 0.0) - mf) / mf if mf else float("nan")
    lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_INSTS_LDS"] if c.get("SQ_INSTS_LDS") else float("nan")
    clk = gui / 8 / (c["_ms"] * 1e-3) / 1e9 if gui else float("nan")
    rows.append(f"| `{k[:64]}` | {int(c['_n'])} | {c['_ms']:.2f} | {busy:.3f} | {vpm:.2f} | {lds:.3f} | {clk:.2f} |")
open(out, "w").write("\n".join(rows) + "\n")
print("\n".join(rows))

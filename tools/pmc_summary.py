#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel into markdown.
Usage: python tools/pmc_summary.py KERNEL_SUBSTR out.md dir1 [dir2 ...]"""
import collections
import csv
import glob
import sys


def main():
    kern, out = sys.argv[1], sys.argv[2]
    vals = collections.OrderedDict()
    meta = {}
    for d in sys.argv[3:]:
        for f in glob.glob(d + "/*counter_collection.csv"):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                    meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                              "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
                    meta["duration_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            last = per[sorted(per, key=int)[-1]]       # last dispatch = warmed up
            vals.update(last)
    lines = [f"| counter (summed over the chip, last `{kern}` dispatch) | value |", "|---|---:|"]
    lines += [f"| {k} | {v:.6g} |" for k, v in vals.items()]
    lines.append("")
    lines.append("dispatch: " + ", ".join(f"{k}={v}" for k, v in meta.items()))
    d = vals
    if "GRBM_GUI_ACTIVE" in d:
        clk = d["GRBM_GUI_ACTIVE"] / 8 / (meta["duration_ms"] * 1e-3) / 1e9
        lines.append(f"\neffective clock ~ GRBM_GUI_ACTIVE / 8 XCDs / duration = {clk:.2f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_INSTS_MFMA" in d:
        lines.append(f"cycles per MFMA = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_INSTS_MFMA']:.1f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        lines.append(f"MFMA pipe utilisation = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE/8) = "
                     f"{d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8):.3f}")
    if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d:
        lines.append(f"non-MFMA VALU instructions per MFMA = {(d['SQ_INSTS_VALU'] - d['SQ_INSTS_MFMA']) / d['SQ_INSTS_MFMA']:.2f}")
    text = "\n".join(lines)
    print(text)
    open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()

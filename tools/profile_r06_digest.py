#!/usr/bin/env python
"""gpurun_out/prof_r06/ (tools/profile_r06.sh) -> the tracked summaries under profiles/:  python tools/profile_r06_digest.py [commit]"""
import collections, csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
O, P = os.path.join(ROOT, "gpurun_out", "prof_r06"), os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_table import short
HEAD = "python bench.py --no-extra-legs --no-k64 --no-cpu-baseline"
KERNEL = "ms_sparse_f16_kernel"


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)


def stats_md(name, passes, title, head, top=34):
    rows = list(csv.DictReader(open(os.path.join(O, f"{name}_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    out = [f"# {title} (1 x MI355X, round 6, commit {commit})", "", head, "",
           f"{passes} passes traced; kernel time per pass {tot / passes:.1f} ms.", "",
           "| kernel | calls per pass | avg ms | ms per pass | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:top]:
        t = float(r["TotalDurationNs"]) / 1e6
        out.append(f"| `{short(r['Name'])[:80]}` | {int(r['Calls']) / passes:g} | {float(r['AverageNs']) / 1e6:.3f} | "
                   f"{t / passes:.2f} | {100 * t / tot:.2f} |")
    return "\n".join(out) + "\n"


def sq_table(files, min_ms=0.3):
    """per-kernel table from one or more SQ counter CSVs of the same command (scratch size from the dispatch records)"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for i, fn in enumerate(files):
        seen = set()
        for r in csv.DictReader(open(os.path.join(O, fn))):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["_scratch"] = max(agg[k]["_scratch"], float(r["Scratch_Size"]))
            if i == 0 and (r["Dispatch_Id"], k) not in seen:
                seen.add((r["Dispatch_Id"], k))
                agg[k]["_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                agg[k]["_n"] += 1
    two = len(files) > 1
    rows = ["| kernel | dispatches | total ms | scratch B/lane | MFMA-pipe busy | VALU per MFMA | LDS conflict cycles per LDS instr | clock GHz |"
            + (" waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | LDS index unit active |" if two else ""),
            "|---|---:|---:|---:|---:|---:|---:|---:|" + ("---:|---:|" if two else "")]
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["_ms"]):
        if c["_ms"] < min_ms:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8) if gui else float("nan")
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        vpm = (c.get("SQ_INSTS_VALU", 0.0) - mf) / mf if mf else float("nan")
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_INSTS_LDS"] if c.get("SQ_INSTS_LDS") else float("nan")
        clk = gui / 8 / (c["_ms"] * 1e-3) / 1e9 if gui else float("nan")
        line = f"| `{k[:64]}` | {int(c['_n'])} | {c['_ms']:.2f} | {int(c['_scratch'])} | {busy:.3f} | {vpm:.2f} | {lds:.3f} | {clk:.2f} |"
        if two:
            wait = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
            # SQ_LDS_IDX_ACTIVE: cycles per CU; launch cycles of the first pass at its clock
            ldsa = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256 / (gui / 8) if gui else float("nan")
            line += f" {wait:.3f} | {ldsa:.3f} |"
        rows.append(line)
    return "\n".join(rows)


b = last_json(os.path.join(O, "bench.out"))
rf = b["roofline"]
open(os.path.join(P, "r06_bench_kernel_stats.md"), "w").write(stats_md(
    "bench", 4, f"rocprofv3 --kernel-trace --stats of `{HEAD} --steps 3 --warmup 1` (the headline leg alone)",
    f"Bench line of the profiled run: {b['value']} clouds/s, {b['ms_per_step']} ms per 64-cloud step, stages {json.dumps(b['stages_ms_per_step'])}. "
    f"roofline.avg_launch_ms = {rf['avg_launch_ms']} (events inside bench.py around the C call: the counting launch of ~0.5 ms, the iteration launch, "
    f"the two split kernels and the item sort) against the rocprofv3 rows of `{KERNEL}` below (2 calls per pass: avg ms x 2 = that "
    f"kernel's time per pass). roofline.frac = {rf['frac']} = executed fp16-MFMA flops ({rf['achieved']} TFLOP/s) / 2500."))

hbm = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: [0.0, 0.0])
    for r in csv.DictReader(open(os.path.join(O, f"bench_{C}.csv"))):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == C:
            per[r["Dispatch_Id"]][0] += float(r["Counter_Value"])
            per[r["Dispatch_Id"]][1] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    main = [v for v in per.values() if v[1] > 20.0]                       # the iteration launches (the counting launches take < 1 ms)
    hbm[C] = (sum(v[0] for v in main) / len(main), sum(v[1] for v in main) / len(main), len(main))
rec = {"kernel": rf["kernel"], "schedule": "block-sparse split-fp16, persistent", "clouds": 64, "iterations": 50,
       "FETCH_SIZE_KB": hbm["FETCH_SIZE"][0], "WRITE_SIZE_KB": hbm["WRITE_SIZE"][0],
       "launch_ms_under_counters": hbm["FETCH_SIZE"][1], "launches_averaged": hbm["FETCH_SIZE"][2],
       "hbm_bytes_per_launch": (2 * hbm["FETCH_SIZE"][0] + hbm["WRITE_SIZE"][0]) * 1024,
       "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read, "
                  "MI355X_MICROARCH.md section HBM; Infinity-Cache hits are counted)",
       "algorithmic_bytes_per_launch": 64 * 2 * 10000 * 128 * 4,
       "command": f"{HEAD} --steps 1 --warmup 1", "commit": commit, "date": "round 6, tools/profile_r06.sh"}
json.dump(rec, open(os.path.join(P, "r06_pmc_ms_iterate.json"), "w"), indent=1)

with open(os.path.join(P, "r06_pmc_kernels.md"), "w") as f:
    f.write(f"# Per-kernel PMC table of the headline leg (1 x MI355X, round 6, commit {commit})\n\n"
            f"Two counter-only runs of `{HEAD} --steps 1 --warmup 1` (`rocprofv3 --pmc ... --kernel-trace`; two passes of the step each): "
            f"(1) SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE, "
            f"(2) SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM. MFMA-pipe busy = "
            f"MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8); scratch = Scratch_Size of the dispatch record. Durations under counter collection are "
            f"longer than in the kernel-trace table.\n\n" + sq_table(["bench_sq.csv", "bench_sq2.csv"]) + "\n\n"
            f"## HBM traffic of the iteration kernel (separate FETCH_SIZE and WRITE_SIZE passes of the same command)\n\n"
            f"`{rf['kernel']}`, iteration launches only: FETCH_SIZE {hbm['FETCH_SIZE'][0]:.0f} KB, WRITE_SIZE {hbm['WRITE_SIZE'][0]:.0f} KB per "
            f"launch of 64 clouds -> (2 x FETCH + WRITE) x 1024 = {rec['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch against "
            f"{rec['algorithmic_bytes_per_launch'] / 1e9:.3f} GB of algorithmic bytes (X in, new X out): the stage images are re-read from "
            f"L2 / Infinity Cache / HBM by every workgroup and iteration ({rec['hbm_bytes_per_launch'] / 1e9 / (hbm['FETCH_SIZE'][1] * 1e-3) / 1e3:.2f} "
            f"TB/s of 8). WRITE_SIZE = {hbm['WRITE_SIZE'][0] * 1024 / (64 * 10000 * 128 * 4):.1f} x the 0.33 GB output: SCRATCH traffic of the row update -- the "
            f"kernel's ~150 B per lane are spilled and reloaded once per iteration around the update of the 64 accumulator + 64 operand registers; no "
            f"scratch instruction sits in a block that holds an MFMA (tools/isa_blocks.py). The spill cannot move into AGPRs: at two waves per SIMD a wave "
            f"owns 256 of the 512 unified registers, VGPRs and AGPRs together (DESIGN.md section 8.2).\n")

h = last_json(os.path.join(O, "hpnet.out"))
open(os.path.join(P, "r06_hpnet_leg_kernel_stats.md"), "w").write(stats_md(
    "hpnet", 4, f"rocprofv3 --kernel-trace --stats of `{HEAD} --hpnet --steps 3 --warmup 1` (the reference script's DEFAULT flow: HPNet stage on)",
    f"Bench line of the profiled run: {h['value']} clouds/s, {h['ms_per_step']} ms per 64-cloud step, stages {json.dumps(h['stages_ms_per_step'])}; "
    f"mean-shift schedule per step {json.dumps(h['mean_shift_schedule'])}; roofline block of the d = 160 iteration kernel: "
    f"{json.dumps({k: h['roofline'].get(k) for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms', 'executed_share_of_dense_work')})}.", top=44)
    + "\n## SQ counters of the same flow (own run, `--steps 1 --warmup 1`)\n\n" + sq_table(["hpnet_sq.csv"], min_ms=1.0) + "\n")

d = last_json(os.path.join(O, "bench_default.out"))
if d:
    if d["roofline"]["kernel"].split("<")[0] == rec["kernel"].split("<")[0] and int(d["roofline"]["clouds_per_launch"]) == rec["clouds"]:
        d["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
        d["roofline"]["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md) of "
                                           f"`{rec['command']}` at commit {rec['commit']}, {rec['date']}; a profile record, not re-measured inside this run")
    json.dump(d, open(os.path.join(P, "r06_bench_line.json"), "w"), indent=1)
print("written:", sorted(x for x in os.listdir(P) if x.startswith("r06")))

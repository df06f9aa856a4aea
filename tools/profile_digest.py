#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> the tracked summaries under profiles/:  python tools/profile_digest.py r02 [commit]"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
O = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
P = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_table import short


def stats_md(name, passes, title, head, top=28):
    rows = list(csv.DictReader(open(os.path.join(O, f"{name}_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    out = [f"# {title} (1 x MI355X, round 2, commit {commit})", "", head, "",
           f"{passes} passes traced; kernel time per pass {tot / passes:.1f} ms.", "",
           "| kernel | calls per pass | avg ms | ms per pass | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:top]:
        t = float(r["TotalDurationNs"]) / 1e6
        out.append(f"| `{short(r['Name'])[:80]}` | {int(r['Calls']) / passes:g} | {float(r['AverageNs']) / 1e6:.3f} | "
                   f"{t / passes:.2f} | {100 * t / tot:.2f} |")
    return "\n".join(out) + "\n"


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


b = last_json(os.path.join(O, "bench.out"))
r = last_json(os.path.join(O, "real.out"))
open(os.path.join(P, f"{tag}_bench_kernel_stats.md"), "w").write(stats_md(
    "bench", 4, "rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-k64 --no-realistic`",
    f"Bench line of the profiled run: {b['value']} clouds/s, {b['ms_per_step']} ms per 64-cloud step; roofline.avg_launch_ms "
    f"{b['roofline']['avg_launch_ms']} (events inside bench.py) vs the rocprofv3 average of the iteration kernel below."))
open(os.path.join(P, f"{tag}_bench_realistic_kernel_stats.md"), "w").write(stats_md(
    "real", 6, "rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-k64` (headline + realistic leg)",
    f"3 headline passes + 3 passes of the realistic leg (planted segments: {r['realistic']['value']} clouds/s, "
    f"{r['realistic']['ms_per_step']} ms per step; schedule {json.dumps(r['realistic']['mean_shift_schedule'], default=str)[:300]})."))
ms_out = open(os.path.join(O, "msstage.out")).read().strip().splitlines()[-1]
open(os.path.join(P, f"{tag}_clustering_stage_kernel_stats.md"), "w").write(stats_md(
    "msstage", 3, "rocprofv3 --kernel-trace --stats of `python tools/ms_stage_only.py 64 2` (guarded mean-shift on the bench's planted embedding)",
    f"`{ms_out}` (under the profiler)."))
tr_out = open(os.path.join(O, "train.out")).read().strip().splitlines()[-1]
open(os.path.join(P, f"{tag}_train_step_kernel_stats.md"), "w").write(stats_md(
    "train", 5, "rocprofv3 --kernel-trace --stats of `python tools/train_bench.py 32 10000 64 3 --bf16`",
    f"`{tr_out}` (under the profiler; 2 warm-up + 3 timed steps)."))
with open(os.path.join(P, f"{tag}_pmc_kernels.md"), "w") as f:
    f.write(f"# Per-kernel PMC table (1 x MI355X, round 2, commit {commit})\n\n"
            "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE "
            "--kernel-trace` (counters only, own run); MFMA-pipe busy = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8). Durations under "
            "counter collection are longer than in the kernel-trace tables.\n\n"
            "## one bench step + warm-up: `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-k64 --no-realistic`\n\n")
    f.write(open(os.path.join(O, "pmc_kernels_bench.md")).read())
    f.write("\n## clustering stage on planted embeddings (block-sparse kernel): `python tools/ms_stage_only.py 64 1`\n\n")
    f.write(open(os.path.join(O, "pmc_kernels_msstage.md")).read())
with open(os.path.join(P, f"{tag}_sparse_and_gemm_tools.md"), "w") as f:
    f.write(f"# Tool outputs (1 x MI355X, round 2, commit {commit})\n")
    for name in ("sparse_check", "sparse_breakdown", "pointwise_bench", "f16h_check"):
        if not os.path.exists(os.path.join(O, name + ".out")):
            continue
        txt = [l for l in open(os.path.join(O, name + ".out")).read().splitlines() if "amdgpu.ids" not in l]
        f.write(f"\n## tools/{ {'sparse_check': 'ms_sparse_f16_check.py 64', 'sparse_breakdown': 'ms_sparse_breakdown.py 64', 'pointwise_bench': 'pointwise_bench.py', 'f16h_check': 'ms_f16h_check.py 64'}[name] }\n\n```\n" + "\n".join(txt) + "\n```\n")
if os.path.exists(os.path.join(O, "pmc_f16_summary.md")):
    open(os.path.join(P, f"{tag}_pmc_ms_iterate_f16.md"), "w").write(
        f"# PMC summary of ms_iterate_d128_f16r_kernel<false, false> (the default dense kernel): `python tools/ms_iter_only.py 64 50 128 f16` "
        f"(SQ passes: 10 iterations) (commit {commit})\n\n" + open(os.path.join(O, "pmc_f16_summary.md")).read())
    # HBM-side traffic record bench.py stamps into roofline.traffic (separate FETCH_SIZE / WRITE_SIZE passes of the 50-iteration launch)
    vals = {}
    for line in open(os.path.join(O, "pmc_f16_summary.md")):
        c = [x.strip() for x in line.split("|")]
        if len(c) >= 3 and c[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[c[1]] = float(c[2])
    if len(vals) == 2:
        json.dump({"kernel": "ms_iterate_d128_f16r_kernel<false, false>", "schedule": "split-fp16", "clouds": 64, "iterations": 50,
                   "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                   "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
                   "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read, "
                              "MI355X_MICROARCH.md section HBM; Infinity-Cache hits are counted)",
                   "algorithmic_bytes_per_launch": 64 * 2 * 10000 * 128 * 4,
                   "command": "python tools/ms_iter_only.py 64 50 128 f16", "commit": commit,
                   "date": f"round 2, tools/profile_round.sh (profiles/{tag}_pmc_ms_iterate_f16.md)"},
                  open(os.path.join(P, f"{tag}_pmc_ms_iterate.json"), "w"), indent=1)
print("written:", sorted(x for x in os.listdir(P) if x.startswith(tag)))

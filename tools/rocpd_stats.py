#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database into a per-kernel table (calls, total, avg, min, max, %),
the same information `--stats` prints for CSV output. Usage: python tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else ("kernel_name" if "kernel_name" in sym_cols else "name")
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1 order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, c, t, mn, mx in rows:
        out.append(f"| `{short(n)}` | {c} | {t / 1e6:.3f} | {t / c / 1e6:.4f} | {mn / 1e6:.4f} | {mx / 1e6:.4f} | {100 * t / total:.2f} |")
    out.append(f"\ntotal kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()

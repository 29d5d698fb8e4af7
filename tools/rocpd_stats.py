"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite output) kernel trace into the `--stats`-style per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .*\]", "", name)
    return name.strip()[:140]


def main(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    q = f"""select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{namecol} order by 3 desc"""
    rows = c.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for n, calls, tot, mn, mx in rows:
        print(f"\"{short(n)}\",{calls},{tot},{tot / calls:.0f},{100.0 * tot / total:.3f},{mn},{mx}")


if __name__ == "__main__":
    main(sys.argv[1])

"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share.
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {ncol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {ncol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
    for n, k, s, a, mn, mx in rows:
        print(f"\"{short(n)}\",{k},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])

"""Implied shader clock per kernel from a rocprofv3 run with --kernel-trace --pmc GRBM_GUI_ACTIVE:
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration.
    python tools/rocpd_clock.py <db>"""
import collections
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    print("# columns:", ",".join(cols))
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    q = f"select kernel_name, dispatch_id, sum(value), min({s}), max({e}) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name, dispatch_id"
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, d, v, t0, t1 in c.execute(q):
        a = agg[re.sub(r"\(anonymous namespace\)::", "", k)[:60]]
        a[0] += v / 8.0
        a[1] += (t1 - t0)
        a[2] += 1
    print("kernel,dispatches,avg_us,avg_cycles_per_xcd,implied_GHz")
    for k, (cy, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'"{k}",{n},{ns / n / 1e3:.1f},{cy / n:.0f},{cy / max(ns, 1):.3f}')


if __name__ == "__main__":
    main(sys.argv[1])

"""Per-kernel PMC sums from rocprofv3 rocpd sqlite output(s).
    python tools/rocpd_pmc.py gpurun_out/pmc/*/*.db
Prints, per kernel name: dispatches and the mean per-dispatch value of every counter found."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:70]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in paths:
        c = sqlite3.connect(path)
        views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
        if "counters_collection" not in views:
            continue
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x and "name" in x][0]
        ncol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
        vcol = "value" if "value" in cols else [x for x in cols if "value" in x][0]
        dcol = "dispatch_id" if "dispatch_id" in cols else None
        q = f"select {kcol}, {ncol}, {dcol or 0}, sum({vcol}) from counters_collection group by {kcol}, {ncol}, {dcol or 0}"
        for k, n, d, v in c.execute(q):
            a = agg[short(k)][n]
            a[0] += v
            a[1] += 1
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        print(k)
        for n, (s, cnt) in sorted(agg[k].items()):
            print(f"    {n:32s} dispatches={cnt:5d}  mean/dispatch={s / max(cnt, 1):.4g}  total={s:.6g}")


if __name__ == "__main__":
    main(sys.argv[1:])

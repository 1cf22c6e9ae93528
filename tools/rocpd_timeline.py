"""Per-dispatch timeline from a rocprofv3 (rocpd sqlite) kernel trace: every kernel of the LAST `--steps`-th part of the run in start
order with grid, duration and the gap since the previous kernel ended (same process, all streams).
    python tools/rocpd_timeline.py <db> [--last N]      # N = number of trailing dispatches to print (default: all)
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"nope::", "", name)
    name = re.sub(r"\(nope::ConvParams\)|\(ConvParams\)", "", name)
    return name[:84]


def main(path, last):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("# columns:", ",".join(cols))
    ncol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    g = [x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols]
    w = [x for x in ("workgroup_x", "workgroup_size_x", "workgroup_size") if x in cols]
    extra = []
    for cand in ("grid_x", "grid_y", "grid_z", "workgroup_x", "stream_id", "queue_id", "lds_size", "vgpr_count"):
        if cand in cols:
            extra.append(cand)
    q = f"select {ncol}, start, end" + "".join(", " + e for e in extra) + " from kernels order by start"
    rows = c.execute(q).fetchall()
    if last:
        rows = rows[-last:]
    print("idx,start_us,dur_us,gap_us," + ",".join(extra) + ",kernel")
    t0 = rows[0][1]
    prev_end = None
    for i, r in enumerate(rows):
        gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(prev_end or 0, r[2])
        print(f"{i},{(r[1] - t0) / 1e3:.1f},{(r[2] - r[1]) / 1e3:.2f},{gap:.2f}," + ",".join(str(x) for x in r[3:]) + f",\"{short(r[0])}\"")


if __name__ == "__main__":
    last = 0
    if "--last" in sys.argv:
        last = int(sys.argv[sys.argv.index("--last") + 1])
    main(sys.argv[1], last)

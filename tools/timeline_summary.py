"""Per (kernel, grid) totals of a tools/rocpd_timeline.py CSV: calls, average and total microseconds.   python tools/timeline_summary.py <csv> [steps] [name width]"""
import collections
import csv
import re
import sys

rows = list(csv.reader(open(sys.argv[1])))[2:]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"_ZN4nope12_GLOBAL__N_1\d+", "", r[-1])
    name = re.sub(r"\(nope::ConvParams\)|\((?:float|unsigned|void|int|nope::).*", "", name)[:int(sys.argv[3]) if len(sys.argv) > 3 else 34]
    agg[(name, r[4], r[5], r[6])].append(float(r[2]))
tot = sum(sum(v) for v in agg.values())
print(f"# {len(rows)} dispatches, {tot / steps:.0f} us of kernel time per step ({steps:g} steps)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:{(int(sys.argv[3]) if len(sys.argv) > 3 else 34) + 2}s} grid {k[1]:>8},{k[2]},{k[3]:<2} x{len(v) / steps:5.1f}  avg {sum(v) / len(v):7.2f} us  per step {sum(v) / steps:7.1f} us")

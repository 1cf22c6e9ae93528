#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/gn_bench.txt
for args in "--act 1" "--act 0" "--act 1 --emb 1" "--act 1 --resid 1"; do
  echo "## $args  (NOPE_GN_VARIANT=${NOPE_GN_VARIANT:-0})" >> $OUT/gn_bench.txt
  rm -rf /tmp/gnprof
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/gnprof -o gn -- python $GRAFT_REPO_ROOT/tools/gn_bench.py $args 2>/dev/null | grep "^gn " >> $OUT/gn_bench.txt)
  f=$(find /tmp/gnprof -name "*kernel_trace.csv" | head -1)
  python - "$f" >> $OUT/gn_bench.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gn_" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:48], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)[: max(1, len(v) - 3)]
    print("   ", k[0], "grid", k[1], "n", len(v), "avg_us", round(sum(v) / len(v), 1), "min", round(v[0], 1))
PY
done
cat $OUT/gn_bench.txt

#!/bin/bash
# round 4, run i: timeline of a 26-template step; the new encoder graph test; CPU-side checks on the box are not needed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "encoder_graph or graph_replay or geodesic" > $OUT/pytest_r04i.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_r04i.log
for n in 26; do
( cd /tmp && rm -rf /tmp/prof_n && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_n -o b -- python $OLDPWD/bench.py --dtype f16 --templates $n --steps 4 --warmup 2 --skip-extras > $OUT/prof_n$n.log 2>&1 )
python tools/rocpd_timeline.py $(find /tmp/prof_n -name "*.db" | head -1) > $OUT/timeline_all.csv
python - $n <<'PY'
import csv, sys
n=sys.argv[1]
rows=list(csv.reader(open('gpurun_out/timeline_all.csv')))
hdr, body = rows[:2], rows[2:]
stems=[i for i,r in enumerate(body) if 'stem_conv' in r[-1]]
start=stems[-8]
w=csv.writer(open(f'gpurun_out/timeline_n{n}.csv','w'))
for r in hdr: w.writerow(r)
for r in body[start:]: w.writerow(r)
PY
python tools/timeline_summary.py $OUT/timeline_n$n.csv 4 | head -60 | tee $OUT/timeline_n${n}_summary.txt
done
rm -f $OUT/timeline_all.csv
echo done

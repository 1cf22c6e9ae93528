#!/bin/bash
# round 4, run e: warm / cold weight streams of the split-K launches (prefetch on / off), reduce kernel v2, parallel group sums,
# finer gn_apply grid, deep ring of the small-tile kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_small.py tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_r04e.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_r04e.log
timeout 300 python tools/conv_split_bench.py > $OUT/conv_split_bench.txt 2>&1; cat $OUT/conv_split_bench.txt
NOPE_PP_VARIANT=2048 timeout 300 python tools/conv_split_bench.py > $OUT/conv_split_bench_nopf.txt 2>&1; echo "-- no prefetch"; grep default $OUT/conv_split_bench_nopf.txt
timeout 900 python tools/small_bank_sweep.py --dtype f16 --settings ";NOPE_PP_VARIANT=2048;NOPE_SMALL_DEEP_MAX=0;NOPE_SMALL_DEEP_MAX=1024;NOPE_GN_MIN_GRID=0;NOPE_HALO_SPLIT_MIN_CHUNKS=6" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; echo "sweep rc=$?"; cat $OUT/small_bank_sweep.txt; tail -3 $OUT/sweep.err
for n in 64; do
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
python tools/timeline_summary.py $OUT/timeline_n$n.csv 4 | head -50 | tee $OUT/timeline_n${n}_summary.txt
done
rm -f $OUT/timeline_all.csv
timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench.txt 2>&1; cat $OUT/encoder_bench.txt
echo done

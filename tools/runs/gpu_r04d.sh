#!/bin/bash
# round 4, run d: split-K partials through the LDS panels, reduce kernel with fused GroupNorm statistics, gn_apply grid for small batches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_small.py tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_r04d.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_r04d.log
timeout 900 python tools/small_bank_sweep.py --dtype f16 --settings ";NOPE_GN_MIN_GRID=0;NOPE_HALO_SPLIT=0;NOPE_HALO_SPLIT_MIN_CHUNKS=6;NOPE_GN_MIN_GRID=1024;NOPE_GN_FOLD_INLINE=0" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; echo "sweep rc=$?"; cat $OUT/small_bank_sweep.txt; tail -3 $OUT/sweep.err
for n in 64 341; do
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
python tools/timeline_summary.py $OUT/timeline_n$n.csv 4 | head -45 | tee $OUT/timeline_n${n}_summary.txt
done
rm -f $OUT/timeline_all.csv
timeout 300 python bench.py --dtype f16 --templates 341 --steps 6 --warmup 2 --extras roofline > $OUT/bench_n341_classes.json 2>$OUT/bench_n341.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n341_classes.json'))
print('n341', round(d['ms_per_step'],3),'ms')
tot=0
for c in d['roofline']['classes']:
    tot+=c['avg_ms']*c['launches']
    print(f"{c['kernel'][:22]:>22} mode {c['mode']} taps {c['taps']} {c['Cin']:>4}->{c['Cout']:<4} @{c['H']}x{c['W']} n={c['n']:<3} x{c['launches']:<2} {c['avg_ms']*1e3:8.1f} us {c['tflops']:7.1f} TF {c['frac']:.3f}")
print('conv total ms', tot)
PY
echo done

#!/bin/bash
# round 4, run c: split-K on the tap-resident kernel for the 4x4 / 8x8 levels of small banks, small-tile kernel only where the tap-resident
# one has no tiles, constants of the small-tile epilogue in LDS: GPU tests of the touched kernels, the policy sweep, the per-launch table and a
# rocprofv3 timeline at 64 hypotheses, the encoder at 1 / 2 / 8 images.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_small.py tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_r04c.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_r04c.log
timeout 900 python tools/small_bank_sweep.py --dtype f16 --settings ";NOPE_HALO_SPLIT=0;NOPE_HALO_SPLIT=0,NOPE_CONV_SMALL=0,NOPE_GN_FOLD_INLINE=0;NOPE_HALO_SPLIT_MIN_CHUNKS=6;NOPE_HALO_SPLIT_MIN_CHUNKS=18;NOPE_SMALL_MAX_TILES=160;NOPE_SMALL_MAX_TILES=480" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; echo "sweep rc=$?"; cat $OUT/small_bank_sweep.txt; tail -3 $OUT/sweep.err
timeout 300 python bench.py --dtype f16 --templates 64 --steps 10 --warmup 3 --extras roofline > $OUT/bench_n64_classes.json 2>$OUT/bench_n64.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n64_classes.json'))
print('n64', round(d['ms_per_step'],3),'ms')
tot=0
for c in d['roofline']['classes']:
    tot+=c['avg_ms']*c['launches']
    print(f"{c['kernel'][:22]:>22} mode {c['mode']} taps {c['taps']} {c['Cin']:>4}->{c['Cout']:<4} @{c['H']}x{c['W']} n={c['n']:<3} x{c['launches']:<2} {c['avg_ms']*1e3:8.1f} us {c['tflops']:7.1f} TF {c['frac']:.3f}")
print('conv total ms', tot)
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_n64 -o b -- python $OLDPWD/bench.py --dtype f16 --templates 64 --steps 4 --warmup 2 --skip-extras > $OUT/prof_n64.log 2>&1 )
python tools/rocpd_timeline.py $(find /tmp/prof_n64 -name "*.db" | head -1) > $OUT/timeline_n64_all.csv
python - <<'PY'
# keep the dispatches of the last 4 steps: find the 6 stem_conv pairs -> the last 8 stem launches (2 per step)
import csv
rows=list(csv.reader(open('gpurun_out/timeline_n64_all.csv')))
hdr, body = rows[:2], rows[2:]
stems=[i for i,r in enumerate(body) if 'stem_conv' in r[-1]]
start=stems[-8]
w=csv.writer(open('gpurun_out/timeline_n64.csv','w'))
for r in hdr: w.writerow(r)
for r in body[start:]: w.writerow(r)
print('kept', len(body)-start, 'dispatches of 4 steps')
PY
python tools/timeline_summary.py $OUT/timeline_n64.csv 4 | head -60 | tee $OUT/timeline_n64_summary.txt
rm -f $OUT/timeline_n64_all.csv
timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench.txt 2>&1; cat $OUT/encoder_bench.txt
NOPE_CONV_SMALL=0 timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench_old.txt 2>&1; cat $OUT/encoder_bench_old.txt
echo done

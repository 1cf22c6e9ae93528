#!/bin/bash
# round 4, run f: the whole GPU suite on the new launch policy, the default bench line, and the deep ring restricted to one-round launches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -18 $OUT/pytest_gpu.log
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 26,64,91,341 --settings ";NOPE_SMALL_DEEP_MAX=256;NOPE_SMALL_DEEP_MAX=128" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; cat $OUT/small_bank_sweep.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['scaling'], 'tolerance_met', d.get('tolerance_met'), 'within_tol', d.get('value_within_tolerance'), d.get('value_within_tolerance_mode'), 'margin', d.get('top1_margin'))
print(r['kernel'], round(r['frac'],3), 'traffic', r['traffic'])
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('top1_margin'))
for l in d.get('scaling_lines',[]): print(' ', l['name'], round(l['value']), round(l['ms_per_step'],2))
print(d['cpu_baseline']['value'], d['cpu_baseline'].get('value_reference_schedule'), d['cpu_baseline']['cores'])
PY
echo done

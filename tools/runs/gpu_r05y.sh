#!/bin/bash
# round 5, run y: the per-tap f16x2 tree -- smoke, a quick default bench (power_ceiling record included), probe --json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
tools/probes/overlap_probe --json | tee $OUT/r05y_power_ceiling.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05y_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/r05y_smoke.log; grep -v amdgpu $OUT/r05y_smoke.log | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/r05y_bench.json 2> $OUT/r05y_bench.err; echo "bench rc=$?"; tail -2 $OUT/r05y_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05y_bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'ms', 'tolerance_met', d.get('tolerance_met'))
print(r['kernel'], round(r['frac'],3), r['achieved'])
print(json.dumps(r['power_ceiling']))
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('top1_margin'))
PY

#!/bin/bash
# round 4, run p (= run j on the final sources): PMC passes on the final kernel sources (profiles/pmc_traffic.json is tied to their hash), then smoke, the whole GPU suite,
# the default bench line with the fresh PMC record in place, and the rocprofv3 kernel stats of the same command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16 > /dev/null 2>&1; echo "pmc unet done"; head -3 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16 --sim $OUT/pmc_sim.txt | cut -c1-600
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['scaling'], 'tolerance_met', d.get('tolerance_met'), 'within_tol', round(d.get('value_within_tolerance') or 0), d.get('value_within_tolerance_mode'), 'margin', d.get('top1_margin'))
print(r['kernel'], round(r['frac'],3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'], 'avg_launch_ms', r['avg_launch_ms'])
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('top1_margin'))
for l in d.get('scaling_lines',[]): print(' ', l['name'], round(l['value']), round(l['ms_per_step'],2))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('value_reference_schedule'), d['cpu_baseline']['cores'], 'speedup', d['speedup_vs_cpu'])
for s in d['scoring_roofline']: print('  scoring', s['bank_dtype'], s['N'], round(s['frac'],3))
PY
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --skip-extras > "$OLDPWD/$OUT/prof.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) > $OUT/bench_f16_kernel_stats.csv; head -16 $OUT/bench_f16_kernel_stats.csv | cut -c1-150
echo done

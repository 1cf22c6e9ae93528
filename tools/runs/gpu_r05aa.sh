#!/bin/bash
# round 5, run aa: records on the FINAL kernel sources (per-tap f16x2, per-tap K loop without tuning tests) -- probes, PMC passes of the f16x2 U-Net step and the scoring kernel (profiles/pmc_traffic.json is tied to
# the source hash), smoke, the whole GPU suite (-s: observed errors in the log), the default bench line with the fresh PMC record in place, rocprofv3
# kernel stats of the same command, the driver's N > 1 command with four gloo ranks on the one GPU, reference-sized banks per mode
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
tools/probes/mx_probe > $OUT/r05aa_mx_probe.txt 2>&1; tail -1 $OUT/r05aa_mx_probe.txt
timeout 200 tools/probes/overlap_probe > $OUT/r05aa_overlap_probe.txt 2>&1; tail -3 $OUT/r05aa_overlap_probe.txt
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16x2 > /dev/null 2>&1; echo "pmc unet done"; head -3 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16x2 --sim $OUT/pmc_sim.txt | cut -c1-400
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05aa_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/r05aa_smoke.log; tail -2 $OUT/r05aa_smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/r05aa_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r05aa_pytest_gpu.log; grep -v "amdgpu.ids" $OUT/r05aa_pytest_gpu.log | tail -16
timeout 900 python bench.py > $OUT/r05aa_bench.json 2> $OUT/r05aa_bench.err; echo "bench rc=$?"; tail -2 $OUT/r05aa_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05aa_bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['scaling'], 'tolerance_met', d.get('tolerance_met'), 'within_tol', round(d.get('value_within_tolerance') or 0), d.get('value_within_tolerance_mode'), 'margin', d.get('top1_margin'))
print(r['kernel'], round(r['frac'],3), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'], 'avg_launch_ms', r['avg_launch_ms'])
print('power_ceiling', json.dumps(r.get('power_ceiling')))
for k,v in d['parity']['modes'].items(): print(' ', k, round(v['hyp_per_s']), v['score_rel_err'], v['top5_equal'], v.get('top1_margin'))
for l in d.get('scaling_lines',[]): print(' ', l['name'], round(l['value']), round(l['ms_per_step'],2))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('value_reference_schedule'), d['cpu_baseline']['cores'], 'speedup', d['speedup_vs_cpu'])
for s in d['scoring_roofline']: print('  scoring', s['bank_dtype'], s['N'], round(s['frac'],3))
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --skip-extras > "$OLDPWD/$OUT/prof.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) > $OUT/r05aa_bench_f16x2_kernel_stats.csv; head -12 $OUT/r05aa_bench_f16x2_kernel_stats.csv | cut -c1-150
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo > $OUT/r05aa_bench_4rank_gloo_one_gpu.json 2> $OUT/bench_4rank.err; echo "4-rank rc=$?"; tail -2 $OUT/bench_4rank.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05aa_bench_4rank_gloo_one_gpu.json'))
print('4 ranks (one GPU, gloo):', d['scaling'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['config']['templates_per_gpu'], 'per rank', d['config']['top5'])
for l in d.get('scaling_lines',[]): print(' ', l['name'], l['scaling'], round(l['value']), round(l['ms_per_step'],2))
PY
timeout 900 python tools/small_bank_sweep.py --dtype f16x2 --banks 26,64,91,128,256,341,512 --steps 20 --settings ";" > $OUT/r05aa_small_banks_f16x2.txt 2>$OUT/sweep.err; cat $OUT/r05aa_small_banks_f16x2.txt
bash tools/gpu_clock.sh > /dev/null 2>&1; cp $OUT/clock_by_kernel.csv $OUT/r05aa_clock_by_kernel.csv; head -8 $OUT/r05aa_clock_by_kernel.csv | cut -c1-160
echo done

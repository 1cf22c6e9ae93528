#!/bin/bash
# round 4, run g: the driver's N > 1 command with four gloo ranks on the one GPU; short-K 1x1 convs on the small-tile kernel at 512
# hypotheses (per-launch tables); the read + write streaming ceiling; rocprofv3 kernel trace of the 1 GB scoring launches; the default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo > $OUT/bench_4rank_gloo_one_gpu.json 2> $OUT/bench_4rank.err; echo "4-rank rc=$?"; tail -2 $OUT/bench_4rank.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_4rank_gloo_one_gpu.json'))
print('4 ranks (one GPU, gloo):', d['scaling'], round(d['value']), round(d['ms_per_step'],2), 'ms', d['config']['templates_per_gpu'], 'per rank', d['config']['top5'])
for l in d.get('scaling_lines',[]): print(' ', l['name'], l['scaling'], round(l['value']), round(l['ms_per_step'],2))
PY
timeout 300 python bench.py --steps 3 --warmup 1 --skip-extras > $OUT/bench_1rank_ref.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_1rank_ref.json')); print('1 rank:', d['config']['top5'], round(d['value']))"
for v in 0 6; do
NOPE_SMALL_1X1_MAXK=$v timeout 300 python bench.py --steps 5 --warmup 2 --extras roofline > $OUT/bench_1x1_$v.json 2>$OUT/bench_1x1.err
python - $v <<'PY'
import json, sys
v=sys.argv[1]
d=json.load(open(f'gpurun_out/bench_1x1_{v}.json'))
print(f'NOPE_SMALL_1X1_MAXK={v}:', round(d['ms_per_step'],3),'ms/step')
for c in d['roofline']['classes']:
    if c['taps']==1 and c['mode']==0:
        print(f"  {c['kernel'][:22]:>22} {c['Cin']:>4}->{c['Cout']:<4} @{c['H']}x{c['W']} x{c['launches']:<2} {c['avg_ms']*1e3:8.1f} us {c['tflops']:7.1f} TF")
PY
done
timeout 120 python tools/copy_ceiling.py | tee $OUT/copy_ceiling.txt
( cd /tmp && rm -rf /tmp/prof_sim && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sim -o s -- python $OLDPWD/tools/sim_step.py --templates 2048 --dtype bf16 > $OUT/prof_sim.log 2>&1 )
python tools/rocpd_stats.py $(find /tmp/prof_sim -name "*.db" | head -1) > $OUT/sim_bf16_32x2048_kernel_stats.csv; head -4 $OUT/sim_bf16_32x2048_kernel_stats.csv | cut -c1-200
( cd /tmp && rm -rf /tmp/prof_sim && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sim -o s -- python $OLDPWD/tools/sim_step.py --templates 1024 --dtype f16 > $OUT/prof_sim2.log 2>&1 )
python tools/rocpd_stats.py $(find /tmp/prof_sim -name "*.db" | head -1) > $OUT/sim_f16_32x1024_kernel_stats.csv; head -3 $OUT/sim_f16_32x1024_kernel_stats.csv | cut -c1-200
echo done
